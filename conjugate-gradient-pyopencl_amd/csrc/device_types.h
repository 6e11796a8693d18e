// Value-type traits and arithmetic helpers shared by all gfx950 kernels.
//
// Four value types: float, double, float2 (complex64, interleaved re/im),
// double2 (complex128).  Complex arithmetic is componentwise exactly as the
// reference defines it (kernel/complex/cmplx.h:6-25); the dot product is
// UNCONJUGATED (kernel/complex/vdot.cl:15).
#pragma once
#include <hip/hip_runtime.h>

namespace cgamd {

constexpr int kWave = 64;  // CDNA4 wavefront

template <typename T> struct VT;
template <> struct VT<float> {
    using acc = double; using real = float;
    static constexpr int dtype = 0; static constexpr bool cplx = false;
};
template <> struct VT<double> {
    using acc = double; using real = double;
    static constexpr int dtype = 1; static constexpr bool cplx = false;
};
template <> struct VT<float2> {
    using acc = double2; using real = float;
    static constexpr int dtype = 2; static constexpr bool cplx = true;
};
template <> struct VT<double2> {
    using acc = double2; using real = double;
    static constexpr int dtype = 3; static constexpr bool cplx = true;
};

#define CG_DEV __device__ __forceinline__

template <typename T> CG_DEV T vzero();
template <> CG_DEV float vzero<float>() { return 0.f; }
template <> CG_DEV double vzero<double>() { return 0.; }
template <> CG_DEV float2 vzero<float2>() { return make_float2(0.f, 0.f); }
template <> CG_DEV double2 vzero<double2>() { return make_double2(0., 0.); }

CG_DEV float vadd(float a, float b) { return a + b; }
CG_DEV double vadd(double a, double b) { return a + b; }
CG_DEV float2 vadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
CG_DEV double2 vadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }

CG_DEV float vsub(float a, float b) { return a - b; }
CG_DEV double vsub(double a, double b) { return a - b; }
CG_DEV float2 vsub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
CG_DEV double2 vsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

CG_DEV float vmul(float a, float b) { return a * b; }
CG_DEV double vmul(double a, double b) { return a * b; }
CG_DEV float2 vmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
CG_DEV double2 vmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// c ? a : b, component-wise (a ?: on the HIP vector structs is lowered through scratch memory)
CG_DEV float vsel(bool c, float a, float b) { return c ? a : b; }
CG_DEV double vsel(bool c, double a, double b) { return c ? a : b; }
CG_DEV float2 vsel(bool c, float2 a, float2 b) { return make_float2(c ? a.x : b.x, c ? a.y : b.y); }
CG_DEV double2 vsel(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }

// c + a*b
template <typename T> CG_DEV T vfma(T a, T b, T c) { return vadd(c, vmul(a, b)); }

// y = a*y + x of the search-direction update (reference aypx.cl:9, complex/aypx.cl:11).  The two-launch loop recomputes
// this value for every gathered column inside the SpMV launch and must get the bits the element-wise kernels store: the
// library is compiled with -ffp-contract=off (csrc/Makefile), so every expression is evaluated as written -- one rounding
// per multiply and per add, like the CPU oracle -- whatever code surrounds it.
template <typename T> CG_DEV T vaypx(T a, T y, T x) { return vadd(vmul(a, y), x); }

// widening to the accumulator type used by reductions and scalar math
CG_DEV double to_acc(float a) { return (double)a; }
CG_DEV double to_acc(double a) { return a; }
CG_DEV double2 to_acc(float2 a) { return make_double2((double)a.x, (double)a.y); }
CG_DEV double2 to_acc(double2 a) { return a; }

template <typename T> CG_DEV T from_acc(typename VT<T>::acc a);
template <> CG_DEV float from_acc<float>(double a) { return (float)a; }
template <> CG_DEV double from_acc<double>(double a) { return a; }
template <> CG_DEV float2 from_acc<float2>(double2 a) { return make_float2((float)a.x, (float)a.y); }
template <> CG_DEV double2 from_acc<double2>(double2 a) { return a; }

// accumulator <-> (x, y) pair, for type-agnostic transport of real and complex sums
CG_DEV double2 to_acc2(double a) { return make_double2(a, 0.); }
CG_DEV double2 to_acc2(double2 a) { return a; }
template <typename A> CG_DEV A from_acc2(double2 a);
template <> CG_DEV double from_acc2<double>(double2 a) { return a.x; }
template <> CG_DEV double2 from_acc2<double2>(double2 a) { return a; }

// scalar division used for alpha = delta/dq and beta = delta_new/delta_old
// (reference clcg.c:326-327,389-391).  Complex: Smith's algorithm, as C99 / numpy do.
CG_DEV double acc_div(double a, double b) { return a / b; }
CG_DEV double2 acc_div(double2 a, double2 b) {
    if (fabs(b.x) >= fabs(b.y)) {
        const double ratio = b.y / b.x, den = b.x + b.y * ratio;
        return make_double2((a.x + a.y * ratio) / den, (a.y - a.x * ratio) / den);
    }
    const double ratio = b.x / b.y, den = b.x * ratio + b.y;
    return make_double2((a.x * ratio + a.y) / den, (a.y * ratio - a.x) / den);
}

// ---- cross-lane reductions (wave64) -----------------------------------------
// wave_sum: the shuffle-down tree  v[l] += v[l + off], off = 32, 16, 8, 4, 2, 1  -- result valid in LANE 0 ONLY -- with the
// cross-lane moves of gfx950 instead of six ds_bpermute round trips: v_permlane32_swap (lanes 0..31 <- 32..63),
// v_permlane16_swap (row 0 <- row 1), then DPP row_shl 8 / 4 / 2 / 1 inside row 0.  The pairs added are the tree's, so the
// sum has the tree's bits (scripts/microbench/wave_sum_probe.hip checks it against __shfl_down); a call must be reached by
// the whole wave.
template <int N> CG_DEV double dpp_row_shl(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, 0x100 + N, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), 0x100 + N, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// v_permlane32_swap / v_permlane16_swap are cross-lane operations whose lane enables come from EXEC.  Every reduction starts with five
// wait states tied to its first operand (the asm's output feeds the swap, so nothing is scheduled between them): a guard against an
// EXEC write or a 64-bit VALU result landing directly in front of the swap, which the compiler spaces for DPP but which I could not
// confirm for these instructions (suspected, not proven, while chasing the complex64 failure noted at resident.hip member_sum).
CG_DEV unsigned exec_settle(unsigned w) {
    asm volatile("s_nop 4" : "+v"(w));
    return w;
}
CG_DEV double lanes_plus32(double v) {
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(v);
    const unsigned long long b = ((b0 >> 32) << 32) | exec_settle((unsigned)b0);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
CG_DEV double lanes_plus16(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
CG_DEV double wave_sum(double v) {
    v += lanes_plus32(v);
    v += lanes_plus16(v);
    v += dpp_row_shl<8>(v);
    v += dpp_row_shl<4>(v);
    v += dpp_row_shl<2>(v);
    v += dpp_row_shl<1>(v);
    return v;
}
CG_DEV double2 wave_sum(double2 v) {
    v.x = wave_sum(v.x);
    v.y = wave_sum(v.y);
    return v;
}

// Several wave sums at once with the SAME tree per value (so the same bits as wave_sum), sharing the cross-lane traffic: the
// 32- and 16-lane steps pair values up -- after v_permlane32_swap of (a, b) one add forms a's step in the lower half of the wave and
// b's in the upper half; v_permlane16_swap does the same for two such registers per 16-lane row -- so four values travel in one
// register from there on, one per row, and the DPP steps (which stay inside a row) finish all four together.  8 values: 14 adds
// instead of 48.  Results: wave_sum4 -> {v0, v2, v1, v3} in the first lane of rows 0..3 (lanes 0, 16, 32, 48) of the returned
// register.  Operand pairs are the tree's (an add may see them in the other order: IEEE addition is commutative).
CG_DEV double pair32(double a, double b) {          // lower half: a[l] + a[l + 32]; upper half: b[l - 32] + b[l]
    const unsigned long long ba0 = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    const unsigned long long ba = ((ba0 >> 32) << 32) | exec_settle((unsigned)ba0);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    const double a2 = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    const double b2 = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    return a2 + b2;
}
CG_DEV double pair16(double c, double d) {          // rows 0 / 2: c's 16-lane step; rows 1 / 3: d's
    const unsigned long long bc = (unsigned long long)__double_as_longlong(c), bd = (unsigned long long)__double_as_longlong(d);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)bc, (unsigned)bd, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(bc >> 32), (unsigned)(bd >> 32), false, false);
    const double c2 = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    const double d2 = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    return c2 + d2;
}
CG_DEV double rows_finish(double g) {
    g += dpp_row_shl<8>(g);
    g += dpp_row_shl<4>(g);
    g += dpp_row_shl<2>(g);
    g += dpp_row_shl<1>(g);
    return g;
}
// four values: their wave sums in lanes 0 (v0), 16 (v2), 32 (v1), 48 (v3)
CG_DEV double wave_sum4(double v0, double v1, double v2, double v3) { return rows_finish(pair16(pair32(v0, v1), pair32(v2, v3))); }
// two values: lanes 0 (v0) and 32 (v1)
CG_DEV double wave_sum2(double v0, double v1) {
    double c = pair32(v0, v1);
    c += lanes_plus16(c);
    return rows_finish(c);
}

// Sum over a thread block; result valid in thread 0.  `smem` holds BLOCK/64 entries.
template <int BLOCK, typename A> CG_DEV A block_sum(A v, A *smem) {
    constexpr int NW = BLOCK / kWave;
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    __syncthreads();  // smem may be reused by the caller
    if (lane == 0) smem[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NW; ++i) v = vadd(v, smem[i]);
    }
    return v;
}

// blockIdx -> logical work-group id such that each XCD (blocks are dealt round-robin over the
// 8 XCDs) owns one contiguous range of logical ids, i.e. a contiguous row range whose gathered
// x window stays in that XCD's private 4 MiB L2.  Bijective because G % 8 == 0 is required.
CG_DEV int xcd_remap(int b, int G) { return ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b; }

}  // namespace cgamd
