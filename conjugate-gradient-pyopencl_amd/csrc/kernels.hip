// Hand-written gfx950 kernels of the CG hot path.
//
//   spmv_rowblock_kernel          CSR SpMV (replaces reference kernel/{real,complex}/spmv.cl) fused with the d.q partial
//                                 reduction (reference vdot.cl + host sum clcg.c:317-324): matrix slice through LDS, one lane
//                                 per row; _chunked: 2/4/8 lanes per row for denser rows; _p2p: with the halo push / wait
//                                 POL = -3: the column indices arrive as one-byte codes (aCols[j] = row + dict[code[j]],
//                                 index_offsets_kernel / index_encode_kernel / build_index_codes): 9 instead of 12 B per fp64 non-zero
//   spmm_rowblock_kernel          the same for nRHS > 1 (RHS-major, the ABI layout); the row-major matrix-core path is rowmajor.hip
//   spmv_stream_kernel            generic chunked CSR stream (huge rows, unaligned pointers)
//   dot_partials_kernel           reference kernel/{real,complex}/vdot.cl (partials stay on the device)
//   ewise_kernel                  reference kernel/{real,complex}/{axpy,aypx,sub}.cl
//   axpy_dot / aypx_beta_x        the fused loop: r -= alpha q + r.r partials; beta, x += alpha d, d = beta d + r
//                                 (reference clcg.c:338-416); axpy2_dot / aypx_beta: the form with x updated in the r launch
//   pcg_*                         diagonally preconditioned recurrence (reference helmFE_var.py:546-586)
//   cg_alpha/beta/delta0          the scalar work the reference does on the host (clcg.c:274-292,317-334,376-411)
//   p2p_*                         peer-to-peer mailbox protocol of the row-partitioned multi-GPU loop
//
// Design (MI355X): every kernel is bound by the memory system.  Work-groups are 256 threads (4 wave64).  The SpMV kernels
// run one work-group per 256-row block, dealt block-cyclically over the 8 XCDs; the vector kernels launch <= 2048 (streaming single-RHS systems: 512) grid-
// stride work-groups (256 CUs x 8).  Loads are 16 B per lane and every load instruction of a wave covers contiguous
// memory; matrix streams are non-temporal unless the matrix fits the Infinity Cache.  Reductions are wave64 shuffles,
// then LDS across the 4 waves, then a fixed-order pass over the per-work-group partials: bitwise reproducible run to
// run (atomics only hand out tickets).
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <functional>
#include <mutex>

namespace cgamd {

// =================================================================================================
// SpMV / SpMM, CSR-stream: a work-group owns BLOCK consecutive rows at a time.  Their non-zeros are one
// contiguous slice of aValues/aCols, streamed with 16 B coalesced non-temporal loads (4 nnz per lane
// per load group, slice start rounded down to a multiple of 4 so every load is aligned); each lane
// multiplies its non-zeros with the gathered x entries and parks the products in LDS; after one
// barrier, lane t sums the products of row t (left to right, CSR order) and writes y[t] coalesced.
// For nRHS > 1 the matrix quads stay in registers while the gather/LDS/sum phase repeats per RHS,
// so the matrix is read from HBM once per SpMM.
// =================================================================================================
template <typename T> struct SpmvArgs {
    int n;
    int nrhs;
    long long nnz;
    const T *vals;
    const int *ptr;
    const int *cols;
    const T *x;
    long long ldx;
    T *y;
    long long ldy;
    const T *dvec;                  // fused dot: sum dvec[row] * y[row]
    typename VT<T>::acc *partials;  // [nrhs][grid]
    int row_blocks;
    const int *rb_list;             // row-block kernel: optional explicit list of row blocks (multi-GPU interior / boundary split)
    int rb_count;
    int cap;   // row-block kernel: LDS slice capacity in entries (multiple of 4)
    int cycle; // row-block kernel: block-cyclic schedule over the XCDs, cycle length in row blocks (1 = one contiguous eighth per XCD)
    const unsigned char *codes;   // coded row-block kernel: one byte per non-zero, aCols[j] = row + dict[codes[j]] (build_index_codes)
    const int *dict;              // [256]
};

// Row-block schedule shared by the row-block kernels: work-group b runs on XCD b%8 as that XCD's (b/8)-th block.
// cycle > 1: block-cyclic -- cycles of `cycle` row blocks, XCD j takes the j-th run of ceil(cycle/8) blocks of each;
// cycle <= 1: XCD j owns the j-th contiguous eighth.  Returns -1 for the padding work-groups of the grid.
__host__ __device__ __forceinline__ int rowblock_of(int b, int row_blocks, int cycle) {
    const int xcd = b & 7, i = b >> 3;
    if (cycle > 1) {
        const int chunk = (cycle + 7) >> 3;
        const int k = i / chunk, lo = xcd * chunk + (i - k * chunk);
        const int rb = k * cycle + lo;
        return (lo < cycle && rb < row_blocks) ? rb : -1;
    }
    const int xb = (int)((long long)xcd * row_blocks / 8), xe = (int)((long long)(xcd + 1) * row_blocks / 8);
    return i < xe - xb ? xb + i : -1;
}
static int rowblock_grid(int row_blocks, int cycle) {
    if (cycle > 1) return 8 * ((cycle + 7) / 8) * ((row_blocks + cycle - 1) / cycle);
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) {
        const int m = (int)((long long)(x + 1) * row_blocks / 8) - (int)((long long)x * row_blocks / 8);
        per_xcd = m > per_xcd ? m : per_xcd;
    }
    return per_xcd * 8;
}

template <typename T, int BLOCK, int QPT, bool VEC, bool FUSE_DOT>
__global__ __launch_bounds__(BLOCK) void spmv_stream_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    constexpr int CHUNK = 4 * QPT * BLOCK;
    __shared__ T prod[CHUNK];
    __shared__ A red[BLOCK / kWave];
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];  // FUSE_DOT && nrhs > 1: A[nrhs][BLOCK/64]
    A *wavedot = reinterpret_cast<A *>(dyn_smem);

    const int t = threadIdx.x;
    const int G = gridDim.x;
    const int L = xcd_remap(blockIdx.x, G);
    const int rb_begin = (int)((long long)L * a.row_blocks / G);
    const int rb_end = (int)((long long)(L + 1) * a.row_blocks / G);

    A dot1 = vzero<A>();
    if (FUSE_DOT && a.nrhs > 1) {
        for (int i = t; i < a.nrhs * (BLOCK / kWave); i += BLOCK) wavedot[i] = vzero<A>();
        __syncthreads();
    }

    for (int rb = rb_begin; rb < rb_end; ++rb) {
        const int r0 = rb * BLOCK;
        const int r1 = min(r0 + BLOCK, a.n);
        const int row = r0 + t;
        const int p0 = a.ptr[r0];   // wave-uniform
        const int p1 = a.ptr[r1];
        int s = 0, e = 0;
        if (row < a.n) { s = a.ptr[row]; e = a.ptr[row + 1]; }
        const int cfirst = p0 & ~3;

        for (int c0 = cfirst; c0 < p1 || c0 == cfirst; c0 += CHUNK) {
            // ---- stream this chunk's matrix entries into registers
            T v[QPT][4];
            int c[QPT][4];
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const long long q = (long long)c0 + 4 * (t + u * BLOCK);
                if (q < p1) {
                    if (VEC && q + 4 <= a.nnz) {
                        ld4_nt<T>(a.vals + q, v[u]);
                        const i32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(a.cols + q));
                        c[u][0] = cc.x; c[u][1] = cc.y; c[u][2] = cc.z; c[u][3] = cc.w;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool ok = q + k < a.nnz;
                            v[u][k] = ok ? a.vals[q + k] : vzero<T>();
                            c[u][k] = ok ? a.cols[q + k] : 0;
                        }
                    }
                }
            }
            const int lo = max(s, c0) - c0, hi = min(e, c0 + CHUNK) - c0;
            for (int r = 0; r < a.nrhs; ++r) {
                const T *xr = a.x + (long long)r * a.ldx;
#pragma unroll
                for (int u = 0; u < QPT; ++u) {
                    const long long q = (long long)c0 + 4 * (t + u * BLOCK);
                    if (q < p1) {
                        T pr[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) pr[k] = vmul(v[u][k], xr[c[u][k]]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) prod[4 * (t + u * BLOCK) + k] = pr[k];
                    }
                }
                __syncthreads();
                T sum = vzero<T>();
                for (int k = lo; k < hi; ++k) sum = vadd(sum, prod[k]);
                if (row < a.n) {
                    T *yr = a.y + (long long)r * a.ldy;
                    if (c0 == cfirst) yr[row] = sum;
                    else if (hi > lo) yr[row] = vadd(yr[row], sum);
                }
                if (FUSE_DOT) {
                    const A contrib = (row < a.n) ? to_acc(vmul(a.dvec[(long long)r * a.ldx + row], sum)) : vzero<A>();
                    if (a.nrhs == 1) dot1 = vadd(dot1, contrib);
                    else {
                        // per-wave running sums in LDS; only this wave touches its slot
                        const A w = wave_sum(contrib);
                        if ((t & (kWave - 1)) == 0) {
                            A *slot = &wavedot[r * (BLOCK / kWave) + t / kWave];
                            *slot = vadd(*slot, w);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    if (FUSE_DOT) {
        if (a.nrhs == 1) {
            const A tot = block_sum<BLOCK>(dot1, red);
            if (t == 0) a.partials[L] = tot;
        } else {
            __syncthreads();
            for (int r = t; r < a.nrhs; r += BLOCK) {
                A tot = wavedot[r * (BLOCK / kWave)];
                for (int w = 1; w < BLOCK / kWave; ++w) tot = vadd(tot, wavedot[r * (BLOCK / kWave) + w]);
                a.partials[(long long)r * G + L] = tot;
            }
        }
    }
}

// 4 consecutive matrix entries (values + columns) starting at the 4-aligned entry q: 16-byte loads.  FULL = the
// caller knows q + 4 <= nnz; otherwise the guarded scalar path covers the last, partial quad of the matrix.
template <typename T, bool NT, bool FULL> CG_DEV void load_quad(const T *__restrict__ vals, const int *__restrict__ cols,
                                                                long long nnz, long long q, T (&v)[4], int (&c)[4]) {
    if (FULL || q + 4 <= nnz) {
        ld4<NT>(vals + q, v);
        const i32x4 cc = ld16<i32x4, NT>(cols + q);
        c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = q + k < nnz;
            v[k] = ok ? vals[q + k] : vzero<T>();
            c[k] = ok ? cols[q + k] : 0;
        }
    }
}

// Park the slice [cfirst, p1) of aValues/aCols raw in LDS.  Every lane issues the loads of TWO quads before it
// waits for either (a plain loop made hipcc wait for quad 1 before issuing quad 2: one more dependent HBM
// round trip per work-group, and the work-group's lifetime is a chain of such round trips).
template <typename T, int BLOCK, bool NT, bool FULL>
CG_DEV void stage_slice_impl(const T *__restrict__ vals, const int *__restrict__ cols, long long nnz, int cfirst, int p1,
                             T *sv, int *sc) {
    const int t = threadIdx.x;
    for (long long base = cfirst; base < p1; base += 8 * BLOCK) {
        const long long q0 = base + 4 * t, q1 = q0 + 4 * BLOCK;
        const bool h0 = q0 < p1, h1 = q1 < p1;
        T v0[4], v1[4];
        int c0[4], c1[4];
        if (h0) load_quad<T, NT, FULL>(vals, cols, nnz, q0, v0, c0);
        if (h1) load_quad<T, NT, FULL>(vals, cols, nnz, q1, v1, c1);
        if (h0) {
            const int o = (int)(q0 - cfirst);
#pragma unroll
            for (int k = 0; k < 4; ++k) { sv[o + k] = v0[k]; sc[o + k] = c0[k]; }
        }
        if (h1) {
            const int o = (int)(q1 - cfirst);
#pragma unroll
            for (int k = 0; k < 4; ++k) { sv[o + k] = v1[k]; sc[o + k] = c1[k]; }
        }
    }
}
// Same job with the VALUE stream interleaved across the lanes in 16-byte chunks: chunk k of a lane is chunk k*64 + lane of
// its wave's 256-entry span, so every load instruction of a wave covers one contiguous 1 KB (the plain version above gives
// a lane 4 consecutive entries = 16/32/64 contiguous bytes, and each of its 1/2/4 load instructions touches every cache
// line of the span partially).  With non-temporal loads the partially used lines of complex128 were fetched again by the
// later instructions: 435 -> 313 us for the N=10M SpMV.  Columns (4 B) keep the quad mapping: one 16-byte load per lane.
template <typename T> struct Chunk16;
template <> struct Chunk16<float> { using V = f32x4; };
template <> struct Chunk16<double> { using V = f64x2; };
template <> struct Chunk16<float2> { using V = f32x4; };
template <> struct Chunk16<double2> { using V = f64x2; };
template <typename T, int BLOCK, bool NT, bool FULL, bool CODED = false>
CG_DEV void stage_slice_ilv(const T *__restrict__ vals, const int *__restrict__ cols, long long nnz, int cfirst, int p1,
                            T *sv, int *sc, const unsigned char *__restrict__ codes = nullptr) {
    using V = typename Chunk16<T>::V;
    constexpr int EPC = 16 / (int)sizeof(T);   // values per chunk
    constexpr int NV = 4 / EPC;                // chunks per lane and quad region
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    for (long long base = cfirst; base < p1; base += 8 * BLOCK) {
        V ch[2][NV];
        i32x4 cc[2];
        unsigned cw[2];
        long long ev[2][NV], qc[2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const long long rbase = base + (long long)rg * 4 * BLOCK;
            qc[rg] = rbase + 4 * t;
            if constexpr (CODED) {      // four one-byte codes per lane; the code array is padded, no tail handling
                if (qc[rg] < p1) {
                    const unsigned *cp = reinterpret_cast<const unsigned *>(codes + qc[rg]);
                    cw[rg] = NT ? __builtin_nontemporal_load(cp) : *cp;
                }
            } else if (qc[rg] < p1 && (FULL || qc[rg] + 4 <= nnz)) cc[rg] = ld16<i32x4, NT>(cols + qc[rg]);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                ev[rg][k] = rbase + (long long)wave * 4 * kWave + (long long)(k * kWave + lane) * EPC;
                if (ev[rg][k] < p1 && (FULL || ev[rg][k] + EPC <= nnz)) ch[rg][k] = ld16<V, NT>(vals + ev[rg][k]);
            }
        }
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            if (qc[rg] < p1) {
                const int o = (int)(qc[rg] - cfirst);
                if constexpr (CODED) {
                    *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(sc) + o) = cw[rg];
                } else if (FULL || qc[rg] + 4 <= nnz) {
                    *reinterpret_cast<i32x4 *>(sc + o) = cc[rg];
                } else {
                    for (int j = 0; j < 4; ++j) sc[o + j] = qc[rg] + j < nnz ? cols[qc[rg] + j] : 0;
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (ev[rg][k] < p1) {
                    const int o = (int)(ev[rg][k] - cfirst);
                    if (FULL || ev[rg][k] + EPC <= nnz) {
                        *reinterpret_cast<V *>(sv + o) = ch[rg][k];
                    } else {
                        for (int j = 0; j < EPC; ++j) sv[o + j] = ev[rg][k] + j < nnz ? vals[ev[rg][k] + j] : vzero<T>();
                    }
                }
            }
        }
    }
}

template <typename T> constexpr bool kIlvDefault = true;   // N=10M SpMV: f64 186 -> 165 us, c64 186 -> 166, c128 435 -> 314, f32 109 -> 105 (ab_ilv.log)
template <typename T, int BLOCK, bool NT, int POL = -1>
CG_DEV void stage_slice(const T *__restrict__ vals, const int *__restrict__ cols, long long nnz, int cfirst, int p1, T *sv,
                        int *sc, const unsigned char *__restrict__ codes = nullptr) {
    if (POL == -3) {       // lane-interleaved value chunks + one-byte column codes
        if (((long long)(p1 + 3) & ~3LL) <= nnz) stage_slice_ilv<T, BLOCK, NT, true, true>(vals, cols, nnz, cfirst, p1, sv, sc, codes);
        else stage_slice_ilv<T, BLOCK, NT, false, true>(vals, cols, nnz, cfirst, p1, sv, sc, codes);
        return;
    }
    if (POL == -2) {       // lane-interleaved value chunks
        if (((long long)(p1 + 3) & ~3LL) <= nnz) stage_slice_ilv<T, BLOCK, NT, true>(vals, cols, nnz, cfirst, p1, sv, sc);
        else stage_slice_ilv<T, BLOCK, NT, false>(vals, cols, nnz, cfirst, p1, sv, sc);
        return;
    }
    // only the work-group that owns the very end of the matrix can meet a partial quad: block-uniform branch,
    // so the common path carries no per-lane tail handling (whose control flow made hipcc serialise the loads)
    if (((long long)(p1 + 3) & ~3LL) <= nnz) stage_slice_impl<T, BLOCK, NT, true>(vals, cols, nnz, cfirst, p1, sv, sc);
    else stage_slice_impl<T, BLOCK, NT, false>(vals, cols, nnz, cfirst, p1, sv, sc);
}

// -------------------------------------------------------------------------------------------------
// Fast path (nRHS == 1, slice of BLOCK rows fits LDS; all stencil / FE matrices of the reference):
// "matrix through LDS, one lane per row", ONE row block per work-group, no persistence.
//   * The slice of aValues/aCols that belongs to BLOCK consecutive rows is contiguous; it is streamed with
//     16 B per lane coalesced non-temporal loads (start rounded down to a multiple of 4 entries so every
//     load is aligned; every load instruction of a wave covers one contiguous 1 KB, stage_slice_ilv) and parked RAW
//     in LDS.
//   * After one barrier lane t walks row t out of LDS (up to UNROLL entries in flight).  The x gather of
//     step k is issued by 64 lanes sitting in 64 consecutive rows: for banded matrices their k-th columns are
//     consecutive, so one wave-level gather touches ~4 cache lines (a nnz-per-lane mapping touches ~24).
//   * y is written coalesced; the fused d.q partial is one value per row block (fixed order later).
//   * Schedule: work-group b runs on XCD b%8 (round-robin dispatch) as the (b/8)-th block of that XCD.  The row
//     blocks are dealt BLOCK-CYCLICALLY: cycles of `cycle` row blocks (cgamd_tune "spmv_cycle", default 64), and in
//     every cycle XCD j takes the j-th run of cycle/8 consecutive blocks.  All 8 XCDs therefore sweep the matrix
//     together (the whole chip streams one ~1 MB region at a time and finishes together) while every XCD still
//     gathers x through runs of consecutive row blocks that share cache lines in its private L2.  Against one
//     contiguous eighth of the matrix per XCD (cycle = 1): SpMV 180.6 -> 174.2 us, CG 3208 -> 3366 it/s on the
//     N=10M 7-point system, 3448 -> 3722 it/s on the 9M-row Helmholtz FE matrix; cycles of 8/16 (runs of 1-2 blocks)
//     lose 4-10 %, 32...800 are within 1 % of each other (profiles/r1_experiments/ab_cyc_*.log).
//     Persistent work-groups re-fetched x once per far diagonal (1128 MB against 957 MB algorithmic) and ran
//     12-15 % slower; software-pipelined persistent variants were no faster either
//     (profiles/r1_experiments/ab*.log, pmc_*_summary.txt).
// LDS is sized at launch from the plan's largest slice (values + columns).
// -------------------------------------------------------------------------------------------------
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int UNROLL, int POL = -1>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);                       // [cap]
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));   // [cap] (coded form: cap BYTES)
    __shared__ A red[BLOCK / kWave];
    constexpr bool CODED = POL == -3;
    __shared__ int sdict[CODED ? BLOCK : 1];

    const int t = threadIdx.x;
    if constexpr (CODED) sdict[t] = a.dict[t];      // BLOCK == 256 entries; visible after the staging barrier
    int rb;
    if (a.rb_list) {                       // explicit subset (interior or boundary row blocks of a partition)
        if ((int)blockIdx.x >= a.rb_count) return;
        rb = a.rb_list[blockIdx.x];
    } else {
        rb = rowblock_of(blockIdx.x, a.row_blocks, a.cycle);
        if (rb < 0) return;
    }
    const int r0 = rb * BLOCK;
    const int row = r0 + t;
    // blockIdx.y = right-hand side ("wide" multi-RHS form for small systems, where round trips, not bytes, are the cost:
    // every right-hand side gets its own work-groups and re-stages the slice out of L2, instead of one work-group walking
    // the right-hand sides in groups -- the reference's sub-domain shape, 16k rows x 9: SpMM 10.0 -> see DESIGN.md)
    const T *xr = a.x + (long long)blockIdx.y * a.ldx;
    T *yr = a.y + (long long)blockIdx.y * a.ldy;
    // The work-group's lifetime is a chain of dependent memory round trips; keep it at three: {row pointers}
    // -> {matrix slice} -> {x gather}.  The per-row pointers are loaded here, branch-free (clamped row), together
    // with the slice bounds, and only consumed after the barrier.
    const int rclamp = min(row, a.n - 1);
    const int s_raw = a.ptr[rclamp], e_raw = a.ptr[rclamp + 1];
    const int p0 = a.ptr[r0], p1 = a.ptr[min(r0 + BLOCK, a.n)];
    const int cfirst = p0 & ~3;
    stage_slice<T, BLOCK, NT, POL>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc, a.codes);
    const int s = s_raw - cfirst, e = (row < a.n) ? e_raw - cfirst : s_raw - cfirst;
    __syncthreads();
    // Row walk, branch-free inside a batch: out-of-range slots re-read the row's LAST entry (a valid LDS slot
    // and a column this row uses anyway) and their term is dropped by a select.  Per-slot `if`s made hipcc emit
    // one exec-masked branch + LDS wait per entry, which serialised the issue of the gathers.
    T sum = vzero<T>();
    for (int k = s; k < e; k += UNROLL) {
        T xv[UNROLL], av[UNROLL];
        int cj[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int idx = min(k + j, e - 1);
            if constexpr (CODED) cj[j] = reinterpret_cast<const unsigned char *>(sc)[idx];
            else cj[j] = sc[idx];
            av[j] = sv[idx];
        }
        if constexpr (CODED) {
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) cj[j] = row + sdict[cj[j]];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) xv[j] = xr[cj[j]];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const T nxt = vfma(av[j], xv[j], sum);
            sum = vsel(k + j < e, nxt, sum);
        }
    }
    A dot1 = vzero<A>();
    if (row < a.n) {
        yr[row] = sum;
        if (FUSE_DOT) dot1 = to_acc(vmul(a.dvec[row + (long long)blockIdx.y * a.ldx], sum));
    }
    if (FUSE_DOT) {
        const A tot = block_sum<BLOCK>(dot1, red);
        if (t == 0) a.partials[(long long)blockIdx.y * a.row_blocks + rb] = tot;
    }
}

// -------------------------------------------------------------------------------------------------
// The same kernel for DENSER rows (a 256-row slice no longer fits LDS: > ~21 non-zeros per row in fp64).  The work-group
// still owns 256 rows and writes one d.q partial, but stages and walks them in LPR chunks of 256/LPR rows, LPR = 2, 4 or
// 8 lanes per row: lane l of a row takes entries s+l, s+l+LPR, ... (consecutive lanes -> consecutive entries -> for
// stencil / FE rows consecutive columns), the LPR partial sums meet in a shuffle tree.  The generic kernel, which these
// matrices used before, runs the 27-point stencil at 50 % of the HBM roofline.
// -------------------------------------------------------------------------------------------------
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int LPR, int UNROLL, bool CODED = false>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_chunked_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));      // CODED: cap bytes of column codes
    __shared__ A red[BLOCK / kWave];
    __shared__ int sdict[CODED ? BLOCK : 1];
    constexpr int RC = BLOCK / LPR;                 // rows per chunk
    const int t = threadIdx.x, j = t / LPR, l = t % LPR;
    if constexpr (CODED) sdict[t] = a.dict[t];      // visible after the first staging barrier
    const int rb = rowblock_of(blockIdx.x, a.row_blocks, a.cycle);
    if (rb < 0) return;
    A dot1 = vzero<A>();
    for (int c = 0; c < LPR; ++c) {
        const int c0 = rb * BLOCK + c * RC;
        if (c0 >= a.n) break;                       // block-uniform
        const int row = c0 + j;
        const int rclamp = min(row, a.n - 1);
        const int s_raw = a.ptr[rclamp], e_raw = a.ptr[rclamp + 1];
        const int p0 = a.ptr[c0], p1 = a.ptr[min(c0 + RC, a.n)];
        const int cfirst = p0 & ~3;
        if (c) __syncthreads();                     // the previous chunk's walk is over before LDS is overwritten
        stage_slice<T, BLOCK, NT, CODED ? -3 : -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc, a.codes);
        const int s = s_raw - cfirst, e = (row < a.n) ? e_raw - cfirst : s_raw - cfirst;
        __syncthreads();
        T sum = vzero<T>();
        for (int k = s + l; k < e; k += UNROLL * LPR) {
            T xv[UNROLL], av[UNROLL];
            int cj[UNROLL];
            const int last = k + ((e - 1 - k) / LPR) * LPR;      // this lane's last valid entry
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int idx = min(k + u * LPR, last);
                if constexpr (CODED) cj[u] = reinterpret_cast<const unsigned char *>(sc)[idx];
                else cj[u] = sc[idx];
                av[u] = sv[idx];
            }
            if constexpr (CODED) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) cj[u] = row + sdict[cj[u]];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) xv[u] = a.x[cj[u]];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const T nxt = vfma(av[u], xv[u], sum);
                sum = vsel(k + u * LPR < e, nxt, sum);
            }
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) {
            if constexpr (VT<T>::cplx) {
                sum.x += __shfl_xor(sum.x, off, kWave);
                sum.y += __shfl_xor(sum.y, off, kWave);
            } else {
                sum += __shfl_xor(sum, off, kWave);
            }
        }
        if (l == 0 && row < a.n) {
            a.y[row] = sum;
            if (FUSE_DOT) dot1 = vadd(dot1, to_acc(vmul(a.dvec[row], sum)));
        }
    }
    if (FUSE_DOT) {
        const A tot = block_sum<BLOCK>(dot1, red);
        if (t == 0) a.partials[rb] = tot;
    }
}

// -------------------------------------------------------------------------------------------------
// SpMM fast path (nRHS > 1, RHS-major vectors as the reference ABI defines them: element i of RHS r at
// i + r*ld).  Same structure as spmv_rowblock_kernel -- the block's matrix slice goes through LDS ONCE -- and
// lane t then walks row t for RB right-hand sides at a time, keeping RB row sums in registers:
//     sum[j] += a_k * x[col_k + (r0+j)*ld]
// For every (k, j) the 64 lanes of a wave gather 64 consecutive rows' k-th column of RHS r0+j: coalesced
// exactly like the single-RHS kernel, with RB (x2 unrolled) independent gathers in flight per lane.  The
// matrix is read from HBM once per SpMM however many right-hand sides there are (the reference re-reads it
// per RHS through L2 at best: spmv.cl:23-26 loops r inside j).
// Fused d.q: per RHS the 64 lane contributions are summed by shuffles and parked per wave in LDS; after the
// last group, thread r adds the 4 wave sums of RHS r -> partials[r*row_blocks + rb].
// -------------------------------------------------------------------------------------------------
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int RB>
__global__ __launch_bounds__(BLOCK) void spmm_rowblock_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);                                   // [cap]
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));   // [cap]
    A *wavedot = reinterpret_cast<A *>(dyn_smem + (size_t)a.cap * (sizeof(T) + 4));   // [nrhs][BLOCK/64]

    const int t = threadIdx.x;
    const int rb = rowblock_of(blockIdx.x, a.row_blocks, a.cycle);
    if (rb < 0) return;
    const int r0 = rb * BLOCK;
    const int row = r0 + t;
    // The work-group's lifetime is a chain of dependent memory round trips; keep it at three: {row pointers}
    // -> {matrix slice} -> {x gather}.  The per-row pointers are loaded here, branch-free (clamped row), together
    // with the slice bounds, and only consumed after the barrier.
    const int rclamp = min(row, a.n - 1);
    const int s_raw = a.ptr[rclamp], e_raw = a.ptr[rclamp + 1];
    const int p0 = a.ptr[r0], p1 = a.ptr[min(r0 + BLOCK, a.n)];
    const int cfirst = p0 & ~3;
    stage_slice<T, BLOCK, NT, -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc);
    const int s = s_raw - cfirst, e = (row < a.n) ? e_raw - cfirst : s_raw - cfirst;
    __syncthreads();
    const int wave = t / kWave, lane = t & (kWave - 1);
    for (int g0 = 0; g0 < a.nrhs; g0 += RB) {
        T sum[RB];
        long long joff[RB];     // out-of-range right-hand sides of the last group alias the group's first one
#pragma unroll
        for (int j = 0; j < RB; ++j) { sum[j] = vzero<T>(); joff[j] = (g0 + j < a.nrhs) ? (long long)j * a.ldx : 0; }
        const T *xg = a.x + (long long)g0 * a.ldx;
        for (int k = s; k < e; k += 2) {         // two entries x RB right-hand sides in flight, branch-free
            const int i0 = k, i1 = min(k + 1, e - 1);
            const T a0 = sv[i0], a1 = sv[i1];
            const int c0 = sc[i0], c1 = sc[i1];
            T x0[RB], x1[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) { x0[j] = xg[c0 + joff[j]]; x1[j] = xg[c1 + joff[j]]; }
            const bool two = k + 1 < e;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                sum[j] = vfma(a0, x0[j], sum[j]);
                const T nxt = vfma(a1, x1[j], sum[j]);
                sum[j] = vsel(two, nxt, sum[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (g0 + j < a.nrhs) {       // wave-uniform
                if (row < a.n) a.y[row + (long long)(g0 + j) * a.ldy] = sum[j];
                if (FUSE_DOT) {
                    const A contrib = (row < a.n) ? to_acc(vmul(a.dvec[row + (long long)(g0 + j) * a.ldx], sum[j])) : vzero<A>();
                    const A w = wave_sum(contrib);
                    if (lane == 0) wavedot[(g0 + j) * (BLOCK / kWave) + wave] = w;
                }
            }
        }
    }
    if (FUSE_DOT) {
        __syncthreads();
        for (int r = t; r < a.nrhs; r += BLOCK) {
            A tot = wavedot[r * (BLOCK / kWave)];
#pragma unroll
            for (int w = 1; w < BLOCK / kWave; ++w) tot = vadd(tot, wavedot[r * (BLOCK / kWave) + w]);
            a.partials[(long long)r * a.row_blocks + rb] = tot;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Two-launch iteration for small systems (at most kFoldAlphaMax d.q partials per RHS), where launches, not bytes, are
// what an iteration costs (the reference's own sub-domain shape, 16k rows x 9 right-hand sides, is pure launch latency):
//   launch 1  spmv_fused_kernel   beta from the r.r partials of the PREVIOUS iteration (every work-group, same fixed order
//                                 -> bit-identical), d_new = beta d_old + r for its own rows, q = A d_new with
//                                 d_new[col] = beta d_old[col] + r[col] recomputed for every gathered column (vaypx: the
//                                 same bits the stored d_new holds), d_new.q partials.  Work-group 0 records delta,
//                                 beta and history[iter] of the previous iteration.
//   launch 2  axpy2_dot_alpha     alpha from the d.q partials, x += alpha d_new, r -= alpha q, r.r partials, iter += 1
// i.e. the reference's aypx (clcg.c:415) moves to the head of the NEXT iteration's SpMV launch, where it costs a second
// gather (L2 hits at these sizes) instead of a launch.  d_old and d_new are different buffers (ping-pong): a work-group
// may not overwrite entries of d its neighbours still gather.  The first iteration runs with beta = 0 (d_1 = r_0).
// cg_tail_kernel (same summation order) records delta / beta / history of the LAST iteration of an iterate() call.
// -------------------------------------------------------------------------------------------------
template <typename T> struct FusedArgs {
    const T *r;                     // residual (RHS-major like x)
    T *dnew;                        // search direction of this iteration (x = previous one)
    const typename VT<T>::acc *part_rr;
    int P;                          // r.r partials per RHS
    T *delta, *beta, *history;
    int history_cap;
    const int *iter;
};

template <typename T, int BLOCK, bool NT, int UNROLL>
__global__ __launch_bounds__(BLOCK) void spmv_fused_kernel(SpmvArgs<T> a, FusedArgs<T> f) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int t = threadIdx.x, rhs = blockIdx.y;              // one right-hand side per work-group ("wide" form)
    const int rb = rowblock_of(blockIdx.x, a.row_blocks, a.cycle);
    if (rb < 0) return;
    const int r0 = rb * BLOCK, row = r0 + t;
    const T *dr = a.x + (long long)rhs * a.ldx, *rr = f.r + (long long)rhs * a.ldx;
    const int rclamp = min(row, a.n - 1);
    const int s_raw = a.ptr[rclamp], e_raw = a.ptr[rclamp + 1];
    const int p0 = a.ptr[r0], p1 = a.ptr[min(r0 + BLOCK, a.n)];
    const int cfirst = p0 & ~3;
    const T d_own = dr[rclamp], r_own = rr[rclamp];
    stage_slice<T, BLOCK, NT, -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc);       // slice loads in flight behind the prologue
    {   // beta of this right-hand side: fixed order (thread-strided, wave tree, 4 wave sums) = aypx_beta_kernel's
        const int it = *f.iter;
        A acc = vzero<A>();
        if (it > 0) {
            const A *p = f.part_rr + (long long)rhs * f.P;
            for (int i = t; i < f.P; i += BLOCK) acc = vadd(acc, p[i]);
        }
        const A tot = block_sum<BLOCK>(acc, red);
        if (t == 0) {
            T b = vzero<T>();
            if (it > 0) {
                const T dnT = from_acc<T>(tot);
                const T dold = f.history[(long long)(it - 1) * a.nrhs + rhs];
                b = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
                if (blockIdx.x == 0) {
                    f.beta[rhs] = b;
                    f.delta[rhs] = dnT;
                    if (it < f.history_cap) f.history[(long long)it * a.nrhs + rhs] = dnT;
                }
            }
            beta_s = b;
        }
        __syncthreads();                                                            // also the barrier the staged slice needs
    }
    const T bt = beta_s;
    const int s = s_raw - cfirst, e = (row < a.n) ? e_raw - cfirst : s_raw - cfirst;
    T sum = vzero<T>();
    for (int k = s; k < e; k += UNROLL) {
        T dv[UNROLL], rv[UNROLL], av[UNROLL];
        int cj[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int idx = min(k + j, e - 1);
            cj[j] = sc[idx];
            av[j] = sv[idx];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) { dv[j] = dr[cj[j]]; rv[j] = rr[cj[j]]; }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const T nxt = vfma(av[j], vaypx(bt, dv[j], rv[j]), sum);
            sum = vsel(k + j < e, nxt, sum);
        }
    }
    A dot1 = vzero<A>();
    if (row < a.n) {
        const T dn = vaypx(bt, d_own, r_own);
        f.dnew[row + (long long)rhs * a.ldx] = dn;
        a.y[row + (long long)rhs * a.ldy] = sum;
        dot1 = to_acc(vmul(dn, sum));
    }
    const A tot = block_sum<BLOCK>(dot1, red);
    if (t == 0) a.partials[(long long)rhs * a.row_blocks + rb] = tot;
}

// delta / beta / history of the iteration whose r.r partials are on the device (end of an iterate() call of the two-launch
// loop): exactly what work-group 0 of the next spmv_fused launch would record, in the same summation order
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void cg_tail_kernel(FusedArgs<T> f, int nrhs) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    const int it = *f.iter, r = blockIdx.x;
    if (it <= 0) return;
    A acc = vzero<A>();
    const A *p = f.part_rr + (long long)r * f.P;
    for (int i = threadIdx.x; i < f.P; i += BLOCK) acc = vadd(acc, p[i]);
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) {
        const T dnT = from_acc<T>(tot);
        const T dold = f.history[(long long)(it - 1) * nrhs + r];
        f.beta[r] = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
        f.delta[r] = dnT;
        if (it < f.history_cap) f.history[(long long)it * nrhs + r] = dnT;
    }
}

// [rows][cols] -> [cols][rows]: RHS-major (the reference ABI, nRHS x N) <-> row-major (N x nRHS)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const T *__restrict__ in, T *__restrict__ out) {
    __shared__ T tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int k = ty; k < 32; k += 8)
        if (by + k < rows && bx + tx < cols) tile[k][tx] = in[(long long)(by + k) * cols + bx + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (bx + k < cols && by + tx < rows) out[(long long)(bx + k) * rows + by + tx] = tile[tx][k];
}

// CSR sanity of a device-resident matrix (CGAMD_MATRIX_ON_DEVICE, cgamd_dist_create): the row-block kernels size LDS
// from pointer differences and gather x[col] directly, so a bad index from a caller must become CGAMD_ERR_INVALID, not an
// out-of-bounds access.  flag bits: 1 ptr[0] != 0, 2 not monotone, 4 ptr[n] != nnz, 8 column out of [0, ncols),
// 16 index list entry out of [0, bound).
__global__ void csr_validate_kernel(int n, long long nnz, int ncols, const int *__restrict__ ptr, const int *__restrict__ cols,
                                    const int *__restrict__ index, int n_index, int index_bound, int *flag) {
    int bad = 0;
    const long long stride = (long long)gridDim.x * blockDim.x, i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 == 0) bad |= (ptr[0] != 0 ? 1 : 0) | (ptr[n] != nnz ? 4 : 0);
    for (long long i = i0; i < n; i += stride) bad |= ptr[i + 1] < ptr[i] ? 2 : 0;
    for (long long j = i0; j < nnz; j += stride) bad |= (cols[j] < 0 || cols[j] >= ncols) ? 8 : 0;
    for (long long k = i0; k < n_index; k += stride) bad |= (index[k] < 0 || index[k] >= index_bound) ? 16 : 0;
    if (bad) atomicOr(flag, bad);
}

// flag[rb] = 1 if any entry of row block rb references a halo column (col >= n_local): the blocks that must wait
// for the boundary exchange in the row-partitioned loop
template <int BLOCK> __global__ void rowblock_halo_flag_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                                int n_local, int row_blocks, int *flag) {
    const int rb = blockIdx.x;
    if (rb >= row_blocks) return;
    const int p0 = ptr[rb * BLOCK], p1 = ptr[min(rb * BLOCK + BLOCK, n)];
    int any = 0;
    for (int j = p0 + (int)threadIdx.x; j < p1; j += blockDim.x) any |= cols[j] >= n_local;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flag[rb] = any ? 1 : 0;
}

// largest (4-aligned) non-zero span of any BLOCK-row slice: decides whether the fast path applies
template <int BLOCK> __global__ void spmv_span_kernel(int n, const int *__restrict__ ptr, int row_blocks, int *out) {
    // out[0]: span of BLOCK-row slices; out[1..3]: of BLOCK/2, BLOCK/4, BLOCK/8-row slices (chunked row-block kernel);
    // out[4]: most non-zeros in 4 consecutive rows starting at a multiple of 4 (row-major SpMM); out[5]: longest row
    int m[4] = {0, 0, 0, 0}, mq = 0, mr = 0;
    for (int rb = blockIdx.x * blockDim.x + threadIdx.x; rb < row_blocks; rb += gridDim.x * blockDim.x) {
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {
            const int rows = BLOCK >> lv;
            for (int c = 0; c < (1 << lv); ++c) {
                const int ra = rb * BLOCK + c * rows;
                if (ra >= n) break;
                const int p0 = ptr[ra], p1 = ptr[min(ra + rows, n)];
                m[lv] = max(m[lv], p1 - (p0 & ~3));
            }
        }
        for (int ra = rb * BLOCK; ra < min(rb * BLOCK + BLOCK, n); ra += 4) mq = max(mq, ptr[min(ra + 4, n)] - ptr[ra]);
        for (int ra = rb * BLOCK; ra < min(rb * BLOCK + BLOCK, n); ++ra) mr = max(mr, ptr[ra + 1] - ptr[ra]);
    }
#pragma unroll
    for (int lv = 0; lv < 4; ++lv)
        if (m[lv] > 0) atomicMax(out + lv, m[lv]);
    if (mq > 0) atomicMax(out + 4, mq);
    if (mr > 0) atomicMax(out + 5, mr);
}

// =================================================================================================
// Streaming vector kernels.  grid = (G, nRHS); RHS r lives at base + r*ld.
// =================================================================================================
// x += alpha d ; r -= alpha q ; partial(r.r)
template <typename T, int BLOCK, bool VEC, int VNT>
CG_DEV void axpy2_dot_body(int n, const T *__restrict__ d, T *__restrict__ x, const T *__restrict__ q, T *__restrict__ rv,
                           long long ld, T al, typename VT<T>::acc *__restrict__ partials, typename VT<T>::acc *red) {
    using A = typename VT<T>::acc;
    const int r = blockIdx.y;
    const long long off = (long long)r * ld;
    d += off; x += off; q += off; rv += off;
    A acc = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pd = ld_pack(d + i * E), pq = (VNT & 2) ? ld_pack_nt(q + i * E) : ld_pack(q + i * E);
            Pack<T> px = (VNT & 1) ? ld_pack_nt(x + i * E) : ld_pack(x + i * E), pr = ld_pack(rv + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                px.v[k] = vadd(px.v[k], vmul(al, pd.v[k]));
                pr.v[k] = vsub(pr.v[k], vmul(al, pq.v[k]));
                acc = vadd(acc, to_acc(vmul(pr.v[k], pr.v[k])));
            }
            if (VNT & 1) st_pack_nt(x + i * E, px); else st_pack(x + i * E, px);
            st_pack(rv + i * E, pr);
        }
        i0 += npack * E;  // scalar tail
    }
    for (long long i = i0; i < n; i += stride) {
        x[i] = vadd(x[i], vmul(al, d[i]));
        const T rn = vsub(rv[i], vmul(al, q[i]));
        rv[i] = rn;
        acc = vadd(acc, to_acc(vmul(rn, rn)));
    }
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) partials[(long long)r * gridDim.x + blockIdx.x] = tot;
}

template <typename T, int BLOCK, bool VEC, int VNT = 0>
__global__ __launch_bounds__(BLOCK) void axpy2_dot_kernel(int n, const T *__restrict__ d, T *__restrict__ x,
                                                          const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                          const T *__restrict__ alpha,
                                                          typename VT<T>::acc *__restrict__ partials) {
    __shared__ typename VT<T>::acc red[BLOCK / kWave];
    axpy2_dot_body<T, BLOCK, VEC, VNT>(n, d, x, q, rv, ld, alpha[blockIdx.y], partials, red);
}

// Three-launch iteration for small systems (at most kFoldAlphaMax d.q partials per RHS): alpha is computed in the
// prologue of this launch -- every work-group adds the SpMV's d.q partials in the same fixed order, so all hold the
// bit-identical alpha = delta / d.q (clcg.c:317-327); work-group 0 records alpha and advances the iteration counter
// (nothing else in this launch reads either).  Saves the cg_alpha launch: 18.8 -> ~14 us per iteration at 250k rows.
constexpr int kFoldAlphaMax = 2048;     // N <= 524k rows; beyond, the separate cg_alpha launch is cheaper than every work-group summing
template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void axpy2_dot_alpha_kernel(int n, const T *__restrict__ d, T *__restrict__ x,
                                                                const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                                const typename VT<T>::acc *__restrict__ part_dq, int P,
                                                                const T *__restrict__ delta, T *alpha, int *iter,
                                                                typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T alpha_s;
    const int r = blockIdx.y;
    {
        A acc = vzero<A>();
        const A *p = part_dq + (long long)r * P;
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, p[i]);
        const A dq = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const T dqT = from_acc<T>(dq);      // the reference rounds d.q to the value type before dividing (clcg.c:318-327)
            const T al = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
            alpha_s = al;
            if (blockIdx.x == 0) {
                alpha[r] = al;
                if (r == 0) *iter = *iter + 1;
            }
        }
        __syncthreads();
    }
    axpy2_dot_body<T, BLOCK, VEC, 0>(n, d, x, q, rv, ld, alpha_s, partials, red);
}

// ---- ten-vector-pass iteration -------------------------------------------------------------------------------
// x is read by nothing inside the loop, so x += alpha d may ride in the aypx launch, which reads d anyway:
//   axpy_dot_kernel      r -= alpha q, partials of r.r                      reads q, r    writes r      3 NV
//   aypx_beta_x_kernel   beta in the prologue; x += alpha d; d = beta d + r reads r, d, x writes d, x   5 NV
// instead of 6 NV + 3 NV: d is read once per iteration, not twice (fused minimum 10 NV + SpMV).  Every element sees the
// same operations in the same order as before, so x is bit-identical.
template <typename T, int BLOCK, bool VEC, int VNT>
CG_DEV void axpy_dot_body(int n, const T *__restrict__ q, T *__restrict__ rv, long long ld, T al,
                          typename VT<T>::acc *__restrict__ partials, typename VT<T>::acc *red) {
    using A = typename VT<T>::acc;
    const int r = blockIdx.y;
    q += (long long)r * ld; rv += (long long)r * ld;
    A acc = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pq = (VNT & 2) ? ld_pack_nt(q + i * E) : ld_pack(q + i * E);
            Pack<T> pr = ld_pack(rv + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                pr.v[k] = vsub(pr.v[k], vmul(al, pq.v[k]));
                acc = vadd(acc, to_acc(vmul(pr.v[k], pr.v[k])));
            }
            st_pack(rv + i * E, pr);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T rn = vsub(rv[i], vmul(al, q[i]));
        rv[i] = rn;
        acc = vadd(acc, to_acc(vmul(rn, rn)));
    }
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) partials[(long long)r * gridDim.x + blockIdx.x] = tot;
}
template <typename T, int BLOCK, bool VEC, int VNT = 0>
__global__ __launch_bounds__(BLOCK) void axpy_dot_kernel(int n, const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                         const T *__restrict__ alpha, typename VT<T>::acc *__restrict__ partials) {
    __shared__ typename VT<T>::acc red[BLOCK / kWave];
    axpy_dot_body<T, BLOCK, VEC, VNT>(n, q, rv, ld, alpha[blockIdx.y], partials, red);
}
// small systems: alpha in the prologue (see axpy2_dot_alpha_kernel)
template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void axpy_dot_alpha_kernel(int n, const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                               const typename VT<T>::acc *__restrict__ part_dq, int P,
                                                               const T *__restrict__ delta, T *alpha, int *iter,
                                                               typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T alpha_s;
    const int r = blockIdx.y;
    {
        A acc = vzero<A>();
        const A *p = part_dq + (long long)r * P;
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, p[i]);
        const A dq = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const T dqT = from_acc<T>(dq);
            const T al = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
            alpha_s = al;
            if (blockIdx.x == 0) {
                alpha[r] = al;
                if (r == 0) *iter = *iter + 1;
            }
        }
        __syncthreads();
    }
    axpy_dot_body<T, BLOCK, VEC, 0>(n, q, rv, ld, alpha_s, partials, red);
}

template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void dot_partials_kernel(int n, const T *__restrict__ a, const T *__restrict__ b,
                                                             long long ld, typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    const int r = blockIdx.y;
    a += (long long)r * ld; b += (long long)r * ld;
    A acc = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pa = ld_pack(a + i * E), pb = ld_pack(b + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) acc = vadd(acc, to_acc(vmul(pa.v[k], pb.v[k])));
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) acc = vadd(acc, to_acc(vmul(a[i], b[i])));
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) partials[(long long)r * gridDim.x + blockIdx.x] = tot;
}

// OP 0: y += a x   1: y -= a x   2: y = a y + x   3: res(y) = x - b
template <typename T, int BLOCK, bool VEC, int OP>
__global__ __launch_bounds__(BLOCK) void ewise_kernel(int n, const T *__restrict__ x, T *__restrict__ y,
                                                      const T *__restrict__ b2, long long ld,
                                                      const T *__restrict__ alpha) {
    const int r = blockIdx.y;
    x += (long long)r * ld; y += (long long)r * ld;
    if (OP == 3) b2 += (long long)r * ld;
    const T al = (OP == 3) ? vzero<T>() : alpha[r];
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    auto f = [&](T xv, T yv, T bv) -> T {
        if (OP == 0) return vadd(yv, vmul(al, xv));
        if (OP == 1) return vsub(yv, vmul(al, xv));
        if (OP == 2) return vaypx(al, yv, xv);
        return vsub(xv, bv);
    };
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py, pb;
            if (OP != 3) py = ld_pack(y + i * E);
            if (OP == 3) pb = ld_pack(b2 + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) py.v[k] = f(px.v[k], OP != 3 ? py.v[k] : vzero<T>(), OP == 3 ? pb.v[k] : vzero<T>());
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride)
        y[i] = f(x[i], OP != 3 ? y[i] : vzero<T>(), OP == 3 ? b2[i] : vzero<T>());
}

// d = beta d + r with beta computed in the prologue (replaces the cg_beta launch of the 5-launch loop):
// every work-group adds the P partials of r.r in the same fixed order (thread-strided, wave tree, 4 wave
// sums), so all of them hold the bit-identical delta_new and beta = delta_new / delta_old
// (clcg.c:376-391); delta_old is history[iter-1] -- nothing in this launch writes that entry, work-group 0
// alone writes delta/beta/history[iter].  The iteration counter was advanced by cg_alpha.
template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void aypx_beta_kernel(int n, const T *__restrict__ x, T *__restrict__ y, long long ld,
                                                          const typename VT<T>::acc *__restrict__ partials, int P,
                                                          int nrhs, T *delta, T *beta, T *history, int history_cap, const int *iter) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int r = blockIdx.y;
    {
        A acc = vzero<A>();
        const A *p = partials + (long long)r * P;
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, p[i]);
        const A tot = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const int it = *iter;
            const T dnT = from_acc<T>(tot);
            const T dold = history[(long long)(it - 1) * nrhs + r];
            const T b = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
            beta_s = b;
            if (blockIdx.x == 0) {
                beta[r] = b;
                delta[r] = dnT;
                if (it < history_cap) history[(long long)it * nrhs + r] = dnT;
            }
        }
        __syncthreads();
    }
    const T al = beta_s;
    x += (long long)r * ld; y += (long long)r * ld;
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py = ld_pack(y + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) py.v[k] = vaypx(al, py.v[k], px.v[k]);
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) y[i] = vaypx(al, y[i], x[i]);
}

// the same with the deferred x += alpha d (ten-vector-pass iteration): xs = solution vector, alpha of THIS iteration
template <typename T, int BLOCK, bool VEC, int VNT>
__global__ __launch_bounds__(BLOCK) void aypx_beta_x_kernel(int n, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ xs,
                                                            long long ld, const typename VT<T>::acc *__restrict__ partials, int P,
                                                            int nrhs, const T *__restrict__ alpha, T *delta, T *beta, T *history,
                                                            int history_cap, const int *iter) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int r = blockIdx.y;
    {
        A acc = vzero<A>();
        const A *p = partials + (long long)r * P;
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, p[i]);
        const A tot = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const int it = *iter;
            const T dnT = from_acc<T>(tot);
            const T dold = history[(long long)(it - 1) * nrhs + r];
            const T b = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
            beta_s = b;
            if (blockIdx.x == 0) {
                beta[r] = b;
                delta[r] = dnT;
                if (it < history_cap) history[(long long)it * nrhs + r] = dnT;
            }
        }
        __syncthreads();
    }
    const T bt = beta_s, al = alpha[r];
    x += (long long)r * ld; y += (long long)r * ld; xs += (long long)r * ld;
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py = ld_pack(y + i * E);
            Pack<T> ps = (VNT & 1) ? ld_pack_nt(xs + i * E) : ld_pack(xs + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                ps.v[k] = vadd(ps.v[k], vmul(al, py.v[k]));
                py.v[k] = vaypx(bt, py.v[k], px.v[k]);
            }
            if (VNT & 1) st_pack_nt(xs + i * E, ps); else st_pack(xs + i * E, ps);
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T dv = y[i];
        xs[i] = vadd(xs[i], vmul(al, dv));
        y[i] = vaypx(bt, dv, x[i]);
    }
}

// =================================================================================================
// Diagonally (Jacobi) preconditioned CG -- the reference's PCG with a diagonal CSR `M`, z = M.dot(r)
// (helmFE_var.py:546-586): rho = r.z, p = z + (rho/rho_old) p, q = A p, alpha = rho / p.q, x += alpha p, r -= alpha q,
// stop on sqrt|r.r|.  Same four launches as the plain loop: the SpMV (+p.q) and cg_alpha are shared (delta holds rho);
//   pcg_axpy2_dot2_kernel : r -= alpha q, partials of r.(m r) and of r.r                    (4NV bytes)
//   pcg_aypx_beta_kernel  : beta in the prologue, x += alpha p, p = m r + beta p            (6NV bytes)
// (x += alpha p rides in the second launch, which reads p anyway: see the ten-vector-pass iteration above)
// m[i] is what multiplies r[i] (the inverse diagonal for Jacobi), shared by all right-hand sides.  rho of the previous
// iteration is read from a two-entry parity buffer so that work-group 0 may publish the new one in the same launch.
// =================================================================================================
template <typename T, int BLOCK, bool VEC, bool INIT>
__global__ __launch_bounds__(BLOCK) void pcg_axpy2_dot2_kernel(int n, const T *__restrict__ d, T *__restrict__ x,
                                                               const T *__restrict__ q, T *__restrict__ rv,
                                                               const T *__restrict__ m, long long ld,
                                                               const T *__restrict__ alpha,
                                                               typename VT<T>::acc *__restrict__ part_rz,
                                                               typename VT<T>::acc *__restrict__ part_rr) {
    // INIT: no update, d = m r instead (set_rhs: p0 = z0), same two dot products
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    const int r = blockIdx.y;
    T *dw = const_cast<T *>(d) + (long long)r * ld;
    d += (long long)r * ld; x += (long long)r * ld; q += (long long)r * ld; rv += (long long)r * ld;
    const T al = INIT ? vzero<T>() : alpha[r];
    A arz = vzero<A>(), arr = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            Pack<T> pr = ld_pack(rv + i * E);
            const Pack<T> pm = ld_pack(m + i * E);
            if (!INIT) {
                const Pack<T> pq = ld_pack(q + i * E);
#pragma unroll
                for (int k = 0; k < E; ++k) pr.v[k] = vsub(pr.v[k], vmul(al, pq.v[k]));
                st_pack(rv + i * E, pr);
            }
            Pack<T> pz;
#pragma unroll
            for (int k = 0; k < E; ++k) {
                pz.v[k] = vmul(pm.v[k], pr.v[k]);
                arz = vadd(arz, to_acc(vmul(pr.v[k], pz.v[k])));
                arr = vadd(arr, to_acc(vmul(pr.v[k], pr.v[k])));
            }
            if (INIT) st_pack(dw + i * E, pz);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        T rn = rv[i];
        if (!INIT) {
            rn = vsub(rn, vmul(al, q[i]));
            rv[i] = rn;
        }
        const T z = vmul(m[i], rn);
        if (INIT) dw[i] = z;
        arz = vadd(arz, to_acc(vmul(rn, z)));
        arr = vadd(arr, to_acc(vmul(rn, rn)));
    }
    const A trz = block_sum<BLOCK>(arz, red);
    if (threadIdx.x == 0) part_rz[(long long)r * gridDim.x + blockIdx.x] = trz;
    const A trr = block_sum<BLOCK>(arr, red);
    if (threadIdx.x == 0) part_rr[(long long)r * gridDim.x + blockIdx.x] = trr;
}

template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void pcg_aypx_beta_kernel(int n, const T *__restrict__ rv, T *__restrict__ pv,
                                                              const T *__restrict__ m, long long ld,
                                                              const typename VT<T>::acc *__restrict__ part_rz,
                                                              const typename VT<T>::acc *__restrict__ part_rr, int P, int nrhs,
                                                              T *delta, T *beta, T *history, int history_cap, T *rho2, const int *iter,
                                                              T *__restrict__ xs, const T *__restrict__ alpha) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int r = blockIdx.y;
    {
        A acc = vzero<A>();
        const A *pz = part_rz + (long long)r * P;
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, pz[i]);
        const A rho = block_sum<BLOCK>(acc, red);
        A acc2 = vzero<A>();
        if (blockIdx.x == 0) {
            const A *pr = part_rr + (long long)r * P;
            for (int i = threadIdx.x; i < P; i += BLOCK) acc2 = vadd(acc2, pr[i]);
            acc2 = block_sum<BLOCK>(acc2, red);
        }
        if (threadIdx.x == 0) {
            const int it = *iter;
            const T rhoT = from_acc<T>(rho);
            const T rold = rho2[(long long)((it - 1) & 1) * nrhs + r];
            const T b = from_acc<T>(acc_div(to_acc(rhoT), to_acc(rold)));
            beta_s = b;
            if (blockIdx.x == 0) {
                beta[r] = b;
                delta[r] = rhoT;                                   // cg_alpha divides this by p.q
                rho2[(long long)(it & 1) * nrhs + r] = rhoT;
                if (it < history_cap) history[(long long)it * nrhs + r] = from_acc<T>(acc2);   // r.r: what the stopping test looks at
            }
        }
        __syncthreads();
    }
    const T bt = beta_s, al = alpha[r];
    rv += (long long)r * ld; pv += (long long)r * ld; xs += (long long)r * ld;
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pr = ld_pack(rv + i * E), pm = ld_pack(m + i * E);
            Pack<T> pp = ld_pack(pv + i * E), px = ld_pack(xs + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                px.v[k] = vadd(px.v[k], vmul(al, pp.v[k]));
                pp.v[k] = vadd(vmul(bt, pp.v[k]), vmul(pm.v[k], pr.v[k]));
            }
            st_pack(xs + i * E, px);
            st_pack(pv + i * E, pp);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T pv0 = pv[i];
        xs[i] = vadd(xs[i], vmul(al, pv0));
        pv[i] = vadd(vmul(bt, pv0), vmul(m[i], rv[i]));
    }
}

// set_rhs: delta = rho0 = sum r.z partials, rho2[0] = rho0, history[0] = r.r, iter = 0
template <typename T>
__global__ __launch_bounds__(1024) void pcg_delta0_kernel(const typename VT<T>::acc *part_rz, const typename VT<T>::acc *part_rr, int P,
                                                          int nrhs, T *delta, T *history, T *rho2, int *iter);

// =================================================================================================
// Scalar kernels: one 256-thread work-group per RHS; fixed summation order (thread-strided, wave
// shuffle, then the 4 wave sums in order) => bitwise reproducible.
// =================================================================================================
constexpr int kScalarBlock = 1024;
template <typename A> CG_DEV A sum_partials_block(const A *p, int grid, A *smem) {
    A acc = vzero<A>();
    int i = threadIdx.x;
    for (; i + 7 * kScalarBlock < grid; i += 8 * kScalarBlock) {   // 8 loads in flight; same summation order
        A v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * kScalarBlock];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = vadd(acc, v[k]);
    }
    for (; i < grid; i += kScalarBlock) acc = vadd(acc, p[i]);
    acc = block_sum<kScalarBlock>(acc, smem);
    __syncthreads();
    if (threadIdx.x == 0) smem[0] = acc;
    __syncthreads();
    return smem[0];
}

template <typename T>
__global__ __launch_bounds__(kScalarBlock) void reduce_to_value_kernel(const typename VT<T>::acc *partials, int grid, int nrhs,
                                                              T *result) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto s = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) result[r] = from_acc<T>(s);
}

template <typename T>
__global__ __launch_bounds__(1024) void pcg_delta0_kernel(const typename VT<T>::acc *part_rz, const typename VT<T>::acc *part_rr, int P,
                                                          int nrhs, T *delta, T *history, T *rho2, int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto rho = sum_partials_block(part_rz + (long long)r * P, P, smem);
    __syncthreads();
    const auto rr = sum_partials_block(part_rr + (long long)r * P, P, smem);
    if (threadIdx.x == 0) {
        delta[r] = from_acc<T>(rho);
        rho2[r] = from_acc<T>(rho);
        history[r] = from_acc<T>(rr);
        if (r == 0) *iter = 0;
    }
}

template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_delta0_kernel(const typename VT<T>::acc *partials, int grid, int nrhs, T *delta,
                                                        T *history, int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto s = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) {
        delta[r] = from_acc<T>(s);
        history[r] = from_acc<T>(s);
        if (r == 0) *iter = 0;
    }
}

// alpha[r] = delta[r] / (d.q)[r].  Block 0 also advances the iteration counter: the counter is only READ by
// the cg_beta kernel of the same iteration (a later launch), never inside this launch.
template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_alpha_kernel(const typename VT<T>::acc *partials, int grid, int nrhs,
                                                       const T *delta, T *alpha, int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto dq = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) {
        // the reference rounds dq to the value type before dividing (clcg.c:318-327)
        const T dqT = from_acc<T>(dq);
        alpha[r] = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
        if (r == 0) *iter = *iter + 1;
    }
}

// Two-level form for long partial arrays (one per row block: 39 063 at N=10M): kAlphaParts work-groups each sum one
// contiguous part (fixed order inside), the last one to finish (device ticket) adds the part sums in part order and does
// the scalar step -- the result does not depend on which work-group came last.  One work-group needed 5 rounds of 8 loads
// per thread (8 us); this needs one (4.5 us).
constexpr int kAlphaParts = 32;
template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_alpha2_kernel(const typename VT<T>::acc *partials, int grid, int nrhs,
                                                        const T *delta, T *alpha, int *iter,
                                                        typename VT<T>::acc *stage, unsigned *ticket) {
    using A = typename VT<T>::acc;
    __shared__ A smem[kScalarBlock / kWave];
    __shared__ bool last;
    const int r = blockIdx.y, part = blockIdx.x;
    const int per = (grid + kAlphaParts - 1) / kAlphaParts;
    const int lo = min(part * per, grid), hi = min(lo + per, grid);
    const A sum = sum_partials_block(partials + (long long)r * grid + lo, hi - lo, smem);
    if (threadIdx.x == 0) {
        __hip_atomic_store(reinterpret_cast<double *>(stage + (long long)r * kAlphaParts + part), to_acc2(sum).x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (VT<T>::cplx)
            __hip_atomic_store(reinterpret_cast<double *>(stage + (long long)r * kAlphaParts + part) + 1, to_acc2(sum).y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(ticket + r, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = prev + 1 == (unsigned)kAlphaParts;
        if (last) {
            __hip_atomic_store(ticket + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double2 tot = make_double2(0., 0.);
            for (int k = 0; k < kAlphaParts; ++k) {
                const double *p = reinterpret_cast<const double *>(stage + (long long)r * kAlphaParts + k);
                tot.x += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (VT<T>::cplx) tot.y += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const T dqT = from_acc<T>(from_acc2<A>(tot));
            alpha[r] = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
            if (r == 0) *iter = *iter + 1;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_beta_kernel(const typename VT<T>::acc *partials, int grid, int nrhs, T *delta,
                                                      T *beta, T *history, int history_cap, const int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto dn = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) {
        const int it = *iter;   // already advanced by cg_alpha of this iteration
        const T dnT = from_acc<T>(dn);
        beta[r] = from_acc<T>(acc_div(to_acc(dnT), to_acc(delta[r])));   // clcg.c:389-391
        delta[r] = dnT;
        if (it < history_cap) history[(long long)it * nrhs + r] = dnT;
    }
}

// partials -> one accumulator value per RHS (input of the RCCL all-reduce in the multi-GPU loop)
template <typename A>
__global__ __launch_bounds__(kScalarBlock) void reduce_to_acc_kernel(const A *partials, int grid, int nrhs, A *out) {
    __shared__ A smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const A s = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) out[r] = s;
}

// halo pack: out[k] = v[index[k]]  (boundary entries of d that neighbouring ranks gather in their SpMV)
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(int count, const int *__restrict__ index, const T *__restrict__ v,
                                                   T *__restrict__ out) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < count; k += gridDim.x * 256) out[k] = v[index[k]];
}

// =================================================================================================
// Peer-to-peer communication over xGMI without RCCL (optional backend of the row-partitioned loop).
// RCCL's latency (10-20 us per small collective) bounds strong scaling of a 40 us iteration; here each rank owns
// an UNCACHED, IPC-shared "mailbox" that its peers write directly:
//     [0,4096)      reduction slots  slot[which in 0..1][source rank] = {value.x, value.y, epoch, pad} (32 B)
//     [4096,6144)   halo flags       flag[source rank] = epoch of the last complete boundary push
//     [6144,8192)   error word
//     [8192,...)    halo entries     laid out exactly like the halo part of d_ext
// Hand-off protocol (placement independent).  Every mailbox access is a system-scope relaxed atomic load/store, i.e. a
// write-through / cache-bypassing access (sc0 sc1) to memory that is mapped uncached anyway.  Producer: payload stores ->
// every thread waits until its stores are acknowledged (p2p_stores_done: s_waitcnt vmcnt(0), NO cache maintenance) ->
// work-group barrier -> one store of the epoch.  Consumer: polls that word, then reads the payload with the same
// cache-bypassing loads.  Deliberately NOT system-scope release/acquire fences: on gfx950 those write back / invalidate
// the whole L2 of the XCD (buffer_wbl2 / buffer_inv sc0 sc1), and inside the SpMV and aypx launches that threw the
// vectors out of L2 once per pushing or waiting work-group (SpMV 21 -> 36 us on a 1.25M-row slab).
// Epochs come from device counters advanced by a later single-work-group kernel, so a hipGraph replays the protocol
// unchanged.  Slot reuse is safe because two full all-reduces separate consecutive uses of any slot or of the halo area.
// Spins are bounded; a timeout sets the error word and the kernels fall through.
// =================================================================================================
constexpr int kMbSlots = 0, kMbHaloFlags = 4096, kMbError = 6144, kMbHalo = 8192;
constexpr long long kSpinLimit = 1LL << 21;   // polls of ~1-2 us each: a few seconds, then the error word is set

// all of this lane's earlier stores are acknowledged by the memory system; compiler-level ordering included
CG_DEV void p2p_stores_done() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}
CG_DEV void st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
CG_DEV unsigned long long ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
CG_DEV void st_sys_val(float *p, float v) { __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
CG_DEV void st_sys_val(double *p, double v) { st_sys(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v)); }
CG_DEV void st_sys_val(float2 *p, float2 v) {
    unsigned long long w = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    st_sys(reinterpret_cast<unsigned long long *>(p), w);
}
CG_DEV void st_sys_val(double2 *p, double2 v) { st_sys_val(reinterpret_cast<double *>(p), v.x); st_sys_val(reinterpret_cast<double *>(p) + 1, v.y); }
CG_DEV float ld_sys_val(const float *p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }
CG_DEV double ld_sys_val(const double *p) { return __longlong_as_double((long long)ld_sys(reinterpret_cast<const unsigned long long *>(p))); }
CG_DEV float2 ld_sys_val(const float2 *p) {
    const unsigned long long w = ld_sys(reinterpret_cast<const unsigned long long *>(p));
    return make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
}
CG_DEV double2 ld_sys_val(const double2 *p) { return make_double2(ld_sys_val(reinterpret_cast<const double *>(p)), ld_sys_val(reinterpret_cast<const double *>(p) + 1)); }

// spin (one lane) until *word == want; false on timeout
CG_DEV bool spin_until(const unsigned long long *word, unsigned long long want, const char *my_mailbox) {
    // once any spin of this rank has timed out (error word set) later spins give up after one look: a broken
    // exchange then costs one time-out, not one per kernel
    const unsigned long long *err = reinterpret_cast<const unsigned long long *>(my_mailbox + kMbError);
    for (long long i = 0; i < kSpinLimit; ++i) {
        if (ld_sys(word) == want) return true;
        if ((i & 1023) == 1023 && ld_sys(err) != 0) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

struct P2pExchangeArgs {
    char *const *mailbox;       // [nranks] device array: mailbox base of every rank, mapped in this process
    int rank, n_peers, n_local;
    const int *peer_rank, *send_off, *send_count, *dst_off, *recv_off, *recv_count;   // device arrays [n_peers]
    const int *send_index;
    unsigned long long *epoch;  // device counter of boundary exchanges
    unsigned *counters;         // device: [0] unpack work-groups done, [1 + p] push work-groups done for peer p (all 0 between launches)
    int max_count;              // largest send/recv count over the peers (grid sizing)
};

// Push: blockIdx.y = peer, blockIdx.x = chunk of kP2pChunk entries.  Every work-group gathers its chunk of my boundary
// entries straight into the peer's mailbox (one 8/16-byte system-scope store per lane, all in flight together); the
// LAST work-group of a peer to finish (device counter) publishes the epoch flag with release semantics.
constexpr int kP2pBlock = 256, kP2pChunk = 1024;
// chunk c (of `chunks`) of my boundary entries for peer p; called by a whole kP2pBlock-thread work-group
template <typename T> CG_DEV void p2p_push_chunk(const P2pExchangeArgs &a, const T *v, int p, int c, int chunks, unsigned long long ep) {
    char *mb = a.mailbox[a.peer_rank[p]];
    T *dst = reinterpret_cast<T *>(mb + kMbHalo) + a.dst_off[p];
    const int *idx = a.send_index + a.send_off[p];
    const int cnt = a.send_count[p], k0 = c * kP2pChunk + threadIdx.x;
    if (c * kP2pChunk < cnt) {
        T val[kP2pChunk / kP2pBlock];
#pragma unroll
        for (int u = 0; u < kP2pChunk / kP2pBlock; ++u) val[u] = v[idx[min(k0 + u * kP2pBlock, cnt - 1)]];
#pragma unroll
        for (int u = 0; u < kP2pChunk / kP2pBlock; ++u) {
            const int k = k0 + u * kP2pBlock;
            if (k < cnt) st_sys_val(dst + k, val[u]);
        }
    }
    p2p_stores_done();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *done = a.counters + 1 + p;
        const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)prev + 1 == chunks) {      // every chunk's stores were acknowledged before its increment
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st_sys(reinterpret_cast<unsigned long long *>(mb + kMbHaloFlags) + a.rank, ep);
        }
    }
}
template <typename T> __global__ __launch_bounds__(kP2pBlock) void p2p_push_kernel(P2pExchangeArgs a, const T *v) {
    p2p_push_chunk<T>(a, v, blockIdx.y, blockIdx.x, gridDim.x, *a.epoch + 1);
}

// Unpack: same grid shape.  Every work-group waits for its peer's epoch (one polling lane), then copies its chunk of the
// landed entries into the halo part of v_ext; the last work-group overall advances the exchange epoch.
template <typename T> __global__ __launch_bounds__(kP2pBlock) void p2p_wait_unpack_kernel(P2pExchangeArgs a, T *v_ext) {
    const int p = blockIdx.y;
    const unsigned long long ep = *a.epoch + 1;
    char *mb = a.mailbox[a.rank];
    const int cnt = a.recv_count[p], k0 = blockIdx.x * kP2pChunk + threadIdx.x;
    if ((int)blockIdx.x * kP2pChunk < cnt) {
        if (threadIdx.x == 0) {
            if (!spin_until(reinterpret_cast<unsigned long long *>(mb + kMbHaloFlags) + a.peer_rank[p], ep, mb))
                st_sys(reinterpret_cast<unsigned long long *>(mb + kMbError), 1ULL);
        }
        __syncthreads();
        const T *src = reinterpret_cast<const T *>(mb + kMbHalo) + a.recv_off[p];
        T *dst = v_ext + a.n_local + a.recv_off[p];
        T val[kP2pChunk / kP2pBlock];
#pragma unroll
        for (int u = 0; u < kP2pChunk / kP2pBlock; ++u) val[u] = ld_sys_val(src + min(k0 + u * kP2pBlock, cnt - 1));
#pragma unroll
        for (int u = 0; u < kP2pChunk / kP2pBlock; ++u) {
            const int k = k0 + u * kP2pBlock;
            if (k < cnt) dst[k] = val[u];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(a.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == gridDim.x * gridDim.y) {       // everybody has read *a.epoch
            __hip_atomic_store(a.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.epoch = ep;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Peer-to-peer CG iteration in four launches (same count as the single-GPU loop):
//   spmv_rowblock_p2p_kernel : push + wait + SpMV + d.q partials     p2p_allreduce_kernel<2>: alpha (bumps the epochs)
//   axpy_dot_kernel          : r -= alpha q, r.r partials             aypx_beta_p2p_kernel  : all-reduce of r.r + beta,
//                                                                                            x += alpha d, d = beta d + r
// spmv_rowblock_p2p_kernel = spmv_rowblock_kernel (same block-cyclic XCD schedule, rotated so that the leading boundary
// row blocks of a slab partition are visited last) plus a per-row-block flag "references a halo column", where
//   * the first n_peers * push_chunks work-groups first ship one chunk of my boundary entries of d into the peers'
//     mailboxes (p2p_push_chunk), so the halo is on the wire before any SpMV work starts;
//   * the boundary row blocks wait for the peers' epoch flags after issuing their matrix slice loads and gather halo
//     columns (col >= n_local) straight from the mailbox; there is no unpack pass;
//   * interior row blocks never look at a flag: the xGMI latency hides behind them.
// Nobody writes the exchange epoch here; the alpha kernel that follows in stream order advances it.
// -------------------------------------------------------------------------------------------------
template <typename T> struct SpmvP2pArgs {
    SpmvArgs<T> s;          // rb_list = per-row-block flag (1 = references a halo column), cycle = block-cyclic schedule
    P2pExchangeArgs x;
    const T *halo;          // my mailbox's halo area: entry h is column n_local + h
    int n_local, rotate, push_chunks;   // rotate: row blocks are visited from this one on, so leading boundary blocks come last
};

template <typename T, int BLOCK, bool NT, int UNROLL, bool CODED = false>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_p2p_kernel(SpmvP2pArgs<T> g) {
    using A = typename VT<T>::acc;
    const SpmvArgs<T> &a = g.s;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));      // CODED: cap bytes of column codes
    __shared__ A red[BLOCK / kWave];
    __shared__ int sdict[CODED ? BLOCK : 1];
    const int t = threadIdx.x, b = blockIdx.x;
    if constexpr (CODED) sdict[t] = a.dict[t];
    // the epoch is loaded only where it is needed: a load here would sit in front of every work-group's first wait
    if (b < g.x.n_peers * g.push_chunks)
        p2p_push_chunk<T>(g.x, a.x, b / g.push_chunks, b % g.push_chunks, g.push_chunks, *g.x.epoch + 1);
    // the schedule is arithmetic (a list lookup here would put one more memory round trip in front of every work-group)
    int rb = rowblock_of(b, a.row_blocks, a.cycle);
    if (rb < 0) return;
    rb += g.rotate;
    if (rb >= a.row_blocks) rb -= a.row_blocks;
    const int bflag = a.rb_list[rb];          // consumed after the slice loads are in flight
    const int r0 = rb * BLOCK, row = r0 + t;
    const int rclamp = min(row, a.n - 1);
    const int s_raw = a.ptr[rclamp], e_raw = a.ptr[rclamp + 1];
    const int p0 = a.ptr[r0], p1 = a.ptr[min(r0 + BLOCK, a.n)];
    const int cfirst = p0 & ~3;
    stage_slice<T, BLOCK, NT, CODED ? -3 : -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc, a.codes);
    const int s = s_raw - cfirst, e = (row < a.n) ? e_raw - cfirst : s_raw - cfirst;
    const bool boundary = bflag != 0;
    if (boundary && t == 0) {
        const char *mb = g.x.mailbox[g.x.rank];
        const unsigned long long ep = *g.x.epoch + 1;
        for (int p = 0; p < g.x.n_peers; ++p) {
            if (g.x.recv_count[p] == 0) continue;
            if (!spin_until(reinterpret_cast<const unsigned long long *>(mb + kMbHaloFlags) + g.x.peer_rank[p], ep, mb))
                st_sys(reinterpret_cast<unsigned long long *>(const_cast<char *>(mb) + kMbError), 1ULL);
        }
    }
    __syncthreads();
    T sum = vzero<T>();
    if (boundary) {
        for (int k = s; k < e; k += UNROLL) {
            T xv[UNROLL], av[UNROLL];
            int cj[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const int idx = min(k + j, e - 1);
                if constexpr (CODED) cj[j] = reinterpret_cast<const unsigned char *>(sc)[idx];
                else cj[j] = sc[idx];
                av[j] = sv[idx];
            }
            if constexpr (CODED) {
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) cj[j] = row + sdict[cj[j]];
            }
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const bool far = cj[j] >= g.n_local;
                xv[j] = a.x[far ? rclamp : cj[j]];
                if (far) xv[j] = ld_sys_val(g.halo + (cj[j] - g.n_local));     // in place from the mailbox, cache-bypassing
            }
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const T nxt = vfma(av[j], xv[j], sum);
                sum = vsel(k + j < e, nxt, sum);
            }
        }
    } else {
        for (int k = s; k < e; k += UNROLL) {
            T xv[UNROLL], av[UNROLL];
            int cj[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const int idx = min(k + j, e - 1);
                if constexpr (CODED) cj[j] = reinterpret_cast<const unsigned char *>(sc)[idx];
                else cj[j] = sc[idx];
                av[j] = sv[idx];
            }
            if constexpr (CODED) {
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) cj[j] = row + sdict[cj[j]];
            }
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) xv[j] = a.x[cj[j]];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const T nxt = vfma(av[j], xv[j], sum);
                sum = vsel(k + j < e, nxt, sum);
            }
        }
    }
    A dot1 = vzero<A>();
    if (row < a.n) {
        a.y[row] = sum;
        dot1 = to_acc(vmul(a.dvec[row], sum));
    }
    const A tot = block_sum<BLOCK>(dot1, red);
    if (t == 0) a.partials[rb] = tot;
}

// d = beta d + r with the all-reduce of r.r and the beta step in the prologue.  Work-group 0 sums the local partials and
// writes the result into slot set `which` of every rank's mailbox; EVERY work-group then waits for all ranks' slots in
// its own mailbox and adds them in rank order (bitwise the same beta everywhere).  The epoch was advanced by the alpha
// kernel of this iteration, so it is read-only here; slots are safe to reuse because a peer can only publish its next
// value after it has seen my next d.q, which I publish after this launch has completed.
template <typename T, int BLOCK, bool VEC, int VNT = 0>
__global__ __launch_bounds__(BLOCK) void aypx_beta_p2p_kernel(int n, const T *x, T *y, T *xs, const T *alpha,
                                                              const typename VT<T>::acc *partials, int P,
                                                              char *const *mailbox, int rank, int nranks, int which,
                                                              const unsigned long long *epoch, T *delta, T *beta, T *history,
                                                              int history_cap, const int *iter) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ double vx[64], vy[64];
    __shared__ T beta_s;
    const unsigned long long ep = *epoch;
    const int s = threadIdx.x;
    if (blockIdx.x == 0) {
        A acc = vzero<A>();
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, partials[i]);
        const A tot = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) { const double2 v2 = to_acc2(tot); vx[0] = v2.x; vy[0] = v2.y; }
        __syncthreads();
        if (s < nranks) {
            unsigned long long *slot = reinterpret_cast<unsigned long long *>(mailbox[s] + kMbSlots) + ((long long)which * 64 + rank) * 4;
            st_sys(slot, (unsigned long long)__double_as_longlong(vx[0]));
            st_sys(slot + 1, (unsigned long long)__double_as_longlong(vy[0]));
            p2p_stores_done();
            st_sys(slot + 2, ep);
        }
        __syncthreads();
    }
    if (s < nranks) {
        const unsigned long long *in = reinterpret_cast<const unsigned long long *>(mailbox[rank] + kMbSlots) + ((long long)which * 64 + s) * 4;
        if (!spin_until(in + 2, ep, mailbox[rank])) st_sys(reinterpret_cast<unsigned long long *>(mailbox[rank] + kMbError), 2ULL);
        vx[s] = __longlong_as_double((long long)ld_sys(in));
        vy[s] = __longlong_as_double((long long)ld_sys(in + 1));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 tot2 = make_double2(0., 0.);
        for (int k = 0; k < nranks; ++k) { tot2.x += vx[k]; tot2.y += vy[k]; }
        const int it = *iter;
        const T dnT = from_acc<T>(from_acc2<A>(tot2));
        const T dold = history[it - 1];
        const T bt = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
        beta_s = bt;
        if (blockIdx.x == 0) {
            beta[0] = bt;
            delta[0] = dnT;
            if (it < history_cap) history[it] = dnT;
        }
    }
    __syncthreads();
    // deferred x += alpha d of this iteration rides along (ten-vector-pass iteration, see aypx_beta_x_kernel)
    const T bt = beta_s, al = alpha[0];
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py = ld_pack(y + i * E), ps = (VNT & 1) ? ld_pack_nt(xs + i * E) : ld_pack(xs + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                ps.v[k] = vadd(ps.v[k], vmul(al, py.v[k]));
                py.v[k] = vaypx(bt, py.v[k], px.v[k]);
            }
            if (VNT & 1) st_pack_nt(xs + i * E, ps); else st_pack(xs + i * E, ps);
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T dv = y[i];
        xs[i] = vadd(xs[i], vmul(al, dv));
        y[i] = vaypx(bt, dv, x[i]);
    }
}

// local partials -> sum over all ranks, in rank order on every rank (bitwise identical everywhere):
// one work-group; thread s < nranks writes my value into rank s's slot, then waits for rank s's value in mine.
// The scalar step that consumes the sum rides in the same launch (MODE): 1 = cg_delta0, 2 = cg_alpha, 3 = cg_beta.
template <typename T, int MODE>
__global__ __launch_bounds__(kScalarBlock) void p2p_allreduce_kernel(const typename VT<T>::acc *partials, int grid,
                                                                     char *const *mailbox, int rank, int nranks, int which,
                                                                     unsigned long long *epoch, T *delta, T *alpha, T *beta,
                                                                     T *history, int history_cap, int *iter,
                                                                     unsigned long long *bump0, unsigned long long *bump1) {
    using A = typename VT<T>::acc;
    __shared__ A smem[kScalarBlock / kWave];
    __shared__ double vx[64], vy[64];
    const A loc = sum_partials_block(partials, grid, smem);   // broadcast to every thread
    const unsigned long long ep = *epoch + 1;
    const int s = threadIdx.x;
    if (s < nranks) {
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(mailbox[s] + kMbSlots) + ((long long)which * 64 + rank) * 4;
        const double2 v2 = to_acc2(loc);
        st_sys(slot, (unsigned long long)__double_as_longlong(v2.x));
        st_sys(slot + 1, (unsigned long long)__double_as_longlong(v2.y));
        p2p_stores_done();
        st_sys(slot + 2, ep);
        const unsigned long long *in = reinterpret_cast<const unsigned long long *>(mailbox[rank] + kMbSlots) + ((long long)which * 64 + s) * 4;
        if (!spin_until(in + 2, ep, mailbox[rank])) st_sys(reinterpret_cast<unsigned long long *>(mailbox[rank] + kMbError), 2ULL);
        vx[s] = __longlong_as_double((long long)ld_sys(in));
        vy[s] = __longlong_as_double((long long)ld_sys(in + 1));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 tot2 = make_double2(0., 0.);
        for (int k = 0; k < nranks; ++k) { tot2.x += vx[k]; tot2.y += vy[k]; }
        const A tot = from_acc2<A>(tot2);
        *epoch = ep;
        if (bump0) *bump0 = *bump0 + 1;     // four-launch loop: the exchange epoch the SpMV launch just used ...
        if (bump1) *bump1 = *bump1 + 1;     // ... and the epoch aypx_beta_p2p_kernel will read
        if (MODE == 1) {                       // cg_delta0 (clcg.c:274-292)
            delta[0] = from_acc<T>(tot);
            history[0] = from_acc<T>(tot);
            *iter = 0;
        } else if (MODE == 2) {                // cg_alpha (clcg.c:317-327)
            const T dqT = from_acc<T>(tot);
            alpha[0] = from_acc<T>(acc_div(to_acc(delta[0]), to_acc(dqT)));
            *iter = *iter + 1;
        } else {                               // cg_beta (clcg.c:376-391)
            const int it = *iter;
            const T dnT = from_acc<T>(tot);
            beta[0] = from_acc<T>(acc_div(to_acc(dnT), to_acc(delta[0])));
            delta[0] = dnT;
            if (it < history_cap) history[it] = dnT;
        }
    }
}

// =================================================================================================
// Synthetic generators (device side, so multi-GB systems never cross PCIe)
// =================================================================================================
__host__ __device__ inline long long lap3d_ptr(long long i, long long nx, long long ny, long long nz) {
    // entries stored before row i = 7 i - (missing neighbours of rows < i), closed form
    const long long pl = nx * ny, n = pl * nz;
    const long long x0 = (i + nx - 1) / nx;                       // rows j<i with ix == 0
    const long long x1 = i / nx;                                  // ix == nx-1
    const long long full = i / pl, rem = i % pl;
    const long long y0 = full * nx + (rem < nx ? rem : nx);       // iy == 0
    const long long y1 = full * nx + (rem > pl - nx ? rem - (pl - nx) : 0);  // iy == ny-1
    const long long z0 = i < pl ? i : pl;                         // iz == 0
    const long long z1 = i > n - pl ? i - (n - pl) : 0;           // iz == nz-1
    return 7 * i - (x0 + x1 + y0 + y1 + z0 + z1);
}
long long laplace3d_ptr(long long i, int nx, int ny, int nz) { return lap3d_ptr(i, nx, ny, nz); }

template <typename T> CG_DEV T real_val(double v);
template <> CG_DEV float real_val<float>(double v) { return (float)v; }
template <> CG_DEV double real_val<double>(double v) { return v; }
template <> CG_DEV float2 real_val<float2>(double v) { return make_float2((float)v, 0.f); }
template <> CG_DEV double2 real_val<double2>(double v) { return make_double2(v, 0.); }

template <typename T>
__global__ void gen_laplace3d_kernel(int nx, int ny, int nz, long long row_begin, long long row_end, T *vals, int *ptr,
                                     int *cols) {
    const long long nloc = row_end - row_begin;
    const long long base = lap3d_ptr(row_begin, nx, ny, nz);
    for (long long li = (long long)blockIdx.x * blockDim.x + threadIdx.x; li <= nloc;
         li += (long long)gridDim.x * blockDim.x) {
        const long long i = row_begin + li;
        long long p = lap3d_ptr(i, nx, ny, nz) - base;
        ptr[li] = (int)p;
        if (li == nloc) break;
        const long long pl = (long long)nx * ny;
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / pl);
        if (iz > 0) { cols[p] = (int)(i - pl); vals[p++] = real_val<T>(-1.0); }
        if (iy > 0) { cols[p] = (int)(i - nx); vals[p++] = real_val<T>(-1.0); }
        if (ix > 0) { cols[p] = (int)(i - 1); vals[p++] = real_val<T>(-1.0); }
        cols[p] = (int)i; vals[p++] = real_val<T>(6.0);
        if (ix < nx - 1) { cols[p] = (int)(i + 1); vals[p++] = real_val<T>(-1.0); }
        if (iy < ny - 1) { cols[p] = (int)(i + nx); vals[p++] = real_val<T>(-1.0); }
        if (iz < nz - 1) { cols[p] = (int)(i + pl); vals[p++] = real_val<T>(-1.0); }
    }
}

__host__ __device__ inline long long poi2d_ptr(long long i, long long N) {
    const long long n = N * N;
    const long long x0 = (i + N - 1) / N, x1 = i / N;
    const long long y0 = i < N ? i : N, y1 = i > n - N ? i - (n - N) : 0;
    return 5 * i - (x0 + x1 + y0 + y1);
}
long long poisson2d_ptr(long long i, int N) { return poi2d_ptr(i, N); }

template <typename T> __global__ void gen_poisson2d_kernel(int N, T *vals, int *ptr, int *cols) {
    const long long n = (long long)N * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x) {
        long long p = poi2d_ptr(i, N);
        ptr[i] = (int)p;
        if (i == n) break;
        const int jx = (int)(i % N), iy = (int)(i / N);
        if (iy > 0) { cols[p] = (int)(i - N); vals[p++] = real_val<T>(-1.0); }
        if (jx > 0) { cols[p] = (int)(i - 1); vals[p++] = real_val<T>(-1.0); }
        cols[p] = (int)i; vals[p++] = real_val<T>(4.0);
        if (jx < N - 1) { cols[p] = (int)(i + 1); vals[p++] = real_val<T>(-1.0); }
        if (iy < N - 1) { cols[p] = (int)(i + N); vals[p++] = real_val<T>(-1.0); }
    }
}

// =================================================================================================
// Host-side launchers
// =================================================================================================
#define CG_DISPATCH(dtype, FN, ...)                                         \
    switch (dtype) {                                                        \
    case CGAMD_F32: return FN<float>(__VA_ARGS__);                          \
    case CGAMD_F64: return FN<double>(__VA_ARGS__);                         \
    case CGAMD_C64: return FN<float2>(__VA_ARGS__);                         \
    case CGAMD_C128: return FN<double2>(__VA_ARGS__);                       \
    default: return fail(CGAMD_ERR_INVALID, "bad dtype");                   \
    }

// cgamd_solver_iterate_timed: the next SpMV launch of this thread carries a start / stop event pair ON THE DISPATCH ITSELF
// (hipExtLaunchKernelGGL), so the pair measures the kernel's execution like a profiler's kernel trace does -- not the
// launch gaps and event barriers that hipEventRecord calls around a launch add (about 15 us per launch at N = 10M).
static thread_local hipEvent_t *t_kernel_events = nullptr;
void set_kernel_event_pair(hipEvent_t *pair) { t_kernel_events = pair; }
#define CG_LAUNCH_EV(KERNEL, GRID, BLOCKDIM, LDS, STREAM, ...)                                                                  \
    do {                                                                                                                       \
        if (t_kernel_events) {                                                                                                 \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, LDS, STREAM, t_kernel_events[0], t_kernel_events[1], 0, __VA_ARGS__); \
            t_kernel_events = nullptr;                                                                                         \
        } else {                                                                                                               \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, LDS, STREAM, __VA_ARGS__);                                              \
        }                                                                                                                      \
    } while (0)

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return CGAMD_OK;
}

SpmvPlan make_spmv_plan(int n) {
    SpmvPlan p;
    p.row_blocks = (n + kBlock - 1) / kBlock;
    const int cap = tune().spmv_grid > 0 ? tune().spmv_grid : kMaxGrid;
    int g = p.row_blocks < cap ? p.row_blocks : cap;
    if (g >= 8) g &= ~7;  // xcd_remap needs a multiple of 8
    if (g < 1) g = 1;
    p.grid = g;
    p.n_partials = g;
    return p;
}

int vec_grid(long long n, int dtype, int nrhs) {
    // 16-byte packs per thread the grid is sized for: 4 for streaming sizes; small systems want every CU busy instead
    // (profiles/r2_experiments/vec_ppt.log: 16k rows 10.4 -> 9.2 us per iteration with 1, 250k rows 16.2 -> 15.8 with 2,
    // N = 1M 32.2 -> 34.0 with 1); "vec_ppt" overrides
    const long long total = n * (long long)(nrhs > 0 ? nrhs : 1);
    // (up to 65536 rows always 1, whatever the number of right-hand sides: the partial-sum structure the resident loop reproduces)
    const int ppt = tune().vec_ppt > 0 ? tune().vec_ppt : ((total <= 262144 || n <= 65536) ? 1 : total <= 524288 ? 2 : 4);
    const long long per_block = (long long)kBlock * (16 / (long long)dtype_size(dtype)) * ppt;
    long long g = (n + per_block - 1) / per_block;
    const long long cap = tune().vec_grid > 0 ? tune().vec_grid : kMaxGrid;
    if (g > cap) g = cap;
    // single right-hand side, streaming sizes: exactly two work-groups per CU.  611 (1.25M rows) or 1221 (2.5M) leave the CUs
    // unevenly loaded, and every work-group of the beta launch adds all the r.r partials in its prologue: 1.25M rows 37.5 -> 36.7 us
    // per iteration, 2.5M 64.9 -> 62.5, 5M 124.8 -> 123.4, 10M 237.5 -> 235.9 (profiles/r2_experiments/vec_grid_ab.log)
    if (nrhs <= 1 && tune().vec_grid == 0 && g > 512) g = 512;
    if (g < 1) g = 1;
    return (int)g;
}

Tuning g_tune;
static std::mutex g_tune_mutex;
static thread_local const Tuning *t_tune = nullptr;
static thread_local Tuning t_tune_fallback;
Tuning tune_snapshot() {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    return g_tune;
}
void tune_set(const std::function<void(Tuning &)> &edit) {
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    edit(g_tune);
}
const Tuning &tune() {
    if (t_tune) return *t_tune;
    t_tune_fallback = tune_snapshot();      // handle-less entry (stand-alone ops): the global configuration as of now
    return t_tune_fallback;
}
// HIP decides per CALLING thread whether an API call (hipMalloc, a synchronous copy, ...) made while some stream is being
// captured is an error that invalidates that capture: the default interaction mode of a thread is "global".  The
// reference's threading model has one thread per device working independently, each capturing its own iteration graphs,
// so every thread that enters the library switches its own mode to "relaxed" once (a worker's hipMalloc must not kill a
// sibling's capture: "operation failed due to a previous error during capture").
void thread_hip_setup() {
    static thread_local bool done = false;
    if (done) return;
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    done = true;
}
TuneScope::TuneScope(const Tuning *t) : prev(t_tune) { t_tune = t; thread_hip_setup(); }
TuneScope::~TuneScope() { t_tune = prev; }

template <typename T>
static int spmv_impl(const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                     const void *x, long long ldx, void *y, long long ldy, int nrhs, const void *dvec, void *partials,
                     const int *rb_list, int rb_count, hipStream_t st) {
    SpmvArgs<T> a;
    a.n = n; a.nrhs = nrhs; a.nnz = nnz;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<const T *>(x); a.ldx = ldx;
    a.y = static_cast<T *>(y); a.ldy = ldy;
    a.dvec = static_cast<const T *>(dvec);
    a.partials = static_cast<typename VT<T>::acc *>(partials);
    a.row_blocks = plan.row_blocks;
    a.rb_list = rb_list; a.rb_count = rb_count;
    a.codes = nullptr; a.dict = nullptr;
    const bool vec = aligned16(vals) && aligned16(cols);
    const bool fuse = partials != nullptr;
    const size_t dyn = (fuse && nrhs > 1) ? sizeof(typename VT<T>::acc) * nrhs * (kBlock / kWave) : 0;
    dim3 grid(plan.grid), block(kBlock);
    const int variant = (vec && ((nrhs == 1 && plan.kind == 5) || (nrhs > 1 && plan.kind == 6 && plan.wide))) ? 5 : 0;
    if (variant == 5) {
        a.cap = (plan.max_span + 3) & ~3;
        a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
        dim3 g5(rb_list ? (rb_count > 0 ? rb_count : 1) : rowblock_grid(plan.row_blocks, a.cycle), nrhs);
        if (rb_list && rb_count <= 0) return CGAMD_OK;
        const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
        // value stream interleaved across the lanes in 16-byte chunks (stage_slice_ilv): "spmv_ilv" 1/0, -1 = auto
        const bool ilv = tune().spmv_ilv >= 0 ? (tune().spmv_ilv != 0) : kIlvDefault<T>;
        // one-byte column codes instead of aCols (build_index_codes; the codes belong to THIS cols array)
        const bool coded = ilv && nrhs == 1 && plan.codes && plan.codes_for == cols && tune().index_codes != 0;
        a.codes = coded ? plan.codes : nullptr;
        a.dict = coded ? plan.dict : nullptr;
        const size_t lds = coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
#define CG_RB(NT, UNR)                                                                                                  \
    do {                                                                                                                \
        if (coded) {                                                                                                    \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, true, UNR, -3>), g5, block, lds, st, a);        \
            else CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, false, UNR, -3>), g5, block, lds, st, a);            \
        } else if (ilv) {                                                                                                      \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, true, UNR, -2>), g5, block, lds, st, a);        \
            else CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, false, UNR, -2>), g5, block, lds, st, a);            \
        } else {                                                                                                        \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, true, UNR>), g5, block, lds, st, a);            \
            else CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, false, UNR>), g5, block, lds, st, a);                \
        }                                                                                                               \
    } while (0)
#define CG_RBX(NT, UNR)      /* batch lengths 5 and 7: the two staged forms only */                                     \
    do {                                                                                                                \
        if (coded) {                                                                                                    \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, true, UNR, -3>), g5, block, lds, st, a);        \
            else CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, false, UNR, -3>), g5, block, lds, st, a);            \
        } else {                                                                                                        \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, true, UNR, -2>), g5, block, lds, st, a);        \
            else CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, false, UNR, -2>), g5, block, lds, st, a);            \
        }                                                                                                               \
    } while (0)
        // up to 8 gathers in flight per lane for 4/8-byte values; 4 for complex128 (8 would cost 3 waves/SIMD of occupancy).  A row is
        // walked in batches of `unroll` slots (slots past the row's end re-read its last entry and are dropped): when no row of the
        // matrix is longer than 5 or 7 entries (5-point, 7-point and P1-FE stencils) the batch is exactly that long -- one batch per
        // row and no idle slot (N = 10M 7-point fp64: SpMV 134.7 -> 131.2 us, CG 4 215 -> 4 292 it/s; same sums, same bits)
        const int fit = plan.max_row <= 0 ? 8 : plan.max_row <= 4 ? 4 : plan.max_row == 5 ? 5 : plan.max_row <= 7 ? 7 : 8;
        const int unroll = tune().spmv_unroll ? tune().spmv_unroll : (sizeof(T) > 8 ? 4 : fit);
        if (unroll == 4) { if (nt) CG_RB(true, 4); else CG_RB(false, 4); }
        else if (unroll == 5 && (coded || ilv)) { if (nt) CG_RBX(true, 5); else CG_RBX(false, 5); }
        else if (unroll == 7 && (coded || ilv)) { if (nt) CG_RBX(true, 7); else CG_RBX(false, 7); }
        else { if (nt) CG_RB(true, 8); else CG_RB(false, 8); }
#undef CG_RB
#undef CG_RBX
        return check_launch("spmv_rowblock");
    }
    if (vec && nrhs == 1 && plan.kind == 7 && !rb_list) {
        const int span = plan.chunk_span[plan.lpr == 2 ? 0 : plan.lpr == 4 ? 1 : 2];
        a.cap = (span + 3) & ~3;
        a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
        const bool coded = plan.codes && plan.codes_for == cols && tune().index_codes != 0;
        a.codes = coded ? plan.codes : nullptr;
        a.dict = coded ? plan.dict : nullptr;
        const size_t lds = coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
        const dim3 g7(rowblock_grid(plan.row_blocks, a.cycle));
        const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
        constexpr int U = sizeof(T) > 8 ? 4 : 8;
#define CG_CH(NT, L)                                                                                                     \
    do {                                                                                                                  \
        if (coded) {                                                                                                      \
            if (fuse) hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, true, L, U, true>), g7, block, lds, st, a);   \
            else hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, false, L, U, true>), g7, block, lds, st, a);       \
        } else if (fuse) hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, true, L, U>), g7, block, lds, st, a);   \
        else hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, false, L, U>), g7, block, lds, st, a);       \
    } while (0)
        if (plan.lpr == 2) { if (nt) CG_CH(true, 2); else CG_CH(false, 2); }
        else if (plan.lpr == 4) { if (nt) CG_CH(true, 4); else CG_CH(false, 4); }
        else { if (nt) CG_CH(true, 8); else CG_CH(false, 8); }
#undef CG_CH
        return check_launch("spmv_rowblock_chunked");
    }
    if (vec && nrhs > 1 && plan.kind == 6) {
        a.cap = (plan.max_span + 3) & ~3;
        a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
        const size_t lds = (size_t)a.cap * (sizeof(T) + 4) + sizeof(typename VT<T>::acc) * nrhs * (kBlock / kWave);
        dim3 g6(rowblock_grid(plan.row_blocks, a.cycle));
        constexpr int RBMAX = sizeof(T) <= 8 ? 8 : 4;
        // One launch covers all right-hand sides (groups of RB inside the kernel).  Splitting into one launch per
        // group (cgamd_tune "spmm_rb") re-reads the matrix per group and shrinks the x window per XCD; measured
        // slower at nRHS = 32 (253 vs 220 us) and at nRHS = 9 -- kept as an experiment knob only.
        const int chunk = (tune().spmm_rb > 0 && tune().spmm_rb < nrhs) ? tune().spmm_rb : nrhs;
        const bool nt6 = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
        // group width: the right-hand sides are cut into ceil(n / RBMAX) groups of (nearly) equal width, so that the last
        // group is not mostly padding -- the reference's own shape, 9 sub-domains, runs as 5 + 4 instead of 8 + 1
        const int ngroups = (chunk + RBMAX - 1) / RBMAX;
        int rbw = tune().spmm_group > 0 ? tune().spmm_group : (chunk + ngroups - 1) / ngroups;
        if (rbw > RBMAX) rbw = RBMAX;
#define CG_MM(RBW)                                                                                                         \
    do {                                                                                                                    \
        if (fuse) {                                                                                                         \
            if (nt6) hipLaunchKernelGGL((spmm_rowblock_kernel<T, kBlock, true, true, RBW>), g6, block, lds, st, b);         \
            else hipLaunchKernelGGL((spmm_rowblock_kernel<T, kBlock, false, true, RBW>), g6, block, lds, st, b);            \
        } else {                                                                                                            \
            if (nt6) hipLaunchKernelGGL((spmm_rowblock_kernel<T, kBlock, true, false, RBW>), g6, block, lds, st, b);        \
            else hipLaunchKernelGGL((spmm_rowblock_kernel<T, kBlock, false, false, RBW>), g6, block, lds, st, b);           \
        }                                                                                                                   \
    } while (0)
        for (int g0 = 0; g0 < nrhs; g0 += chunk) {
            SpmvArgs<T> b = a;
            b.nrhs = (nrhs - g0 < chunk) ? nrhs - g0 : chunk;
            b.x = a.x + (long long)g0 * ldx;
            b.y = a.y + (long long)g0 * ldy;
            if (fuse) {
                b.dvec = a.dvec + (long long)g0 * ldx;
                b.partials = a.partials + (long long)g0 * plan.row_blocks;
            }
            if (RBMAX == 8 && rbw > 6) CG_MM(RBMAX);
            else if (RBMAX == 8 && rbw == 6) CG_MM(6);
            else if (RBMAX == 8 && rbw == 5) CG_MM(5);
            else if (rbw == 4 || (RBMAX == 4 && rbw > 3)) CG_MM(4);
            else if (rbw == 3) CG_MM(3);
            else CG_MM(2);
        }
#undef CG_MM
        return check_launch("spmm_rowblock");
    }
    if (vec) {
        if (fuse) hipLaunchKernelGGL((spmv_stream_kernel<T, kBlock, kQuadsPerThread, true, true>), grid, block, dyn, st, a);
        else hipLaunchKernelGGL((spmv_stream_kernel<T, kBlock, kQuadsPerThread, true, false>), grid, block, dyn, st, a);
    } else {
        if (fuse) hipLaunchKernelGGL((spmv_stream_kernel<T, kBlock, kQuadsPerThread, false, true>), grid, block, dyn, st, a);
        else hipLaunchKernelGGL((spmv_stream_kernel<T, kBlock, kQuadsPerThread, false, false>), grid, block, dyn, st, a);
    }
    return check_launch("spmv");
}

int validate_csr_device(int n, long long nnz, int ncols, const int *ptr_dev, const int *cols_dev, const int *index_dev, int n_index,
                        int index_bound, int *scratch_dev, hipStream_t st) {
    CG_HIP(hipMemsetAsync(scratch_dev, 0, sizeof(int), st));
    long long work = std::max<long long>(nnz, n);
    int g = (int)std::min<long long>((work + 255) / 256, 4096);
    if (g < 1) g = 1;
    hipLaunchKernelGGL(csr_validate_kernel, dim3(g), dim3(256), 0, st, n, nnz, ncols, ptr_dev, cols_dev, index_dev, n_index, index_bound, scratch_dev);
    if (int rc = check_launch("csr_validate")) return rc;
    int flag = 0;
    CG_HIP(hipMemcpyAsync(&flag, scratch_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    if (flag & 1) return fail(CGAMD_ERR_INVALID, "CSR: aPointers[0] != 0");
    if (flag & 2) return fail(CGAMD_ERR_INVALID, "CSR: aPointers not monotone");
    if (flag & 4) return fail(CGAMD_ERR_INVALID, "CSR: aPointers[size] != nonZeros");
    if (flag & 8) return fail(CGAMD_ERR_INVALID, "CSR: column index out of range");
    if (flag & 16) return fail(CGAMD_ERR_INVALID, "partition plan: send_index entry out of range");
    return CGAMD_OK;
}

int compute_spmv_plan(const int *ptr_dev, const int *cols_dev, int n, int *scratch_dev, hipStream_t st, SpmvPlan *plan) {
    // (1) largest slice span -> which kernels apply, LDS size
    const int row_blocks = (n + kBlock - 1) / kBlock;
    CG_HIP(hipMemsetAsync(scratch_dev, 0, 6 * sizeof(int), st));
    int g = (row_blocks + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL((spmv_span_kernel<kBlock>), dim3(g), dim3(256), 0, st, n, ptr_dev, row_blocks, scratch_dev);
    if (int rc = check_launch("spmv_span")) return rc;
    int spans[6] = {0, 0, 0, 0, 0, 0};
    CG_HIP(hipMemcpyAsync(spans, scratch_dev, 6 * sizeof(int), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    plan->max_span = spans[0];
    plan->max_row = spans[5];
    for (int lv = 0; lv < 3; ++lv) plan->chunk_span[lv] = spans[lv + 1];
    plan->max_quad = spans[4];
    (void)cols_dev;
    return CGAMD_OK;
}

// ---- one-byte column codes (SpmvPlan::codes) ---------------------------------------------------------------------------
constexpr int kDictSlots = 1024;            // open-addressing table of the distinct offsets (<= 256 accepted)
constexpr int kDictEmpty = -2147483647 - 1;
CG_DEV unsigned dict_hash(int d) { return ((unsigned)d * 2654435761u) >> 22; }   // 10 bits
// one lane per row: every (column - row) offset goes into the table; count[0] = distinct offsets so far
__global__ __launch_bounds__(256) void index_offsets_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                            int *table, int *count) {
    for (long long row = blockIdx.x * 256LL + threadIdx.x; row < n; row += 256LL * gridDim.x) {
        if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 256) return;     // not codable: stop early
        int last = kDictEmpty;
        for (int j = ptr[row], e = ptr[row + 1]; j < e; ++j) {
            const int d = cols[j] - (int)row;
            if (d == last) continue;
            last = d;
            unsigned h = dict_hash(d);
            int probes = 0;
            for (; probes < kDictSlots; ++probes, h = (h + 1) & (kDictSlots - 1)) {
                int v = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == kDictEmpty) {
                    v = atomicCAS(table + h, kDictEmpty, d);
                    if (v == kDictEmpty) { atomicAdd(count, 1); break; }
                }
                if (v == d) break;
            }
            if (probes == kDictSlots) { atomicAdd(count, kDictSlots); return; }       // table full
        }
    }
}
// table slot -> code (position of the offset in the sorted dictionary); one lane per row writes its codes
__global__ __launch_bounds__(256) void index_encode_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                           const int *__restrict__ table, const unsigned char *__restrict__ slot_code,
                                                           unsigned char *__restrict__ codes) {
    __shared__ int stab[kDictSlots];
    __shared__ unsigned char scode[kDictSlots];
    for (int i = threadIdx.x; i < kDictSlots; i += 256) { stab[i] = table[i]; scode[i] = slot_code[i]; }
    __syncthreads();
    for (long long row = blockIdx.x * 256LL + threadIdx.x; row < n; row += 256LL * gridDim.x) {
        for (int j = ptr[row], e = ptr[row + 1]; j < e; ++j) {
            const int d = cols[j] - (int)row;
            unsigned h = dict_hash(d);
            while (stab[h] != d) h = (h + 1) & (kDictSlots - 1);      // every offset is in the table
            codes[j] = scode[h];
        }
    }
}

int build_index_codes(int n, long long nnz, const int *ptr_dev, const int *cols_dev, hipStream_t st, unsigned char **codes_out,
                      int **dict_out, int *distinct_out) {
    *codes_out = nullptr;
    *dict_out = nullptr;
    *distinct_out = 0;
    if (n <= 0 || nnz <= 0) return CGAMD_OK;
    int *work = nullptr;        // [1024 table | 1 count | 256 dict]
    CG_HIP(hipMalloc((void **)&work, (kDictSlots + 64 + 256) * sizeof(int) + kDictSlots));
    std::vector<int> table(kDictSlots + 1, kDictEmpty);
    table[kDictSlots] = 0;
    hipError_t e = hipMemcpyAsync(work, table.data(), (kDictSlots + 1) * sizeof(int), hipMemcpyHostToDevice, st);
    int g = (int)std::min<long long>((n + 255LL) / 256, 8192);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(index_offsets_kernel, dim3(g), dim3(256), 0, st, n, ptr_dev, cols_dev, work, work + kDictSlots);
        e = hipMemcpyAsync(table.data(), work, (kDictSlots + 1) * sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(work); return fail(CGAMD_ERR_HIP, std::string("build_index_codes: ") + hipGetErrorString(e)); }
    const int distinct = table[kDictSlots];
    if (distinct < 1 || distinct > 256) { (void)hipFree(work); return CGAMD_OK; }
    std::vector<int> dict;
    for (int i = 0; i < kDictSlots; ++i)
        if (table[i] != kDictEmpty) dict.push_back(table[i]);
    std::sort(dict.begin(), dict.end());
    std::vector<unsigned char> slot_code(kDictSlots, 0);
    for (int i = 0; i < kDictSlots; ++i)
        if (table[i] != kDictEmpty) slot_code[i] = (unsigned char)(std::lower_bound(dict.begin(), dict.end(), table[i]) - dict.begin());
    dict.resize(256, dict[0]);
    unsigned char *codes = nullptr;
    int *dict_dev = nullptr;
    unsigned char *slot_dev = reinterpret_cast<unsigned char *>(work + kDictSlots + 64 + 256);
    e = hipMalloc((void **)&codes, (size_t)nnz + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&dict_dev, 256 * sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(codes + nnz, 0, 64, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dict_dev, dict.data(), 256 * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(slot_dev, slot_code.data(), kDictSlots, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(index_encode_kernel, dim3(g), dim3(256), 0, st, n, ptr_dev, cols_dev, work, slot_dev, codes);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(work);
    if (e != hipSuccess) {
        if (codes) (void)hipFree(codes);
        if (dict_dev) (void)hipFree(dict_dev);
        return fail(CGAMD_ERR_HIP, std::string("build_index_codes: ") + hipGetErrorString(e));
    }
    *codes_out = codes;
    *dict_out = dict_dev;
    *distinct_out = distinct;
    return CGAMD_OK;
}

// decides once which SpMV kernel a solver uses (and therefore how many dot partials it produces)
void finalize_spmv_plan(SpmvPlan *plan, int dtype, int nrhs, int n, long long nnz, const void *vals, const int *cols) {
    // Cache policy (profiles/r1_experiments/ab_nt_sizes2.log, z-slabs of the 250x200x200 system, fp64).  A matrix of up to
    // ~256 MB stays mostly resident in the 256 MB Infinity Cache from one iteration to the next: streaming it
    // non-temporally only throws that away (1.25M rows: 47.1 -> 43.3 us/iteration, 2.5M rows: 80.3 -> 76.0).  Larger
    // matrices are streamed non-temporally so that the vectors, which ARE re-used within the iteration, keep the cache
    // (3.75M rows: 117 -> 111 us, 5M: 152 -> 144).  axpy2_dot's streaming hints for x and q pay when the working set is
    // far beyond the cache (>= 7.5M rows) or when the matrix competes for it (2.5M rows), not in between.
    const size_t matrix_bytes = (size_t)nnz * (dtype_size(dtype) + 4) + ((size_t)n + 1) * 4;
    const size_t vector_bytes = (size_t)n * dtype_size(dtype) * (size_t)nrhs;
    const size_t MB = (size_t)1 << 20;
    plan->nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (matrix_bytes > 256 * MB);
    if (tune().vec_nt >= 0) plan->vec_nt = tune().vec_nt;
    else if (!plan->nt) plan->vec_nt = (matrix_bytes + 5 * vector_bytes <= 200 * MB) ? 0 : 3;
    else plan->vec_nt = (matrix_bytes <= 512 * MB) ? 0 : 3;
    int kind = tune().spmv_variant;
    const bool vec = aligned16(vals) && aligned16(cols);
    if (!vec || plan->max_span <= 0) kind = 0;
    plan->lpr = 1;
    if (kind == 5 && (size_t)plan->max_span * (dtype_size(dtype) + 4) + acc_size(dtype) * (size_t)nrhs * (kBlock / 64) >
                         (size_t)(tune().spmv_slice_kb > 0 ? tune().spmv_slice_kb * 1024 : nrhs > 1 ? kMaxSpmmSliceBytes : kMaxSliceBytes)) {
        kind = 0;
        // denser rows: the chunked form of the row-block kernel (single right-hand side).  Smallest LPR whose chunk slice
        // stays below ~32 KB (27-point stencil fp64: 4 lanes per row 138 us / CG 170 us, 2 lanes 139 / 177, 8 lanes 209;
        // f32: 2 lanes 91 / 117, 4 lanes 94 / 122); rows
        // so dense that even 32 of them exceed that may use up to 48 KB with 8 lanes per row
        if (nrhs == 1 && tune().spmv_chunked != 0) {
            const size_t ebytes = dtype_size(dtype) + 4;
            const size_t want = (size_t)(tune().spmv_chunk_kb > 0 ? tune().spmv_chunk_kb * 1024 : kChunkBytes);
            for (int lv = 0; lv < 3 && kind == 0; ++lv)
                if (plan->chunk_span[lv] > 0 && (size_t)plan->chunk_span[lv] * ebytes <= want) {
                    kind = 7;
                    plan->lpr = 2 << lv;
                }
            if (kind == 0 && plan->chunk_span[2] > 0 && (size_t)plan->chunk_span[2] * ebytes <= (size_t)kMaxChunkBytes) {
                kind = 7;
                plan->lpr = 8;
            }
        }
    }
    if (kind != 5 && kind != 7) kind = 0;
    if (kind == 5 && nrhs > 1) kind = 6;      // SpMM form of the row-block kernel
    // small multi-RHS systems are bound by round trips per work-group, not by bytes: one work-group per (row block, RHS)
    // runs the single-RHS kernel ("wide" form) instead of one work-group walking the right-hand sides in register groups
    plan->wide = kind == 6 && (long long)plan->row_blocks * nrhs <= (tune().spmm_wide_max >= 0 ? tune().spmm_wide_max : 4096) &&
                 (size_t)plan->max_span * (dtype_size(dtype) + 4) <= (size_t)kMaxSliceBytes;
    plan->kind = kind;
    plan->n_partials = kind ? plan->row_blocks : plan->grid;
}

int launch_halo_flags(int n, const int *ptr, const int *cols, int n_local, int row_blocks, int *flag, hipStream_t st) {
    hipLaunchKernelGGL((rowblock_halo_flag_kernel<kBlock>), dim3(row_blocks), dim3(256), 0, st, n, ptr, cols, n_local, row_blocks, flag);
    return check_launch("halo_flags");
}

int launch_spmv(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr,
                const int *cols, const void *x, long long ldx, void *y, long long ldy, int nrhs, const void *dvec,
                void *partials, hipStream_t st, const int *rb_list, int rb_count) {
    if (n <= 0) return CGAMD_OK;
    if (rb_list && !(plan.kind == 5 && nrhs == 1)) return fail(CGAMD_ERR_INVALID, "spmv: row-block lists need the row-block kernel");
    CG_DISPATCH(dtype, spmv_impl, plan, n, nnz, vals, ptr, cols, x, ldx, y, ldy, nrhs, dvec, partials, rb_list, rb_count, st);
}

static bool vec_ok(int dtype, long long ld, int nrhs, std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    if (nrhs > 1 && ((ld * (long long)dtype_size(dtype)) & 15)) return false;
    return true;
}

template <typename T>
static int dot_impl(int n, const void *a, const void *b, long long ld, int nrhs, void *partials, int grid, bool vec,
                    hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    auto *pp = static_cast<typename VT<T>::acc *>(partials);
    if (vec) hipLaunchKernelGGL((dot_partials_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)a, (const T *)b, ld, pp);
    else hipLaunchKernelGGL((dot_partials_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)a, (const T *)b, ld, pp);
    return check_launch("vdot");
}
int launch_dot_partials(int dtype, int n, const void *a, const void *b, long long ld, int nrhs, void *partials, int grid,
                        hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {a, b});
    CG_DISPATCH(dtype, dot_impl, n, a, b, ld, nrhs, partials, grid, vec, st);
}

template <typename T> static int reduce_impl(const void *partials, int grid, int nrhs, void *result, hipStream_t st) {
    hipLaunchKernelGGL((reduce_to_value_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st,
                       static_cast<const typename VT<T>::acc *>(partials), grid, nrhs, static_cast<T *>(result));
    return check_launch("reduce");
}
int launch_reduce_to_value(int dtype, const void *partials, int grid, int nrhs, void *result, hipStream_t st) {
    CG_DISPATCH(dtype, reduce_impl, partials, grid, nrhs, result, st);
}

template <typename T, int OP>
static int ewise_impl(int n, const void *x, void *y, const void *b2, long long ld, const void *alpha, int nrhs, bool vec,
                      hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    if (vec) hipLaunchKernelGGL((ewise_kernel<T, kBlock, true, OP>), g, blk, 0, st, n, (const T *)x, (T *)y, (const T *)b2, ld, (const T *)alpha);
    else hipLaunchKernelGGL((ewise_kernel<T, kBlock, false, OP>), g, blk, 0, st, n, (const T *)x, (T *)y, (const T *)b2, ld, (const T *)alpha);
    return check_launch("ewise");
}
template <typename T> static int axpy_p(int n, const void *x, void *y, long long ld, const void *a, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 0>(n, x, y, nullptr, ld, a, nrhs, v, st); }
template <typename T> static int axpy_m(int n, const void *x, void *y, long long ld, const void *a, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 1>(n, x, y, nullptr, ld, a, nrhs, v, st); }
template <typename T> static int aypx_i(int n, const void *x, void *y, long long ld, const void *a, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 2>(n, x, y, nullptr, ld, a, nrhs, v, st); }
template <typename T> static int sub_i(int n, const void *a, const void *b, void *res, long long ld, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 3>(n, a, res, b, ld, nullptr, nrhs, v, st); }

int launch_axpy(int dtype, int n, const void *x, void *y, long long ld, const void *a, int sign, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y});
    if (sign) { CG_DISPATCH(dtype, axpy_p, n, x, y, ld, a, nrhs, v, st); }
    CG_DISPATCH(dtype, axpy_m, n, x, y, ld, a, nrhs, v, st);
}
int launch_aypx(int dtype, int n, const void *x, void *y, long long ld, const void *a, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y});
    CG_DISPATCH(dtype, aypx_i, n, x, y, ld, a, nrhs, v, st);
}
int launch_sub(int dtype, int n, const void *a, const void *b, void *res, long long ld, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {a, b, res});
    CG_DISPATCH(dtype, sub_i, n, a, b, res, ld, nrhs, v, st);
}

template <typename T>
static int axpy2_impl(int n, const void *d, void *x, const void *q, void *r, long long ld, const void *alpha, int nrhs,
                      void *partials, int grid, bool vec, int vnt, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    auto *pp = static_cast<typename VT<T>::acc *>(partials);
    if (vec && vnt == 1) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true, 1>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec && vnt == 2) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true, 2>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec && vnt == 3) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true, 3>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    return check_launch("axpy2_dot");
}
template <typename T>
static int axpy2_alpha_impl(int n, const void *d, void *x, const void *q, void *r, long long ld, const void *part_dq, int P,
                            const CgScalars &sc, int nrhs, void *partials, int grid, bool vec, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
    if (vec) hipLaunchKernelGGL((axpy2_dot_alpha_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const A *)part_dq, P, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    else hipLaunchKernelGGL((axpy2_dot_alpha_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const A *)part_dq, P, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    return check_launch("axpy2_dot_alpha");
}
template <typename T>
static int axpy_dot_impl(int n, const void *q, void *r, long long ld, const void *alpha, int nrhs, void *partials, int grid, bool vec,
                         int vnt, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    auto *pp = static_cast<typename VT<T>::acc *>(partials);
    if (vec && (vnt & 2)) hipLaunchKernelGGL((axpy_dot_kernel<T, kBlock, true, 2>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec) hipLaunchKernelGGL((axpy_dot_kernel<T, kBlock, true, 0>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else hipLaunchKernelGGL((axpy_dot_kernel<T, kBlock, false, 0>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    return check_launch("axpy_dot");
}
int launch_axpy_dot(int dtype, int n, const void *q, void *r, long long ld, const void *alpha, int nrhs, void *partials, int grid,
                    hipStream_t st, int vec_nt) {
    const bool vec = vec_ok(dtype, ld, nrhs, {q, r});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, axpy_dot_impl, n, q, r, ld, alpha, nrhs, partials, grid, vec, vnt, st);
}
template <typename T>
static int axpy_dot_alpha_impl(int n, const void *q, void *r, long long ld, const void *part_dq, int P, const CgScalars &sc, int nrhs,
                               void *partials, int grid, bool vec, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
    if (vec) hipLaunchKernelGGL((axpy_dot_alpha_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const A *)part_dq, P, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    else hipLaunchKernelGGL((axpy_dot_alpha_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const A *)part_dq, P, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    return check_launch("axpy_dot_alpha");
}
int launch_axpy_dot_alpha(int dtype, int n, const void *q, void *r, long long ld, const void *part_dq, int P, const CgScalars &sc,
                          int nrhs, void *partials, int grid, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {q, r});
    CG_DISPATCH(dtype, axpy_dot_alpha_impl, n, q, r, ld, part_dq, P, sc, nrhs, partials, grid, vec, st);
}
template <typename T>
static int aypx_beta_x_impl(int n, const void *x, void *y, void *xs, long long ld, const void *partials, int P, int nrhs,
                            const CgScalars &sc, bool vec, int vnt, hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    auto *pp = static_cast<const typename VT<T>::acc *>(partials);
#define CG_AX(V, N) hipLaunchKernelGGL((aypx_beta_x_kernel<T, kBlock, V, N>), g, blk, 0, st, n, (const T *)x, (T *)y, (T *)xs, ld, pp, P, nrhs, \
                                       (const T *)sc.alpha, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, (const int *)sc.iter)
    if (vec && (vnt & 1)) CG_AX(true, 1); else if (vec) CG_AX(true, 0); else CG_AX(false, 0);
#undef CG_AX
    return check_launch("aypx_beta_x");
}
int launch_aypx_beta_x(int dtype, int n, const void *x, void *y, void *xs, long long ld, const void *partials, int P, int nrhs,
                       const CgScalars &sc, hipStream_t st, int vec_nt) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y, xs});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, aypx_beta_x_impl, n, x, y, xs, ld, partials, P, nrhs, sc, v, vnt, st);
}
bool fold_alpha_ok(int n_partials) { return tune().fold_alpha != 0 && n_partials <= kFoldAlphaMax; }
int launch_axpy2_dot_alpha(int dtype, int n, const void *d, void *x, const void *q, void *r, long long ld, const void *part_dq,
                           int P, const CgScalars &sc, int nrhs, void *partials, int grid, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {d, x, q, r});
    CG_DISPATCH(dtype, axpy2_alpha_impl, n, d, x, q, r, ld, part_dq, P, sc, nrhs, partials, grid, vec, st);
}
int launch_axpy2_dot(int dtype, int n, const void *d, void *x, const void *q, void *r, long long ld, const void *alpha,
                     int nrhs, void *partials, int grid, hipStream_t st, int vec_nt) {
    const bool vec = vec_ok(dtype, ld, nrhs, {d, x, q, r});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, axpy2_impl, n, d, x, q, r, ld, alpha, nrhs, partials, grid, vec, vnt, st);
}

template <typename T> static int delta0_impl(const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    hipLaunchKernelGGL((cg_delta0_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, static_cast<const typename VT<T>::acc *>(partials),
                       grid, nrhs, (T *)s.delta, (T *)s.history, s.iter);
    return check_launch("cg_delta0");
}
int launch_cg_delta0(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    CG_DISPATCH(dtype, delta0_impl, partials, grid, nrhs, s, st);
}
template <typename T> static int alpha_impl(const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    using A = typename VT<T>::acc;
    if (s.stage && s.ticket && grid >= 16384 && tune().alpha_two_level != 0)
        hipLaunchKernelGGL((cg_alpha2_kernel<T>), dim3(kAlphaParts, nrhs), dim3(kScalarBlock), 0, st, static_cast<const A *>(partials),
                           grid, nrhs, (const T *)s.delta, (T *)s.alpha, s.iter, (A *)s.stage, s.ticket);
    else
        hipLaunchKernelGGL((cg_alpha_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, static_cast<const A *>(partials),
                           grid, nrhs, (const T *)s.delta, (T *)s.alpha, s.iter);
    return check_launch("cg_alpha");
}
int launch_cg_alpha(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    CG_DISPATCH(dtype, alpha_impl, partials, grid, nrhs, s, st);
}
template <typename T> static int beta_impl(const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    hipLaunchKernelGGL((cg_beta_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, static_cast<const typename VT<T>::acc *>(partials),
                       grid, nrhs, (T *)s.delta, (T *)s.beta, (T *)s.history, s.history_cap, s.iter);
    return check_launch("cg_beta");
}
int launch_cg_beta(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    CG_DISPATCH(dtype, beta_impl, partials, grid, nrhs, s, st);
}

template <typename T>
static int gen3d_impl(int nx, int ny, int nz, long long rb, long long re, void *vals, int *ptr, int *cols, hipStream_t st) {
    const long long nloc = re - rb + 1;
    int g = (int)((nloc + 255) / 256 < 8192 ? (nloc + 255) / 256 : 8192);
    hipLaunchKernelGGL((gen_laplace3d_kernel<T>), dim3(g), dim3(256), 0, st, nx, ny, nz, rb, re, (T *)vals, ptr, cols);
    return check_launch("gen_laplace3d");
}
int launch_gen_laplace3d(int dtype, int nx, int ny, int nz, long long row_begin, long long row_end, void *vals, int *ptr,
                         int *cols, hipStream_t st) {
    CG_DISPATCH(dtype, gen3d_impl, nx, ny, nz, row_begin, row_end, vals, ptr, cols, st);
}
template <typename T> static int gen2d_impl(int N, void *vals, int *ptr, int *cols, hipStream_t st) {
    const long long n = (long long)N * N + 1;
    int g = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL((gen_poisson2d_kernel<T>), dim3(g), dim3(256), 0, st, N, (T *)vals, ptr, cols);
    return check_launch("gen_poisson2d");
}
int launch_gen_poisson2d(int dtype, int N, void *vals, int *ptr, int *cols, hipStream_t st) {
    CG_DISPATCH(dtype, gen2d_impl, N, vals, ptr, cols, st);
}

int launch_reduce_to_acc(int dtype, const void *partials, int grid, int nrhs, void *out, hipStream_t st) {
    if (dtype == CGAMD_F32 || dtype == CGAMD_F64)
        hipLaunchKernelGGL((reduce_to_acc_kernel<double>), dim3(nrhs), dim3(kScalarBlock), 0, st, (const double *)partials, grid, nrhs, (double *)out);
    else
        hipLaunchKernelGGL((reduce_to_acc_kernel<double2>), dim3(nrhs), dim3(kScalarBlock), 0, st, (const double2 *)partials, grid, nrhs, (double2 *)out);
    return check_launch("reduce_to_acc");
}

template <typename T> static int pack_impl(int count, const int *index, const void *v, void *out, hipStream_t st) {
    int g = (count + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL((pack_kernel<T>), dim3(g), dim3(256), 0, st, count, index, (const T *)v, (T *)out);
    return check_launch("pack");
}
int launch_pack(int dtype, int count, const int *index, const void *v, void *out, hipStream_t st) {
    if (count <= 0) return CGAMD_OK;
    CG_DISPATCH(dtype, pack_impl, count, index, v, out, st);
}

template <typename T> static int transpose_impl(int rows, int cols, const void *in, void *out, hipStream_t st) {
    dim3 g((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL((transpose_kernel<T>), g, dim3(256), 0, st, rows, cols, (const T *)in, (T *)out);
    return check_launch("transpose");
}
int launch_transpose(int dtype, int rows, int cols, const void *in, void *out, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return CGAMD_OK;
    CG_DISPATCH(dtype, transpose_impl, rows, cols, in, out, st);
}

template <typename T>
static int aypx_beta_impl(int n, const void *x, void *y, long long ld, const void *partials, int P, int nrhs,
                          const CgScalars &sc, bool vec, hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    auto *pp = static_cast<const typename VT<T>::acc *>(partials);
    if (vec) hipLaunchKernelGGL((aypx_beta_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)x, (T *)y, ld, pp, P, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, sc.iter);
    else hipLaunchKernelGGL((aypx_beta_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)x, (T *)y, ld, pp, P, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, sc.iter);
    return check_launch("aypx_beta");
}
int launch_aypx_beta(int dtype, int n, const void *x, void *y, long long ld, const void *partials, int P, int nrhs,
                     const CgScalars &sc, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y});
    CG_DISPATCH(dtype, aypx_beta_impl, n, x, y, ld, partials, P, nrhs, sc, v, st);
}

// ---- two-launch iteration: SpMV fused with the previous iteration's beta / aypx ---------------------------------
bool fused2_ok(const SpmvPlan &plan, int dtype, int nrhs, const void *vals, const int *cols) {
    (void)dtype;
    if (tune().two_launch == 0 || !fold_alpha_ok(plan.n_partials)) return false;
    if (!aligned16(vals) || !aligned16(cols)) return false;
    // measured (profiles/r2/configs_two_launch.log): the second gather pays for the saved launch up to a few hundred
    // thousand rows; at N = 1M (3907 row blocks) the three/four-launch loops are faster
    return nrhs == 1 ? plan.kind == 5 : (plan.kind == 6 && plan.wide);
}
template <typename T>
static int spmv_fused_impl(const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                           const void *d_old, void *d_new, const void *r, void *q, int nrhs, void *part_dq, const void *part_rr, int P,
                           const CgScalars &sc, hipStream_t st) {
    using A = typename VT<T>::acc;
    SpmvArgs<T> a;
    a.n = n; a.nrhs = nrhs; a.nnz = nnz;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<const T *>(d_old); a.ldx = n;
    a.y = static_cast<T *>(q); a.ldy = n;
    a.dvec = nullptr; a.partials = static_cast<A *>(part_dq);
    a.row_blocks = plan.row_blocks; a.rb_list = nullptr; a.rb_count = 0;
    a.cap = (plan.max_span + 3) & ~3;
    a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
    FusedArgs<T> f;
    f.r = static_cast<const T *>(r); f.dnew = static_cast<T *>(d_new);
    f.part_rr = static_cast<const A *>(part_rr); f.P = P;
    f.delta = (T *)sc.delta; f.beta = (T *)sc.beta; f.history = (T *)sc.history; f.history_cap = sc.history_cap; f.iter = sc.iter;
    const dim3 g(rowblock_grid(plan.row_blocks, a.cycle), nrhs), b(kBlock);
    const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
    const size_t lds = (size_t)a.cap * (sizeof(T) + 4);
    constexpr int U = sizeof(T) > 8 ? 4 : 8;
    if (nt) hipLaunchKernelGGL((spmv_fused_kernel<T, kBlock, true, U>), g, b, lds, st, a, f);
    else hipLaunchKernelGGL((spmv_fused_kernel<T, kBlock, false, U>), g, b, lds, st, a, f);
    return check_launch("spmv_fused");
}
int launch_spmv_fused(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                      const void *d_old, void *d_new, const void *r, void *q, int nrhs, void *part_dq, const void *part_rr, int P,
                      const CgScalars &sc, hipStream_t st) {
    CG_DISPATCH(dtype, spmv_fused_impl, plan, n, nnz, vals, ptr, cols, d_old, d_new, r, q, nrhs, part_dq, part_rr, P, sc, st);
}
template <typename T> static int cg_tail_impl(const void *part_rr, int P, int nrhs, const CgScalars &sc, hipStream_t st) {
    using A = typename VT<T>::acc;
    FusedArgs<T> f;
    f.r = nullptr; f.dnew = nullptr; f.part_rr = static_cast<const A *>(part_rr); f.P = P;
    f.delta = (T *)sc.delta; f.beta = (T *)sc.beta; f.history = (T *)sc.history; f.history_cap = sc.history_cap; f.iter = sc.iter;
    hipLaunchKernelGGL((cg_tail_kernel<T, kBlock>), dim3(nrhs), dim3(kBlock), 0, st, f, nrhs);
    return check_launch("cg_tail");
}
int launch_cg_tail(int dtype, const void *part_rr, int P, int nrhs, const CgScalars &sc, hipStream_t st) {
    CG_DISPATCH(dtype, cg_tail_impl, part_rr, P, nrhs, sc, st);
}

// ---- diagonally preconditioned CG -----------------------------------------------------------------
template <typename T>
static int pcg_axpy2_impl(bool init, int n, const void *d, void *x, const void *q, void *r, const void *m, long long ld,
                          const void *alpha, int nrhs, void *part_rz, void *part_rr, int grid, bool vec, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
#define CG_PCG(V, I) hipLaunchKernelGGL((pcg_axpy2_dot2_kernel<T, kBlock, V, I>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, \
                                        (T *)r, (const T *)m, ld, (const T *)alpha, (A *)part_rz, (A *)part_rr)
    if (init) { if (vec) CG_PCG(true, true); else CG_PCG(false, true); }
    else { if (vec) CG_PCG(true, false); else CG_PCG(false, false); }
#undef CG_PCG
    return check_launch("pcg_axpy2_dot2");
}
int launch_pcg_axpy2_dot2(int dtype, bool init, int n, const void *d, void *x, const void *q, void *r, const void *m,
                          long long ld, const void *alpha, int nrhs, void *part_rz, void *part_rr, int grid, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {d, x, q, r, m});
    CG_DISPATCH(dtype, pcg_axpy2_impl, init, n, d, x, q, r, m, ld, alpha, nrhs, part_rz, part_rr, grid, vec, st);
}
template <typename T>
static int pcg_aypx_impl(int n, const void *r, void *p, const void *m, long long ld, const void *part_rz, const void *part_rr,
                         int P, int nrhs, const CgScalars &sc, void *rho2, void *xs, bool vec, hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
    if (vec) hipLaunchKernelGGL((pcg_aypx_beta_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)r, (T *)p, (const T *)m, ld, (const A *)part_rz, (const A *)part_rr, P, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, (T *)rho2, (const int *)sc.iter, (T *)xs, (const T *)sc.alpha);
    else hipLaunchKernelGGL((pcg_aypx_beta_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)r, (T *)p, (const T *)m, ld, (const A *)part_rz, (const A *)part_rr, P, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, (T *)rho2, (const int *)sc.iter, (T *)xs, (const T *)sc.alpha);
    return check_launch("pcg_aypx_beta");
}
int launch_pcg_aypx_beta(int dtype, int n, const void *r, void *p, const void *m, long long ld, const void *part_rz,
                         const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, void *xs, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {r, p, m, xs});
    CG_DISPATCH(dtype, pcg_aypx_impl, n, r, p, m, ld, part_rz, part_rr, P, nrhs, sc, rho2, xs, vec, st);
}
template <typename T>
static int pcg_delta0_impl(const void *part_rz, const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, hipStream_t st) {
    using A = typename VT<T>::acc;
    hipLaunchKernelGGL((pcg_delta0_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, (const A *)part_rz, (const A *)part_rr, P, nrhs,
                       (T *)sc.delta, (T *)sc.history, (T *)rho2, sc.iter);
    return check_launch("pcg_delta0");
}
int launch_pcg_delta0(int dtype, const void *part_rz, const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, hipStream_t st) {
    CG_DISPATCH(dtype, pcg_delta0_impl, part_rz, part_rr, P, nrhs, sc, rho2, st);
}

// ---- peer-to-peer backend launchers ---------------------------------------------------------------
static P2pExchangeArgs p2p_args(const P2pExchange &e) {
    P2pExchangeArgs a;
    a.mailbox = e.mailbox; a.rank = e.rank; a.n_peers = e.n_peers; a.n_local = e.n_local;
    a.peer_rank = e.peer_rank; a.send_off = e.send_off; a.send_count = e.send_count; a.dst_off = e.dst_off;
    a.recv_off = e.recv_off; a.recv_count = e.recv_count; a.send_index = e.send_index; a.epoch = e.epoch;
    a.counters = e.counters; a.max_count = e.max_count;
    return a;
}
int p2p_push_chunks(const P2pExchange &e) { return e.max_count > 0 ? (e.max_count + kP2pChunk - 1) / kP2pChunk : 1; }
template <typename T>
static int spmv_p2p_impl(const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                         const void *d_ext, void *q, void *partials, const int *halo_flag, int rotate, const P2pExchange &e,
                         hipStream_t st) {
    SpmvP2pArgs<T> g;
    SpmvArgs<T> &a = g.s;
    a.n = n; a.nrhs = 1; a.nnz = nnz;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<const T *>(d_ext); a.ldx = 0;
    a.y = static_cast<T *>(q); a.ldy = 0;
    a.dvec = static_cast<const T *>(d_ext);
    a.partials = static_cast<typename VT<T>::acc *>(partials);
    a.row_blocks = plan.row_blocks; a.rb_list = halo_flag; a.rb_count = plan.row_blocks;
    a.cap = (plan.max_span + 3) & ~3;
    a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
    g.x = p2p_args(e);
    g.halo = static_cast<const T *>(e.my_halo);
    g.n_local = e.n_local; g.rotate = rotate; g.push_chunks = p2p_push_chunks(e);
    const bool coded = plan.codes && plan.codes_for == cols && tune().index_codes != 0;
    a.codes = coded ? plan.codes : nullptr;
    a.dict = coded ? plan.dict : nullptr;
    const size_t lds = coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
    const int grid = rowblock_grid(plan.row_blocks, a.cycle);
    if (grid < e.n_peers * g.push_chunks) return fail(CGAMD_ERR_STATE, "spmv_p2p: fewer work-groups than push chunks");
    const dim3 gd(grid), block(kBlock);
    const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
    constexpr int U = sizeof(T) > 8 ? 4 : 8;
    // batch length of the row walk follows the longest row, as in spmv_impl (5 / 7: one batch per row of a 5- / 7-point stencil)
    const int fit = (sizeof(T) > 8 || tune().spmv_unroll) ? U : plan.max_row == 5 ? 5 : (plan.max_row == 6 || plan.max_row == 7) ? 7 : U;
#define CG_P2P(UU)                                                                                                      \
    do {                                                                                                                \
        if (coded) {                                                                                                    \
            if (nt) hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, true, UU, true>), gd, block, lds, st, g);   \
            else hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, false, UU, true>), gd, block, lds, st, g);     \
        } else if (nt) hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, true, UU>), gd, block, lds, st, g);      \
        else hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, false, UU>), gd, block, lds, st, g);               \
    } while (0)
    if (fit == 5) CG_P2P(5);
    else if (fit == 7) CG_P2P(7);
    else CG_P2P(U);
#undef CG_P2P
    return check_launch("spmv_rowblock_p2p");
}
int launch_spmv_p2p(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                    const void *d_ext, void *q, void *partials, const int *halo_flag, int rotate, const P2pExchange &e,
                    hipStream_t st) {
    if (plan.kind != 5 || !aligned16(vals) || !aligned16(cols)) return fail(CGAMD_ERR_STATE, "spmv_p2p: needs the row-block kernel");
    CG_DISPATCH(dtype, spmv_p2p_impl, plan, n, nnz, vals, ptr, cols, d_ext, q, partials, halo_flag, rotate, e, st);
}
// work-groups the four-launch SpMV runs (they carry the push chunks)
int spmv_p2p_grid(const SpmvPlan &plan) { return rowblock_grid(plan.row_blocks, tune().spmv_cycle > 0 ? tune().spmv_cycle : 1); }

// Every work-group of aypx_beta_p2p_kernel spins until all ranks' r.r slots have arrived, and this rank's own slot is
// published by work-group 0 of the same launch: the launch is only safe if the whole grid is resident at once (a queued
// work-group 0 behind spinning ones would never publish).  The grid is therefore capped at what the occupancy query
// admits, minus one work-group per CU (the hardware can admit one fewer than the API reports: MI355X_MICROARCH.md
// "Residency and cooperative launch"); the kernel is grid-stride, so a smaller grid only changes who updates what.
template <typename K> static int resident_grid_cap(K kernel) {
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, kBlock, 0) != hipSuccess) return 0;
    if (per > 8) per = 8;
    if (per > 1) per -= 1;
    return cus * per;
}
template <typename T>
static int aypx_beta_p2p_impl(int n, const void *x, void *y, void *xs, const void *partials, int P, char *const *mailbox, int rank,
                              int nranks, int which, const unsigned long long *epoch, const CgScalars &sc, bool vec, int vnt, hipStream_t st) {
    auto *pp = static_cast<const typename VT<T>::acc *>(partials);
#define CG_AP(V, N)                                                                                                          \
    do {                                                                                                                      \
        static const int cap = resident_grid_cap(aypx_beta_p2p_kernel<T, kBlock, V, N>);                                     \
        if (cap < 1) return fail(CGAMD_ERR_HIP, "aypx_beta_p2p: occupancy query failed; refusing an all-work-group spin");    \
        const dim3 g(std::min(vec_grid(n, VT<T>::dtype), cap)), blk(kBlock);                                                 \
        hipLaunchKernelGGL((aypx_beta_p2p_kernel<T, kBlock, V, N>), g, blk, 0, st, n, (const T *)x, (T *)y, (T *)xs,         \
                           (const T *)sc.alpha, pp, P, mailbox, rank, nranks, which, epoch, (T *)sc.delta, (T *)sc.beta,     \
                           (T *)sc.history, sc.history_cap, (const int *)sc.iter);                                           \
    } while (0)
    if (vec && (vnt & 1)) CG_AP(true, 1); else if (vec) CG_AP(true, 0); else CG_AP(false, 0);
#undef CG_AP
    return check_launch("aypx_beta_p2p");
}
int launch_aypx_beta_p2p(int dtype, int n, const void *x, void *y, void *xs, const void *partials, int P, char *const *mailbox,
                         int rank, int nranks, int which, const unsigned long long *epoch, const CgScalars &sc, hipStream_t st,
                         int vec_nt) {
    if (n <= 0) return CGAMD_OK;
    if (nranks > 64) return fail(CGAMD_ERR_INVALID, "p2p all-reduce: at most 64 ranks");
    const bool v = vec_ok(dtype, n, 1, {x, y, xs});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, aypx_beta_p2p_impl, n, x, y, xs, partials, P, mailbox, rank, nranks, which, epoch, sc, v, vnt, st);
}

template <typename T> static int p2p_exchange_impl(const P2pExchange &e, void *v_ext, hipStream_t st) {
    P2pExchangeArgs a = p2p_args(e);
    const dim3 g(p2p_push_chunks(e), e.n_peers);
    hipLaunchKernelGGL((p2p_push_kernel<T>), g, dim3(kP2pBlock), 0, st, a, (const T *)v_ext);
    hipLaunchKernelGGL((p2p_wait_unpack_kernel<T>), g, dim3(kP2pBlock), 0, st, a, (T *)v_ext);
    return check_launch("p2p_exchange");
}
int launch_p2p_exchange(int dtype, const P2pExchange &e, void *v_ext, hipStream_t st) {
    if (e.n_peers <= 0) return CGAMD_OK;
    CG_DISPATCH(dtype, p2p_exchange_impl, e, v_ext, st);
}
template <typename T>
static int p2p_ar_impl(int mode, const void *partials, int grid, char *const *mailbox, int rank, int nranks, int which,
                       unsigned long long *epoch, const CgScalars &sc, unsigned long long *bump0, unsigned long long *bump1,
                       hipStream_t st) {
    auto *pp = static_cast<const typename VT<T>::acc *>(partials);
#define CG_AR(M)                                                                                                            \
    hipLaunchKernelGGL((p2p_allreduce_kernel<T, M>), dim3(1), dim3(kScalarBlock), 0, st, pp, grid, mailbox, rank, nranks, which, \
                       epoch, (T *)sc.delta, (T *)sc.alpha, (T *)sc.beta, (T *)sc.history, sc.history_cap, sc.iter, bump0, bump1)
    if (mode == 1) CG_AR(1); else if (mode == 2) CG_AR(2); else CG_AR(3);
#undef CG_AR
    return check_launch("p2p_allreduce");
}
int launch_p2p_allreduce(int dtype, int mode, const void *partials, int grid, char *const *mailbox, int rank, int nranks,
                         int which, unsigned long long *epoch, const CgScalars &sc, hipStream_t st, unsigned long long *bump0,
                         unsigned long long *bump1) {
    if (nranks > 64) return fail(CGAMD_ERR_INVALID, "p2p all-reduce: at most 64 ranks");
    CG_DISPATCH(dtype, p2p_ar_impl, mode, partials, grid, mailbox, rank, nranks, which, epoch, sc, bump0, bump1, st);
}

}  // namespace cgamd
