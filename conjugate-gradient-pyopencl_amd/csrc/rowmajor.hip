// Row-major multi-RHS path (BASELINE config 4, "MFMA tall-B tile path"): the block of right-hand sides is kept as
// X[N][R] (element i of RHS r at i*R + r) through the whole CG loop, so that the operand gathered for one non-zero,
// X[col][0:R], is one dense contiguous row (64...256 B) and the product over a strip of rows is a real contraction on the
// matrix cores.  Replaces reference kernel/{real,complex}/spmv.cl with N_RHS > 1 (+ vdot.cl, axpy.cl, aypx.cl on the
// same layout); the reference's ABI layout (RHS-major, spmv.cl:25,48) is converted once per solve at the boundary
// (set_rhs / get_x), not per iteration.
//
//   spmm_rm_kernel        Y = A X (+ per-RHS d.q partials), CSR A, f32 / f64 / complex64, R*(1|2) in {16, 32, 64} real columns
//   rm_dot / rm_axpy_dot / rm_aypx_x   the vector kernels of the fused loop with per-column scalars
//
// SpMM design (MI355X).  Measured problem of the round-1 kernels (profiles/r1/pmc_spmm_c4_summary.txt): X was fetched
// 2.75x because each XCD had 256 work-groups x 256 rows = 16 MB of X rows open at once, far beyond its 4 MB L2, so a
// line brought in for row i was gone before rows i +- nx asked for it.  Here:
//   * the unit of work is a STRIP of 16 rows handled by ONE wave; waves are persistent and every XCD sweeps its own
//     contiguous eighth of the strips in order (wave w of the XCD takes strips w, w + W, w + 2W, ...), so the rows open
//     per XCD are W x 16 (W = waves per XCD; 512 -> 2 MB of fp64 X rows at R = 32): the +-nx reuse falls inside L2;
//   * a wave keeps a whole strip's gathers in flight (up to UB K-steps of 4 non-zeros = 64 lanes x 16 B each), and
//     prefetches the NEXT strip's row pointers and matrix entries (registers -> wave-private LDS, no work-group barrier
//     anywhere in the loop), so a strip costs one memory round trip, not three;
//   * fp64 uses v_mfma_f64_4x4x4_4b (4 blocks = 4 quads of right-hand sides, 4 rows x 4 non-zeros each): the selection
//     matrix S[row][k] = a_k if non-zero k lies in that row has ONE non-zero per column, so the 16x16x4 form wastes
//     15/16 of a 64-cycle instruction, the 4x4x4 form 3/4 of a ~20-cycle one (scripts/microbench/mfma_probe.hip:
//     29.8 ns vs 9.9 ns per instruction, lane maps A[b][i][k] @ 16k+4b+i, B[b][k][j] @ 16k+4b+j, D[b][i][j] @ 16i+4b+j);
//     f32 and complex64 use v_mfma_f32_16x16x4 (32 cycles);
//   * column permutation: lane (k = l>>4, m = l&15) loads NH consecutive values X[col_k][NH*m .. NH*m+NH-1] with one
//     8/16-byte load and MFMA h uses component h, so MFMA h produces the real columns {NH*m + h}: every load
//     instruction of a wave covers 4 whole X rows and every store instruction 4 whole Y rows;
//   * complex64: the row is 2R interleaved floats; a lane holds whole (re, im) pairs, so Y += a_re * X + a_im * X' with
//     X' = (-im, re) formed in registers: two real accumulations per K-step, no cross-lane traffic;
//   * accumulation order inside a row is CSR order (K-slots are consecutive non-zeros; other rows contribute exact zeros);
//   * fused d.q: the strip's own X rows are loaded once more (L1/L2 hits), per-lane column sums live in registers for the
//     whole sweep and leave as ONE partial per work-group and RHS (fixed order -> bitwise reproducible).
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"

#include <algorithm>

namespace cgamd {

template <typename T, int NH> struct alignas(NH * sizeof(T)) RowVec { T v[NH]; };
template <typename T, int NH> CG_DEV RowVec<T, NH> ld_rowvec(const T *p) { return *reinterpret_cast<const RowVec<T, NH> *>(p); }
template <bool NT, typename T, int NH> CG_DEV void st_rowvec(T *p, const RowVec<T, NH> &v) {
    if constexpr (!NT) {
        *reinterpret_cast<RowVec<T, NH> *>(p) = v;
    } else {
        constexpr int W = NH * (int)sizeof(T) / 4;
        typedef unsigned uvec __attribute__((ext_vector_type(W)));
        union { RowVec<T, NH> r; uvec u; unsigned s; } cv;
        cv.r = v;
        if constexpr (W == 1) __builtin_nontemporal_store(cv.s, reinterpret_cast<unsigned *>(p));
        else __builtin_nontemporal_store(cv.u, reinterpret_cast<uvec *>(p));
    }
}

// 16-byte store with an explicit cache policy: 0 plain, 1 non-temporal, 2 sc1 (write-through; the line does not stay in
// the XCD's L2 -- MI355X_MICROARCH.md "stores of each flavour")
CG_DEV void st16_policy(void *p, const void *src, int policy) {
    const u32x4 d = *reinterpret_cast<const u32x4 *>(src);
    if (policy == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
    else if (policy == 1) __builtin_nontemporal_store(d, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = d;
}

CG_DEV f32x4 mfma_f32_16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
CG_DEV double mfma_f64_4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <typename T> struct SpmmRmArgs {
    int n, strips, nwg, ynt;
    long long nnz;
    const T *vals;          // nnz values (complex: 2 nnz interleaved re, im)
    const int *ptr, *cols;
    const T *x;             // [columns][RC] row-major, RC = 16 NH real columns
    T *y;                   // [n][RC]
    double *partials;       // fused d.q: real [RC][nwg] doubles; complex [RC/2][nwg] (re, im) pairs
    int *pace;              // [8][kPaceWaves] BYTES: steps finished per wave of an XCD's sweep, cumulative mod 256; null = unpaced
    int lead;               // a wave gathers step g only when every wave of its XCD has finished g - lead steps
};
constexpr int kPaceWaves = 256;

template <typename T, int NH, int TQ, int NQ, bool CPLX>
struct RmGeom {
    static constexpr bool F64 = sizeof(T) == 8;
    static constexpr int BV = NQ * TQ * NH * (int)sizeof(T) / 4;      // registers holding one strip's gathers
    static constexpr bool PIPE = BV <= 80;                            // two strips in flight fit 256 VGPRs
    static constexpr int WAVES = (PIPE ? 2 * BV : BV) <= 64 ? 4 : 2;  // waves per SIMD the instance is built for
};

template <typename T, int NH, int TQ, int NQ, bool CPLX, bool FUSE_DOT>
__global__ __launch_bounds__(256, (RmGeom<T, NH, TQ, NQ, CPLX>::WAVES)) void spmm_rm_kernel(SpmmRmArgs<T> a) {
    using G = RmGeom<T, NH, TQ, NQ, CPLX>;
    constexpr bool F64 = G::F64, PIPE = G::PIPE;
    static_assert(!(F64 && CPLX), "complex128 runs the RHS-major kernel");
    static_assert(F64 || NQ == 4, "the fp32 tile is the 16-row strip");
    static_assert(!CPLX || NH % 2 == 0, "a lane must hold whole (re, im) pairs");
    constexpr int ROWS = 4 * NQ, RC = 16 * NH, QS = 4 * TQ, SLOTS = NQ * QS, EPL = (SLOTS + 63) / 64;
    constexpr int VW = CPLX ? 2 : 1;                        // value words per entry
    constexpr int VS = SLOTS + 4;                           // fp64: bank-conflict-free stride between the 4 operand copies
    constexpr int SW = CPLX ? 4 : 2;                        // fp32: words per staged slot (value[, imag], row[, pad])
    constexpr int LDSV = F64 ? 4 * VS * 2 : SLOTS * SW;     // 32-bit words of the value area per wave and buffer
    using RV = RowVec<T, NH>;
    __shared__ int scol[4][2][SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned sraw[4][2][LDSV];
    __shared__ double red[4][2 * RC];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 15, kq = lane >> 4;
    const int xcd = blockIdx.x & 7, W = ((int)gridDim.x >> 3) * 4, wl = ((int)blockIdx.x >> 3) * 4 + wave;
    const int sb = (int)((long long)xcd * a.strips / 8), se = (int)((long long)(xcd + 1) * a.strips / 8);

    // Paced sweep.  The strips are handed out statically (the fused d.q partial of a work-group must not depend on the schedule),
    // so nothing but the L2's patience keeps a fast wave from running steps ahead of a slow one, and an X row that three strips
    // 62 apart use (5-point stencil on a 1000-wide grid) was fetched again for the stragglers: FETCH 1.26-1.45x (profiles/r2).
    // Every wave publishes the number of steps it has finished in its own byte (one shared counter bumped with atomics cost
    // ~100 ns per bump, serialised at the memory side -- 7 800 strips per XCD = 0.9 ms); before a wave issues the gathers
    // of step g it wants every wave of its XCD to have finished g - lead steps.  All the words (bytes) of an XCD arrive with ONE 4-byte
    // load per lane, issued a step early, before that step's gathers, so that waiting for it never waits for gathers (vmcnt
    // retires in order); only a wave that finds itself ahead polls.  The kernel ends with the slowest wave either way: waiting
    // costs the leaders nothing that counts.  The wait is ADVISORY (results never depend on it) and bounded: a wave not answered
    // within kPacePolls polls (grid not co-resident) stops pacing for the rest of the launch.  The words are cumulative over
    // launches (every wave leaves base + MX, MX = steps of the longest wave), so nothing has to be cleared in between.
    constexpr int kPacePolls = 256;
    const int S = se - sb, MX = (S + W - 1) / W;
    unsigned char *const pw = a.pace ? reinterpret_cast<unsigned char *>(a.pace) + kPaceWaves * xcd : nullptr;     // one BYTE per wave (mod 256)
    bool paced = pw != nullptr && W <= kPaceWaves && S >= W;
    const int base = pw && wl < kPaceWaves ? __builtin_amdgcn_readfirstlane((int)pw[wl]) : 0;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pw, 0, pw ? kPaceWaves : 0, 0x00020000);
    int stepno = 0;                                         // step of s_cur
    auto pace_load = [&]() -> unsigned { return __builtin_amdgcn_raw_buffer_load_b32(prs, (unsigned)lane * 4u, 0, 16 /* sc1: from the L2 */); };
    auto pace_ok = [&](unsigned v, int need) -> bool {      // (waves are never 128 steps apart while the pacing holds: mod-256 compare)
        const int lo = 4 * lane, t = base + need;           // bytes past the sweep's last wave do not count
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) ok = ok && (lo + k >= W || (signed char)(unsigned char)((v >> (8 * k)) - (unsigned)t) >= 0);
        return __builtin_amdgcn_ballot_w64(!ok) == 0;
    };
    unsigned pv = 0u;
    auto pace_wait = [&](int g) {                           // before the gathers of step g
        if (!paced) return;
        const int need = g - a.lead;
        if (need > 0 && !pace_ok(pv, need)) {               // pv: the words as they were a step ago
            int polls = 0;
            while (!pace_ok(pace_load(), need)) {
                if (++polls > kPacePolls) { paced = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        pv = pace_load();                                   // for the next step's check; older than the gathers issued next
    };
    auto pace_done = [&](int steps) {
        // PLAIN store: it lands in the XCD's L2, where the other waves' sc1 loads find it.  A write-through (sc1 / agent-scope)
        // store of a byte is a read-modify-write of the line at the memory side, ~100 ns each and serialised per line: 7 800 of
        // them per XCD made the launch 8x longer (scripts/microbench/sload_probe.hip)
        if (pw && wl < kPaceWaves && lane == 0) asm volatile("global_store_byte %0, %1, off" ::"v"(pw + wl), "v"(base + steps) : "memory");
    };

    double dsum[NH], dcross[CPLX ? NH : 1];
#pragma unroll
    for (int h = 0; h < NH; ++h) dsum[h] = 0.;
#pragma unroll
    for (int h = 0; h < (CPLX ? NH : 1); ++h) dcross[h] = 0.;

    auto load_ptr = [&](int s) -> int {
        const int row = s * ROWS + (lane < ROWS ? lane : ROWS);
        return a.ptr[row < a.n ? row : a.n];
    };
    struct Quads { int tb, Q[NQ + 1]; };                      // quad starts relative to the strip's first entry (uniform)
    auto quads = [&](int p) -> Quads {
        Quads g;
        g.tb = __builtin_amdgcn_readlane(p, 0);
        g.Q[0] = 0;
#pragma unroll
        for (int q = 1; q <= NQ; ++q) g.Q[q] = __builtin_amdgcn_readlane(p, 4 * q) - g.tb;
        return g;
    };
    // slot idx = e 64 + lane of the staging area <-> (quad, position); entry j of the strip it holds in round r
    auto slot_entry = [&](const Quads &g, int idx, int r, int &q, int &Qq, int &Qn) -> int {
        q = idx / QS;
        Qq = 0; Qn = g.Q[1];
#pragma unroll
        for (int k = 1; k < NQ; ++k) { Qq = q == k ? g.Q[k] : Qq; Qn = q == k ? g.Q[k + 1] : Qn; }
        return Qq + r + (idx - q * QS);
    };
    auto fetch = [&](int p, int r, T (&ev)[EPL][VW], int (&ec)[EPL]) {
        const Quads g = quads(p);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int q, Qq, Qn;
            const int idx = e * 64 + lane, j = slot_entry(g, idx, r, q, Qq, Qn);
            if (idx < SLOTS && j < Qn) {
                ec[e] = __builtin_nontemporal_load(a.cols + (long long)g.tb + j);
                if constexpr (CPLX) {
                    const f32x2 w = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(a.vals) + (long long)g.tb + j);
                    ev[e][0] = w.x; ev[e][1] = w.y;
                } else {
                    ev[e][0] = __builtin_nontemporal_load(a.vals + (long long)g.tb + j);
                }
            }
        }
    };
    auto stage = [&](int buf, int p, int r, int safe_col, const T (&ev)[EPL][VW], const int (&ec)[EPL]) {
        const Quads g = quads(p);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int q, Qq, Qn;
            const int idx = e * 64 + lane, j = slot_entry(g, idx, r, q, Qq, Qn);
            // row of the entry inside its quad: rows 4q+1 .. 4q+3 start at R1 <= R2 <= R3 (all lanes shuffle: outside the branch)
            const int qc = idx < SLOTS ? q : 0;
            const int R1 = __shfl(p, 4 * qc + 1, 64) - g.tb, R2 = __shfl(p, 4 * qc + 2, 64) - g.tb, R3 = __shfl(p, 4 * qc + 3, 64) - g.tb;
            if (idx < SLOTS) {       // every slot is written every round: a live entry, or zeros + a harmless column
                const bool live = j < Qn;
                const int ri = live ? (j >= R1) + (j >= R2) + (j >= R3) : -1;
                scol[wave][buf][idx] = live ? ec[e] : safe_col;
                if constexpr (F64) {
                    double *sv = reinterpret_cast<double *>(sraw[wave][buf]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sv[k * VS + idx] = (k == ri) ? ev[e][0] : 0.;
                } else {
                    unsigned *sv = sraw[wave][buf] + idx * SW;
#pragma unroll
                    for (int w = 0; w < VW; ++w) sv[w] = live ? __float_as_uint(ev[e][w]) : 0u;
                    sv[VW] = live ? (unsigned)(4 * q + ri) : 255u;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // one wave's LDS ops execute in order; compiler ordering only
    };
    auto steps_of = [&](const Quads &g, int q, int r) -> int {      // K-steps of quad q in round r (uniform)
        return min(max((g.Q[q + 1] - g.Q[q] - r + 3) >> 2, 0), TQ);
    };
    // the gathers of one round: NQ TQ loads of NH values per lane, each instruction = 4 whole X rows.  Branch-free, LDS
    // offsets are immediates (slots past a quad's last K-step hold `safe_col`: an L1 hit that the multiply never consumes),
    // addresses are uniform base + 32-bit byte offset (the launcher guarantees the block is < 4 GiB)
    const char *xbase = reinterpret_cast<const char *>(a.x) + NH * m * sizeof(T);
    auto issue = [&](int buf, RV (&bv)[NQ][TQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int u = 0; u < TQ; ++u) {
                const unsigned c = (unsigned)scol[wave][buf][q * QS + 4 * u + kq];
                bv[q][u] = *reinterpret_cast<const RV *>(xbase + (size_t)(c * (unsigned)(RC * sizeof(T))));
            }
        }
    };
    // accumulators of one strip: fp64 one per quad and column group (4x4x4 blocks), fp32 one 16x16 tile per column group
    struct Acc {
        double d[F64 ? NQ : 1][NH];
        f32x4 f[F64 ? 1 : NH];
    };
    // fp64: K-step u of all quads before K-step u + 1: consecutive MFMAs go to NQ NH different accumulators, so none waits
    // for its predecessor's result (with the quad loop outside, each accumulator chain stalled the next issue:
    // SQ_WAIT_INST_ANY 38 % of the wave cycles).  fp32: consecutive MFMAs alternate between the NH tiles.
    auto multiply = [&](int buf, int p, int r, const RV (&bv)[NQ][TQ], Acc &acc) {
        const Quads g = quads(p);
        int st[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) st[q] = steps_of(g, q, r);
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            if constexpr (F64) {
                const double *sv = reinterpret_cast<const double *>(sraw[wave][buf]);
                double av[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) av[q] = sv[(lane & 3) * VS + q * QS + 4 * u + kq];
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (u < st[q]) acc.d[q][h] = mfma_f64_4(av[q], bv[q][u].v[h], acc.d[q][h]);     // wave-uniform guard
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (u < st[q]) {                                                                     // wave-uniform guard
                        const unsigned *sv = sraw[wave][buf] + (q * QS + 4 * u + kq) * SW;
                        const bool mine = (int)sv[VW] == m;
                        const float ar = mine ? __uint_as_float(sv[0]) : 0.f;
#pragma unroll
                        for (int h = 0; h < NH; ++h) acc.f[h] = mfma_f32_16(ar, bv[q][u].v[h], acc.f[h]);
                        if constexpr (CPLX) {
                            // (ar + i ai)(xr + i xi): the re column gets -ai xi, the im column +ai xr (cmplx.h:20-25)
                            const float ai = mine ? __uint_as_float(sv[1]) : 0.f;
#pragma unroll
                            for (int h = 0; h < NH; ++h) {
                                const float xs = (h & 1) ? bv[q][u].v[h - 1] : -bv[q][u].v[h + 1];
                                acc.f[h] = mfma_f32_16(ai, xs, acc.f[h]);
                            }
                        }
                    }
                }
            }
        }
    };
    // local row of output slot i (i = 0..3) of this lane: fp64 D[b][i][j] @ 16 i + 4 b + j -> quad i, row l >> 4;
    // fp32 16x16 tile: row 4 (l >> 4) + i
    auto lrow = [&](int i) -> int { return F64 ? 4 * i + kq : 4 * kq + i; };

    const int s0 = sb + wl;
    if (s0 >= se) pace_done(MX);
    if (s0 < se) {
        auto clampS = [&](int s) -> int { return s < se ? s : se - 1; };
        int s_cur = s0, s_nxt = s0 + W, s_nn = s0 + 2 * W;
        int p_cur = load_ptr(s_cur), p_nxt = load_ptr(clampS(s_nxt)), p_nn = load_ptr(clampS(s_nn));
        int buf = 0;
        T evN[EPL][VW]; int ecN[EPL];                           // entries of the next strip, fetched one step ahead
        RV bvA[NQ][TQ], bvB[PIPE ? NQ : 1][PIPE ? TQ : 1];
        {
            T ev[EPL][VW]; int ec[EPL];
            fetch(p_cur, 0, ev, ec);
            stage(0, p_cur, 0, min(s_cur * ROWS, a.n - 1), ev, ec);
        }
        fetch(p_nxt, 0, evN, ecN);
        if constexpr (PIPE) issue(0, bvA);

        // one pipeline step: strip s_cur is multiplied out of bvC while the next strip's gathers go into bvN
        auto step = [&](RV (&bvC)[NQ][TQ], auto &bvN) -> bool {
            const int rowbase = s_cur * ROWS;
            RV xo[F64 ? NQ : 4];
            if (FUSE_DOT) {
#pragma unroll
                for (int i = 0; i < (F64 ? NQ : 4); ++i)
                    xo[i] = *reinterpret_cast<const RV *>(xbase + (size_t)((unsigned)min(rowbase + lrow(i), a.n - 1) * (unsigned)(RC * sizeof(T))));
            }
            if constexpr (!PIPE) { pace_wait(stepno); issue(buf, bvC); }
            stage(buf ^ 1, p_nxt, 0, min(clampS(s_nxt) * ROWS, a.n - 1), evN, ecN);
            const int p_n3 = load_ptr(clampS(s_nn + W));
            fetch(p_nn, 0, evN, ecN);
            if constexpr (PIPE) { pace_wait(stepno + 1); issue(buf ^ 1, bvN); }

            Acc acc;
            if constexpr (F64) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc.d[q][h] = 0.;
            } else {
#pragma unroll
                for (int h = 0; h < NH; ++h) acc.f[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            multiply(buf, p_cur, 0, bvC, acc);
            {   // quads longer than one round (more than 4 TQ non-zeros in 4 rows): further rounds, not pipelined
                const Quads g = quads(p_cur);
                int maxlen = 0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) maxlen = max(maxlen, g.Q[q + 1] - g.Q[q]);
                for (int r = QS; r < maxlen; r += QS) {
                    T ev[EPL][VW]; int ec[EPL];
                    fetch(p_cur, r, ev, ec);
                    stage(buf, p_cur, r, min(rowbase, a.n - 1), ev, ec);
                    issue(buf, bvC);
                    multiply(buf, p_cur, r, bvC, acc);
                }
            }
            // store: every store instruction of the wave covers 4 whole rows of Y
#pragma unroll
            for (int i = 0; i < (F64 ? NQ : 4); ++i) {
                const int row = rowbase + lrow(i);
                RV out;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if constexpr (F64) out.v[h] = acc.d[i][h];
                    else out.v[h] = acc.f[h][i];
                }
                if (row < a.n) {
                    char *yp = reinterpret_cast<char *>(a.y) + NH * m * sizeof(T) + (size_t)((unsigned)row * (unsigned)(RC * sizeof(T)));
                    if constexpr (sizeof(RV) == 16) st16_policy(yp, &out, a.ynt);     // wave-uniform policy
                    else *reinterpret_cast<RV *>(yp) = out;
                    if (FUSE_DOT) {
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            dsum[h] += (double)(xo[i].v[h] * out.v[h]);
                            if constexpr (CPLX) dcross[h] += (double)(xo[i].v[h] * out.v[h ^ 1]);
                        }
                    }
                }
            }
            // the strip's X rows are consumed (the multiply waited for them); a wave's last publish is MX for every wave
            pace_done(s_nxt >= se ? MX : stepno + 1);
            if (s_nxt >= se) return false;
            ++stepno;
            s_cur = s_nxt; s_nxt = s_nn; s_nn += W;
            p_cur = p_nxt; p_nxt = p_nn; p_nn = p_n3;
            buf ^= 1;
            return true;
        };
        if constexpr (PIPE) {
            while (true) {
                if (!step(bvA, bvB)) break;
                if (!step(bvB, bvA)) break;
            }
        } else {
            while (step(bvA, bvA)) {}
        }
    }
    if (FUSE_DOT) {
        // per-lane column sums -> per-wave (lanes with equal m) -> per work-group, fixed order
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            dsum[h] += __shfl_xor(dsum[h], 16, 64);
            dsum[h] += __shfl_xor(dsum[h], 32, 64);
            if constexpr (CPLX) {
                dcross[h] += __shfl_xor(dcross[h], 16, 64);
                dcross[h] += __shfl_xor(dcross[h], 32, 64);
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                red[wave][NH * m + h] = dsum[h];
                if constexpr (CPLX) red[wave][RC + NH * m + h] = dcross[h];
            }
        }
        __syncthreads();
        auto tot = [&](int c) { return ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]; };
        if constexpr (CPLX) {
            if (t < RC / 2) {
                const double re = tot(2 * t) - tot(2 * t + 1);                     // unconjugated (complex/vdot.cl:15)
                const double im = tot(RC + 2 * t) + tot(RC + 2 * t + 1);
                double *p = a.partials + 2 * ((long long)t * a.nwg + blockIdx.x);
                p[0] = re; p[1] = im;
            }
        } else {
            if (t < RC) a.partials[(long long)t * a.nwg + blockIdx.x] = tot(t);
        }
    }
}

// =================================================================================================
// Vector kernels on the row-major block: T = float | double | float2 | double2, R right-hand sides, element (i, r) at
// i*R + r.  A thread's 16-byte packs always fall on the same columns (the grid stride is a multiple of R), so the
// per-column scalars are loaded once and the per-column dot partials accumulate in registers.  Column sums: lanes with
// equal columns meet by xor-shuffles, the 4 waves through LDS, one partial per work-group and column (fixed order).
// =================================================================================================
template <typename A> CG_DEV A shfl_xor_acc(A v, int off);
template <> CG_DEV double shfl_xor_acc<double>(double v, int off) { return __shfl_xor(v, off, 64); }
template <> CG_DEV double2 shfl_xor_acc<double2>(double2 v, int off) { return make_double2(__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64)); }

template <typename T, int BLOCK>
CG_DEV void rm_column_partials(typename VT<T>::acc (&acc)[Pack<T>::N], int R, typename VT<T>::acc *partials, typename VT<T>::acc *red /* [4][R] */) {
    using A = typename VT<T>::acc;
    constexpr int E = Pack<T>::N;
    const int groups = R / E;                                   // distinct column groups among the lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = groups; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < E; ++k) acc[k] = vadd(acc[k], shfl_xor_acc<A>(acc[k], off));
    if (lane < groups)
#pragma unroll
        for (int k = 0; k < E; ++k) red[wave * R + lane * E + k] = acc[k];
    __syncthreads();
    if ((int)threadIdx.x < R) {
        const int c = threadIdx.x;
        A tot = red[c];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) tot = vadd(tot, red[w * R + c]);
        partials[(long long)c * gridDim.x + blockIdx.x] = tot;
    }
}

// partials of a.b per column (setup: r.r)
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void rm_dot_kernel(long long total, int R, const T *__restrict__ a, const T *__restrict__ b,
                                                        typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    constexpr int E = Pack<T>::N;
    __shared__ A red[(BLOCK / 64) * 64];
    A acc[E];
#pragma unroll
    for (int k = 0; k < E; ++k) acc[k] = vzero<A>();
    const long long npack = total / E, stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < npack; i += stride) {
        const Pack<T> pa = ld_pack(a + i * E), pb = ld_pack(b + i * E);
#pragma unroll
        for (int k = 0; k < E; ++k) acc[k] = vadd(acc[k], to_acc(vmul(pa.v[k], pb.v[k])));
    }
    rm_column_partials<T, BLOCK>(acc, R, partials, red);
}

// r -= alpha[c] q ; partials of r.r per column      (reference axpy.cl with aSign = 0 + vdot.cl; clcg.c:345-374)
template <typename T, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void rm_axpy_dot_kernel(long long total, int R, const T *__restrict__ q, T *__restrict__ rv,
                                                             const T *__restrict__ alpha, typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    constexpr int E = Pack<T>::N;
    __shared__ A red[(BLOCK / 64) * 64];
    const int c0 = (int)(((long long)threadIdx.x * E) % R);
    T al[E];
    A acc[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { al[k] = alpha[c0 + k]; acc[k] = vzero<A>(); }
    const long long npack = total / E, stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < npack; i += stride) {
        const Pack<T> pq = NT ? ld_pack_nt(q + i * E) : ld_pack(q + i * E);      // q: last use of the iteration
        Pack<T> pr = ld_pack(rv + i * E);
#pragma unroll
        for (int k = 0; k < E; ++k) {
            pr.v[k] = vsub(pr.v[k], vmul(al[k], pq.v[k]));
            acc[k] = vadd(acc[k], to_acc(vmul(pr.v[k], pr.v[k])));
        }
        st_pack(rv + i * E, pr);
    }
    rm_column_partials<T, BLOCK>(acc, R, partials, red);
}

// x += alpha[c] d ; d = beta[c] d + r                (axpy.cl with aSign = 1, aypx.cl; clcg.c:338-342,415)
template <typename T, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void rm_aypx_x_kernel(long long total, int R, const T *__restrict__ rv, T *__restrict__ d,
                                                           T *__restrict__ xs, const T *__restrict__ alpha, const T *__restrict__ beta) {
    constexpr int E = Pack<T>::N;
    const int c0 = (int)(((long long)threadIdx.x * E) % R);
    T al[E], bt[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { al[k] = alpha[c0 + k]; bt[k] = beta[c0 + k]; }
    const long long npack = total / E, stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < npack; i += stride) {
        const Pack<T> pr = ld_pack(rv + i * E);
        Pack<T> pd = ld_pack(d + i * E), px = NT ? ld_pack_nt(xs + i * E) : ld_pack(xs + i * E);   // x: touched once per iteration
#pragma unroll
        for (int k = 0; k < E; ++k) {
            px.v[k] = vadd(px.v[k], vmul(al[k], pd.v[k]));
            pd.v[k] = vaypx(bt[k], pd.v[k], pr.v[k]);
        }
        if (NT) st_pack_nt(xs + i * E, px); else st_pack(xs + i * E, px);
        st_pack(d + i * E, pd);
    }
}

// =================================================================================================
// Host side
// =================================================================================================
static int rm_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return CGAMD_OK;
}

// real columns of the row-major block (0 = this type / width has no matrix-core path)
static int rm_real_columns(int dtype, int nrhs) {
    if (dtype == CGAMD_C128) return 0;
    if (dtype == CGAMD_C64) return (nrhs == 16 || nrhs == 32) ? 2 * nrhs : 0;     // a lane must hold whole (re, im) pairs
    if (dtype == CGAMD_F64) return (nrhs == 16 || nrhs == 32) ? nrhs : 0;         // 64 fp64 columns: a strip's gathers exceed the registers
    return (nrhs == 16 || nrhs == 32 || nrhs == 64) ? nrhs : 0;
}
// the kernels address X and Y with 32-bit byte offsets: the block must stay below 4 GiB
bool spmm_rm_supported(int dtype, int nrhs, int n) {
    return rm_real_columns(dtype, nrhs) != 0 && (unsigned long long)n * nrhs * dtype_size(dtype) < (1ULL << 32);
}

template <typename K> static int rm_blocks_per_cu(K kernel) {
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, 0) != hipSuccess || per < 1) per = 1;
    return per > 8 ? 8 : per;
}
// which instance runs: K-steps per quad and round (5 when no 4 consecutive rows hold more than 20 non-zeros -- 5-point
// stencils --, else 8).  Strips are 16 rows (8-row strips at four waves per SIMD were measured and dropped: 3 % faster only
// with 96 of the 128 resident work-groups per XCD, slower inside CG; profiles/r2_experiments/spmm_ab9.log, spmm_ab10.log).
struct RmInstance { int tq, nq, rows, per_cu; };
template <typename T, int NH, bool CPLX> static RmInstance rm_instance(int max_quad, bool dot) {
    RmInstance r;
    const bool tq5 = max_quad > 0 && max_quad <= 20;
    r.tq = tq5 ? 5 : 8; r.nq = 4; r.rows = 4 * r.nq;
#define CG_OCC(TQ, NQ) (dot ? rm_blocks_per_cu(spmm_rm_kernel<T, NH, TQ, NQ, CPLX, true>) : rm_blocks_per_cu(spmm_rm_kernel<T, NH, TQ, NQ, CPLX, false>))
    r.per_cu = tq5 ? CG_OCC(5, 4) : CG_OCC(8, 4);
#undef CG_OCC
    return r;
}
static int rm_grid_for(int n, int rows, int per_cu) {
    const int strips = (n + rows - 1) / rows;
    const int cap = tune().spmm_wgs > 0 ? tune().spmm_wgs : 32 * per_cu;
    int per_xcd = ((strips + 7) / 8 + 3) / 4;
    if (per_xcd > cap) per_xcd = cap;
    if (per_xcd < 1) per_xcd = 1;
    return 8 * per_xcd;
}
// dispatch on (dtype, nrhs): F(T, NH, CPLX)
#define CG_RM_TYPES(dtype, rc, F)                                                                        \
    do {                                                                                                  \
        if (dtype == CGAMD_F64) { if (rc == 16) F(double, 1, false); F(double, 2, false); }               \
        if (dtype == CGAMD_F32) { if (rc == 16) F(float, 1, false); if (rc == 32) F(float, 2, false); F(float, 4, false); } \
        if (rc == 32) F(float, 2, true);      /* complex64, 16 right-hand sides */                        \
        F(float, 4, true);                                                                                \
    } while (0)

// Work-groups of the sweep (a multiple of 8; 4 waves each) = fused-dot partials per RHS.  The sweep only keeps its locality
// if every work-group of the grid is resident from the start (a queued work-group would run its interleaved strips after
// the others have moved on), so the grid is the instance's resident capacity (occupancy query x 32 CUs per XCD;
// `spmm_wgs` overrides), never more waves than strips.
int spmm_rm_grid(int dtype, int nrhs, int n, int max_quad, bool dot) {
    const int rc = rm_real_columns(dtype, nrhs);
    if (!rc) return 8;
#define CG_GRID(T, NH, C) do { const RmInstance i = rm_instance<T, NH, C>(max_quad, dot); return rm_grid_for(n, i.rows, i.per_cu); } while (0)
    CG_RM_TYPES(dtype, rc, CG_GRID);
#undef CG_GRID
}

template <typename T, int NH, bool CPLX>
static int spmm_rm_launch(int n, long long nnz, const void *vals, const int *ptr, const int *cols, const void *x, void *y,
                          void *partials, int max_quad, int *pace, hipStream_t st) {
    SpmmRmArgs<T> a;
    a.n = n; a.nnz = nnz;
    a.lead = tune().spmm_lead > 0 ? tune().spmm_lead : 3;
    a.pace = (tune().spmm_lead < 0 || !pace) ? nullptr : pace + (partials ? 0 : 8 * kPaceWaves / 4);
    a.ynt = tune().spmm_ynt >= 0 ? tune().spmm_ynt : 2;      // Y stores write-through (sc1): the lines do not displace X in L2
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<const T *>(x); a.y = static_cast<T *>(y); a.partials = static_cast<double *>(partials);
    const RmInstance inst = rm_instance<T, NH, CPLX>(max_quad, partials != nullptr);
    a.strips = (n + inst.rows - 1) / inst.rows;
    a.nwg = rm_grid_for(n, inst.rows, inst.per_cu);
    const dim3 g(a.nwg), b(256);
#define CG_SPMM(TQ, NQ)                                                                                       \
    do {                                                                                                       \
        if (partials) hipLaunchKernelGGL((spmm_rm_kernel<T, NH, TQ, NQ, CPLX, true>), g, b, 0, st, a);         \
        else hipLaunchKernelGGL((spmm_rm_kernel<T, NH, TQ, NQ, CPLX, false>), g, b, 0, st, a);                 \
    } while (0)
    if (inst.tq == 5) CG_SPMM(5, 4); else CG_SPMM(8, 4);
#undef CG_SPMM
    return rm_check_launch("spmm_rm");
}

int launch_spmm_rm(int dtype, int n, long long nnz, const void *vals, const int *ptr, const int *cols, const void *x, void *y,
                   int nrhs, void *partials, int max_quad, int *pace, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const int rc = rm_real_columns(dtype, nrhs);
    if (rc && !spmm_rm_supported(dtype, nrhs, n)) return fail(CGAMD_ERR_INVALID, "spmm_rowmajor: the right-hand-side block must be smaller than 4 GiB");
    if (!rc) return fail(CGAMD_ERR_INVALID, "spmm_rowmajor: needs f64 with 16 or 32 right-hand sides, f32 with 16, 32 or 64, or complex64 with 16 or 32");
#define CG_RM(T, NH, C) return spmm_rm_launch<T, NH, C>(n, nnz, vals, ptr, cols, x, y, partials, max_quad, pace, st)
    CG_RM_TYPES(dtype, rc, CG_RM);
#undef CG_RM
}

// grid of the row-major vector kernels: ~4 packs per thread.  NOT capped at the resident 2048 work-groups like the
// single-vector kernels: a block of 16-64 right-hand sides is 16-64 vectors long, and with 2048 work-groups every thread
// walked 16+ packs one after the other (one iteration's loads in flight per thread) -- the RHS-major loop, which launches
// nRHS x 2048 work-groups, ran its vector part 25 % faster.  One partial per work-group and column: cg_alpha/cg_beta sum them.
int rm_vec_grid(long long total_elems, int dtype) {
    const long long per_block = (long long)kBlock * (16 / (long long)dtype_size(dtype)) * 4;
    long long g = (total_elems + per_block - 1) / per_block;
    const long long cap = tune().vec_grid > 0 ? tune().vec_grid : 16 * kMaxGrid;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static bool rm_vec_ok(int dtype, int nrhs, std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if (!aligned16(p)) return false;
    const int e = 16 / (int)dtype_size(dtype);
    return nrhs >= e && nrhs <= 64 && (nrhs & (nrhs - 1)) == 0;      // 256 E is a multiple of R; columns per pack divide R
}

#define CG_RM_DISPATCH(dtype, FN, ...)                                      \
    switch (dtype) {                                                        \
    case CGAMD_F32: return FN<float>(__VA_ARGS__);                          \
    case CGAMD_F64: return FN<double>(__VA_ARGS__);                         \
    case CGAMD_C64: return FN<float2>(__VA_ARGS__);                         \
    case CGAMD_C128: return FN<double2>(__VA_ARGS__);                       \
    default: return fail(CGAMD_ERR_INVALID, "bad dtype");                   \
    }

template <typename T> static int rm_dot_impl(long long total, int R, const void *a, const void *b, void *partials, int grid, hipStream_t st) {
    hipLaunchKernelGGL((rm_dot_kernel<T, kBlock>), dim3(grid), dim3(kBlock), 0, st, total, R, (const T *)a, (const T *)b,
                       (typename VT<T>::acc *)partials);
    return rm_check_launch("rm_dot");
}
int launch_rm_dot(int dtype, int n, int nrhs, const void *a, const void *b, void *partials, int grid, hipStream_t st) {
    if (!rm_vec_ok(dtype, nrhs, {a, b})) return fail(CGAMD_ERR_INVALID, "rm_dot: unsupported width or misaligned vectors");
    CG_RM_DISPATCH(dtype, rm_dot_impl, (long long)n * nrhs, nrhs, a, b, partials, grid, st);
}
template <typename T> static int rm_axpy_dot_impl(long long total, int R, const void *q, void *r, const void *alpha, void *partials, int grid, hipStream_t st) {
    // streaming hints once the block is far beyond the 256 MB Infinity Cache (as the single-vector kernels do: Tuning::vec_nt)
    const bool nt = tune().vec_nt >= 0 ? (tune().vec_nt & 2) != 0 : (size_t)total * sizeof(T) > ((size_t)96 << 20);
    if (nt) hipLaunchKernelGGL((rm_axpy_dot_kernel<T, kBlock, true>), dim3(grid), dim3(kBlock), 0, st, total, R, (const T *)q, (T *)r, (const T *)alpha,
                               (typename VT<T>::acc *)partials);
    else hipLaunchKernelGGL((rm_axpy_dot_kernel<T, kBlock, false>), dim3(grid), dim3(kBlock), 0, st, total, R, (const T *)q, (T *)r, (const T *)alpha,
                            (typename VT<T>::acc *)partials);
    return rm_check_launch("rm_axpy_dot");
}
int launch_rm_axpy_dot(int dtype, int n, int nrhs, const void *q, void *r, const void *alpha, void *partials, int grid, hipStream_t st) {
    if (!rm_vec_ok(dtype, nrhs, {q, r})) return fail(CGAMD_ERR_INVALID, "rm_axpy_dot: unsupported width or misaligned vectors");
    CG_RM_DISPATCH(dtype, rm_axpy_dot_impl, (long long)n * nrhs, nrhs, q, r, alpha, partials, grid, st);
}
template <typename T> static int rm_aypx_x_impl(long long total, int R, const void *r, void *d, void *x, const void *alpha, const void *beta, int grid, hipStream_t st) {
    const bool nt = tune().vec_nt >= 0 ? (tune().vec_nt & 1) != 0 : (size_t)total * sizeof(T) > ((size_t)96 << 20);
    if (nt) hipLaunchKernelGGL((rm_aypx_x_kernel<T, kBlock, true>), dim3(grid), dim3(kBlock), 0, st, total, R, (const T *)r, (T *)d, (T *)x, (const T *)alpha,
                               (const T *)beta);
    else hipLaunchKernelGGL((rm_aypx_x_kernel<T, kBlock, false>), dim3(grid), dim3(kBlock), 0, st, total, R, (const T *)r, (T *)d, (T *)x, (const T *)alpha,
                            (const T *)beta);
    return rm_check_launch("rm_aypx_x");
}
int launch_rm_aypx_x(int dtype, int n, int nrhs, const void *r, void *d, void *x, const void *alpha, const void *beta, int grid, hipStream_t st) {
    if (!rm_vec_ok(dtype, nrhs, {r, d, x})) return fail(CGAMD_ERR_INVALID, "rm_aypx_x: unsupported width or misaligned vectors");
    CG_RM_DISPATCH(dtype, rm_aypx_x_impl, (long long)n * nrhs, nrhs, r, d, x, alpha, beta, grid, st);
}

}  // namespace cgamd
