// CSR SpMV / SpMM kernels of the CG hot path (replace reference kernel/{real,complex}/spmv.cl), hand-written for gfx950.
//
//   spmv_rowblock_kernel          SpMV fused with the d.q partial reduction (reference vdot.cl + host sum clcg.c:317-324): matrix
//                                 slice through LDS, one lane per row; _chunked: 2/4/8 lanes per row for denser rows.
//                                 POL = -3: the column indices arrive as one-byte codes (aCols[j] = row + dict[code[j]],
//                                 index_codes.hip): 9 instead of 12 B per fp64 non-zero
//   spmm_rowblock_kernel          the same for nRHS > 1 (RHS-major, the ABI layout); the row-major matrix-core path is rowmajor.hip
//   spmv_fused_kernel             two-launch iteration of small systems: beta / d = beta d + r at the head of the SpMV launch
//   spmv_stream_kernel            generic chunked CSR stream (huge rows, unaligned pointers)
//
// Design (MI355X): bound by the memory system.  Work-groups are 256 threads (4 wave64), one per 256-row block, dealt
// block-cyclically over the 8 XCDs.  Loads are 16 B per lane and every load instruction of a wave covers contiguous memory;
// matrix streams are non-temporal unless the matrix fits the Infinity Cache.
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"
#include "spmv_device.h"
#include "reduce_device.h"
#include "launch_util.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

namespace cgamd {

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
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int UNROLL, int POL = -2>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);                       // [cap]
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));   // [cap] (coded form: cap BYTES)
    __shared__ A red[BLOCK / kWave];
    constexpr bool CODED = POL == -3, CODED16 = POL == -4;
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
    [[maybe_unused]] const int cbase = CODED16 ? a.dict[rb] : 0;      // first column of the row block (16-bit codes are relative to it)
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
            else if constexpr (CODED16) cj[j] = cbase + reinterpret_cast<const unsigned short *>(sc)[idx];
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
// The FULLY CODED form of the row-block kernel: one-byte column codes AND one-byte value codes (matrices with at most 256 distinct
// entries, build_value_codes) -- 2 bytes per non-zero from HBM instead of sizeof(T) + 1.  The value dictionary sits in LDS beside
// the offset dictionary; a slot is code byte -> ds_read of the value, code byte -> ds_read of the offset.  Same row walk, same
// order of fused multiply-adds on the same bits as every other form of the kernel: bit-identical results.
// -------------------------------------------------------------------------------------------------
// With 2 bytes per non-zero a 256-row block is ~4 KB of matrix: a work-group that owned ONE block would live for a single chain
// of round trips (row pointers -> codes -> gathers) and the kernel is bound by the latency of the gathers.  So a work-group walks
// kVcBlocks consecutive row blocks, TWO at a time: every thread walks one row of each block of the pair (twice the gathers in flight
// per thread: 92.9 -> see DESIGN.md), and the next pair's code dwords travel (registers) while the current pair is multiplied out of
// LDS (two LDS buffers; the row pointers of the pair after that are loaded a step earlier still).  One d.q partial per 256-row
// block, formed exactly as block_sum<256> forms it, as everywhere.
constexpr int kVcBlocks = 4;      // row blocks per work-group
constexpr int kVcRows = 1;        // of which a thread walks this many at a time (one row of each): 2 were measured slower (108 VGPRs)
template <typename T> CG_DEV T buf_gather(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    if constexpr (sizeof(T) == 4) {
        const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0);
        T v; __builtin_memcpy(&v, &w, 4); return v;
    } else {
        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
        T v; __builtin_memcpy(&v, &w, 8); return v;
    }
}
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int UNROLL>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_vc_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    constexpr int PR = kVcRows;          // rows per thread = row blocks walked at a time
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    // [2 buffers][PR blocks][column codes cap | value codes cap]
    __shared__ A red[PR][BLOCK / kWave];
    __shared__ int sdict[BLOCK];
    __shared__ T sdictv[BLOCK];
    static_assert(BLOCK == 256 && kVcBlocks % PR == 0, "one dictionary entry per thread; whole groups of row blocks");
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    sdict[t] = a.dict[t] * (int)sizeof(T);     // byte offsets into x; visible after the first staging barrier
    sdictv[t] = a.vdict[t];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(a.x), 0, 0x7ffffffc, 0x00020000);      // (no bound of its own: the columns were validated, and a rank's x carries halo entries behind its n rows)
    const int supers = (a.row_blocks + kVcBlocks - 1) / kVcBlocks;
    const int srb = rowblock_of(blockIdx.x, supers, a.cycle);
    if (srb < 0) return;
    const int rb0 = srb * kVcBlocks, nb = min(kVcBlocks, a.row_blocks - rb0);

    struct Bounds { int s_raw, e_raw, p0, p1; };
    auto bounds_of = [&](int rb) -> Bounds {          // branch-free: clamped rows (a block past the matrix: empty)
        const int r0 = min(rb, a.row_blocks) * BLOCK, rclamp = min(r0 + t, a.n - 1);
        Bounds b;
        b.s_raw = a.ptr[rclamp]; b.e_raw = a.ptr[rclamp + 1];
        b.p0 = a.ptr[min(r0, a.n)]; b.p1 = a.ptr[min(r0 + BLOCK, a.n)];
        return b;
    };
    // the code dwords of a block: two rounds of 4 BLOCK entries cover the 8 BLOCK entries a slice may have (the host checked cap)
    struct Codes { unsigned cw[2], vw[2]; };
    auto load_codes = [&](const Bounds &b) -> Codes {
        Codes c;
        const int cfirst = b.p0 & ~3;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const long long q = (long long)cfirst + rg * 4 * BLOCK + 4 * t;
            const long long qq = q < b.p1 ? q : cfirst;         // (a valid address either way: unconditional loads)
            const unsigned *cp = reinterpret_cast<const unsigned *>(a.codes + qq), *vp = reinterpret_cast<const unsigned *>(a.vcodes + qq);
            c.cw[rg] = NT ? __builtin_nontemporal_load(cp) : *cp;
            c.vw[rg] = NT ? __builtin_nontemporal_load(vp) : *vp;
        }
        return c;
    };
    Bounds bc[PR], bn[PR];
    Codes cc[PR];
#pragma unroll
    for (int h = 0; h < PR; ++h) { bc[h] = bounds_of(rb0 + h); bn[h] = bounds_of(rb0 + PR + h); }
#pragma unroll
    for (int h = 0; h < PR; ++h) cc[h] = load_codes(bc[h]);
    for (int i = 0; i < nb; i += PR) {
        unsigned char *buf = reinterpret_cast<unsigned char *>(dyn_smem) + (size_t)((i / PR) & 1) * 2 * PR * a.cap;
        int cfirst[PR];
#pragma unroll
        for (int h = 0; h < PR; ++h) {
            cfirst[h] = bc[h].p0 & ~3;
            unsigned char *scc = buf + (size_t)h * 2 * a.cap, *svc = scc + a.cap;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                const int o = rg * 4 * BLOCK + 4 * t;
                if (cfirst[h] + o < bc[h].p1) {
                    *reinterpret_cast<unsigned *>(scc + o) = cc[h].cw[rg];
                    *reinterpret_cast<unsigned *>(svc + o) = cc[h].vw[rg];
                }
            }
        }
        __syncthreads();
        // the next pair's codes and the row pointers of the pair after it: in flight behind this pair's gathers
        Bounds bcur[PR];
#pragma unroll
        for (int h = 0; h < PR; ++h) bcur[h] = bc[h];
        if (i + PR < nb) {
#pragma unroll
            for (int h = 0; h < PR; ++h) { cc[h] = load_codes(bn[h]); bc[h] = bn[h]; bn[h] = bounds_of(rb0 + i + 2 * PR + h); }
        }
        int row[PR], s[PR], e[PR];
        T dv[PR];
#pragma unroll
        for (int h = 0; h < PR; ++h) {
            row[h] = (rb0 + i + h) * BLOCK + t;
            const bool live = row[h] < a.n && i + h < nb;
            s[h] = bcur[h].s_raw - cfirst[h];
            e[h] = live ? bcur[h].e_raw - cfirst[h] : s[h];
            if (FUSE_DOT) dv[h] = a.dvec[min(row[h], a.n - 1)];      // early: it need not wait for the gathers
        }
        // Row walk with as few instructions as the form allows (the kernel is bound by instruction issue at 7 waves per SIMD, not by
        // bytes): code bytes at immediate LDS offsets from the row's first entry (unclamped: slots past the row's end read the next
        // row's codes -- valid LDS -- and are masked), byte offsets of x from a pre-multiplied dictionary, gathers as buffer loads
        // (base in scalar registers + 32-bit offset: no 64-bit address arithmetic); a masked slot gathers x[row] and its term is
        // dropped by the same select as in every other form of the kernel.
        T sum[PR];
        int len[PR], kmax = 0;
        unsigned row8[PR];
#pragma unroll
        for (int h = 0; h < PR; ++h) {
            sum[h] = vzero<T>();
            len[h] = e[h] - s[h];
            kmax = max(kmax, len[h]);
            row8[h] = (unsigned)min(row[h], a.n - 1) * (unsigned)sizeof(T);
        }
        // waves whose rows all have exactly UNROLL entries (the interior of a stencil: about half the waves of the 7-point Laplacian on
        // a 250-wide grid) take the walk without a single mask: no compare, no select -- the same multiply-adds in the same order
        bool full = true;
#pragma unroll
        for (int h = 0; h < PR; ++h) full = full && len[h] == UNROLL;
        if (__builtin_amdgcn_ballot_w64(!full) == 0) {
            T xv[PR][UNROLL], av[PR][UNROLL];
            unsigned off[PR][UNROLL];
#pragma unroll
            for (int h = 0; h < PR; ++h) {
                const unsigned char *pc = buf + (size_t)h * 2 * a.cap + s[h], *pv = pc + a.cap;
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) {
                    off[h][j] = pc[j];
                    av[h][j] = sdictv[pv[j]];
                }
            }
#pragma unroll
            for (int h = 0; h < PR; ++h)
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) off[h][j] = row8[h] + (unsigned)sdict[off[h][j]];
#pragma unroll
            for (int h = 0; h < PR; ++h)
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) xv[h][j] = buf_gather<T>(xrs, off[h][j]);
#pragma unroll
            for (int h = 0; h < PR; ++h)
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) sum[h] = vfma(av[h][j], xv[h][j], sum[h]);
        } else
        for (int k = 0; k < kmax; k += UNROLL) {
            T xv[PR][UNROLL], av[PR][UNROLL];
            unsigned off[PR][UNROLL];
#pragma unroll
            for (int h = 0; h < PR; ++h) {
                const unsigned char *pc = buf + (size_t)h * 2 * a.cap + s[h] + k, *pv = pc + a.cap;
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) {
                    off[h][j] = pc[j];
                    av[h][j] = sdictv[pv[j]];
                }
            }
#pragma unroll
            for (int h = 0; h < PR; ++h)
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) off[h][j] = k + j < len[h] ? row8[h] + (unsigned)sdict[off[h][j]] : row8[h];
#pragma unroll
            for (int h = 0; h < PR; ++h)
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) xv[h][j] = buf_gather<T>(xrs, off[h][j]);
#pragma unroll
            for (int h = 0; h < PR; ++h)
#pragma unroll
                for (int j = 0; j < UNROLL; ++j) {
                    const T nxt = vfma(av[h][j], xv[h][j], sum[h]);
                    sum[h] = vsel(k + j < len[h], nxt, sum[h]);
                }
        }
        A dot1[PR];
#pragma unroll
        for (int h = 0; h < PR; ++h) dot1[h] = vzero<A>();
#pragma unroll
        for (int h = 0; h < PR; ++h)
            if (row[h] < a.n && i + h < nb) {
                a.y[row[h]] = sum[h];
                if (FUSE_DOT) dot1[h] = to_acc(vmul(dv[h], sum[h]));
            }
        if (FUSE_DOT) {                     // block_sum<256> of both blocks behind one pair of barriers (same tree, same order)
#pragma unroll
            for (int h = 0; h < PR; ++h) dot1[h] = wave_sum(dot1[h]);
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int h = 0; h < PR; ++h) red[h][wave] = dot1[h];
            }
            __syncthreads();
            if (t < PR && i + t < nb) {
                A v = red[t][0];
#pragma unroll
                for (int w = 1; w < BLOCK / kWave; ++w) v = vadd(v, red[t][w]);
                a.partials[rb0 + i + t] = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// The same with the gathers PIPELINED across the row blocks of a work-group (matrices whose longest row fits one batch: max_row <=
// UNROLL).  PMC of the form above (gpurun pmc_bench_vc): waves spend 71 % of their cycles in s_waitcnt, the VALU is 36 % busy -- the
// kernel waits for one gather round trip per row block.  Here the gathers of block i + 1 are issued (codes out of LDS, offsets, 7
// buffer loads) BEFORE block i is multiplied and stored, so two blocks' gathers are in flight per wave and a block's round trip
// hides behind its predecessor's arithmetic, store and block sum.  vmcnt retires in order: waiting for block i's gathers does not
// wait for block i + 1's.  The code dwords of block i + 2 travel in registers meanwhile.  LDS buffers alternate; a buffer is only
// read by issue() and every issue() lies between two barriers that separate it from the writes of the same buffer.
// -------------------------------------------------------------------------------------------------
// JOINT: a.codes holds one byte per non-zero naming the (offset, value) PAIR (build_joint_codes; a.dict / a.vdict are the pair
// dictionaries): one code stream, 1 byte per non-zero; a.vcodes is not read
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int UNROLL, bool JOINT>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_vcp_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];       // [2 buffers][column codes cap | value codes cap]
    __shared__ A red[BLOCK / kWave];
    __shared__ int sdict[BLOCK];
    __shared__ T sdictv[BLOCK];
    static_assert(BLOCK == 256, "one dictionary entry per thread");
    const int t = threadIdx.x;
    sdict[t] = a.dict[t] * (int)sizeof(T);
    sdictv[t] = a.vdict[t];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(a.x), 0, 0x7ffffffc, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t xrs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(FUSE_DOT ? a.dvec : a.x), 0, 0x7ffffffc, 0x00020000);
    const int supers = (a.row_blocks + kVcBlocks - 1) / kVcBlocks;
    const int srb = rowblock_of(blockIdx.x, supers, a.cycle);
    if (srb < 0) return;
    const int rb0 = srb * kVcBlocks, nb = min(kVcBlocks, a.row_blocks - rb0);

    struct Bounds { int s_raw, e_raw, p0, p1; };
    auto bounds_of = [&](int rb) -> Bounds {          // branch-free: clamped rows (a block past the matrix: empty)
        const int r0 = min(rb, a.row_blocks) * BLOCK, rclamp = min(r0 + t, a.n - 1);
        Bounds b;
        b.s_raw = a.ptr[rclamp]; b.e_raw = a.ptr[rclamp + 1];
        b.p0 = a.ptr[min(r0, a.n)]; b.p1 = a.ptr[min(r0 + BLOCK, a.n)];
        return b;
    };
    struct Codes { unsigned cw[2], vw[2]; };
    auto load_codes = [&](const Bounds &b) -> Codes {
        Codes c;
        const int cfirst = b.p0 & ~3;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const long long q = (long long)cfirst + rg * 4 * BLOCK + 4 * t;
            const long long qq = q < b.p1 ? q : cfirst;
            const unsigned *cp = reinterpret_cast<const unsigned *>(a.codes + qq);
            c.cw[rg] = NT ? __builtin_nontemporal_load(cp) : *cp;
            if constexpr (!JOINT) {
                const unsigned *vp = reinterpret_cast<const unsigned *>(a.vcodes + qq);
                c.vw[rg] = NT ? __builtin_nontemporal_load(vp) : *vp;
            }
        }
        return c;
    };
    auto stage = [&](int i, const Bounds &b, const Codes &c) {
        unsigned char *scc = reinterpret_cast<unsigned char *>(dyn_smem) + (size_t)(i & 1) * 2 * a.cap, *svc = scc + a.cap;
        const int cfirst = b.p0 & ~3;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const int o = rg * 4 * BLOCK + 4 * t;
            if (cfirst + o < b.p1) {
                *reinterpret_cast<unsigned *>(scc + o) = c.cw[rg];
                if constexpr (!JOINT) *reinterpret_cast<unsigned *>(svc + o) = c.vw[rg];
            }
        }
    };
    // one block's gathers in flight: the values of x, the matrix values (out of the dictionary) and what the epilogue needs
    struct Gath { T xv[UNROLL]; unsigned vc[UNROLL]; T dv; int len, row; };      // (vc: the value codes; the dictionary is read when the block is multiplied)
    auto issue = [&](int i, const Bounds &b) -> Gath {
        Gath g;
        const unsigned char *scc = reinterpret_cast<const unsigned char *>(dyn_smem) + (size_t)(i & 1) * 2 * a.cap, *svc = JOINT ? scc : scc + a.cap;
        const int cfirst = b.p0 & ~3;
        g.row = (rb0 + i) * BLOCK + t;
        const bool live = g.row < a.n && i < nb;
        const int s = b.s_raw - cfirst;
        g.len = live ? b.e_raw - b.s_raw : 0;
        const unsigned row8 = (unsigned)min(g.row, a.n - 1) * (unsigned)sizeof(T);
        unsigned off[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            off[j] = scc[s + j];                     // unclamped: slots past the row's end read the next row's codes (valid LDS) and are masked
            g.vc[j] = svc[s + j];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) off[j] = j < g.len ? row8 + (unsigned)sdict[off[j]] : row8;
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) g.xv[j] = buf_gather<T>(xrs, off[j]);
        if (FUSE_DOT) g.dv = buf_gather<T>(xrs_d, row8);
        return g;
    };
    auto consume = [&](int i, const Gath &g) {
        T sum = vzero<T>();
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const T nxt = vfma(sdictv[g.vc[j]], g.xv[j], sum);
            sum = vsel(j < g.len, nxt, sum);
        }
        A dot1 = vzero<A>();
        if (g.row < a.n && i < nb) {
            a.y[g.row] = sum;
            if (FUSE_DOT) dot1 = to_acc(vmul(g.dv, sum));
        }
        if (FUSE_DOT) {
            const A tot = block_sum<BLOCK>(dot1, red);
            if (t == 0 && i < nb) a.partials[rb0 + i] = tot;
        }
    };
    static_assert(kVcBlocks == 4, "the pipeline below is written out for four row blocks");
    // two blocks' gathers in flight (three were measured slower: 88.8 against 80.9 us)
    Bounds b0 = bounds_of(rb0), b1 = bounds_of(rb0 + 1);
    Codes c0 = load_codes(b0);
    Bounds b2 = bounds_of(rb0 + 2);
    Codes c1 = load_codes(b1);
    stage(0, b0, c0);
    __syncthreads();
    Gath gA = issue(0, b0);
    // block 1 staged and gathered, block 0 multiplied
    Bounds b3 = bounds_of(rb0 + 3);
    Codes c2 = load_codes(b2);
    stage(1, b1, c1);
    __syncthreads();
    Gath gB = issue(1, b1);
    consume(0, gA);
    if (nb > 2) {                                  // (uniform; the last super block of the matrix may hold fewer row blocks)
        Codes c3 = load_codes(b3);
        stage(2, b2, c2);
        __syncthreads();
        gA = issue(2, b2);
        consume(1, gB);
        stage(3, b3, c3);
        __syncthreads();
        gB = issue(3, b3);
        consume(2, gA);
        consume(3, gB);
    } else {
        consume(1, gB);
    }
}

// -------------------------------------------------------------------------------------------------
// The same kernel for DENSER rows (a 256-row slice no longer fits LDS: > ~21 non-zeros per row in fp64).  The work-group
// still owns 256 rows and writes one d.q partial, but stages and walks them in LPR chunks of 256/LPR rows, LPR = 2, 4 or
// 8 lanes per row: lane l of a row takes entries s+l, s+l+LPR, ... (consecutive lanes -> consecutive entries -> for
// stencil / FE rows consecutive columns), the LPR partial sums meet in a shuffle tree.  The generic kernel, which these
// matrices used before, runs the 27-point stencil at 50 % of the HBM roofline.
// -------------------------------------------------------------------------------------------------
template <typename T, int BLOCK, bool NT, bool FUSE_DOT, int LPR, int UNROLL, int CODED = 0>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_chunked_kernel(SpmvArgs<T> a) {
    using A = typename VT<T>::acc;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));      // CODED: cap bytes of column codes
    __shared__ A red[BLOCK / kWave];
    __shared__ int sdict[CODED == 1 ? BLOCK : 1];
    constexpr int RC = BLOCK / LPR;                 // rows per chunk
    const int t = threadIdx.x, j = t / LPR, l = t % LPR;
    if constexpr (CODED == 1) sdict[t] = a.dict[t];      // visible after the first staging barrier
    const int rb = rowblock_of(blockIdx.x, a.row_blocks, a.cycle);
    if (rb < 0) return;
    [[maybe_unused]] const int cbase = CODED == 2 ? a.dict[rb] : 0;
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
        stage_slice<T, BLOCK, NT, CODED == 1 ? -3 : CODED == 2 ? -4 : -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc, a.codes);
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
                if constexpr (CODED == 1) cj[u] = reinterpret_cast<const unsigned char *>(sc)[idx];
                else if constexpr (CODED == 2) cj[u] = cbase + reinterpret_cast<const unsigned short *>(sc)[idx];
                else cj[u] = sc[idx];
                av[u] = sv[idx];
            }
            if constexpr (CODED == 1) {
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
    int P, K;                       // r.r partials per RHS; order of their sum (reduce_device.h thread_partials)
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
            acc = thread_partials<BLOCK>(f.part_rr + (long long)rhs * f.P, f.P, f.K);
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
    const A acc = thread_partials<BLOCK>(f.part_rr + (long long)r * f.P, f.P, f.K);
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
    // out[0]: span of BLOCK-row slices; out[1..5]: of BLOCK/2 ... BLOCK/32-row slices (chunked row-block kernel: 2 ... 32 lanes per row);
    // out[6]: most non-zeros in 4 consecutive rows starting at a multiple of 4 (row-major SpMM); out[7]: longest row
    int m[6] = {0, 0, 0, 0, 0, 0}, mq = 0, mr = 0;
    for (int rb = blockIdx.x * blockDim.x + threadIdx.x; rb < row_blocks; rb += gridDim.x * blockDim.x) {
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) {
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
    for (int lv = 0; lv < 6; ++lv)
        if (m[lv] > 0) atomicMax(out + lv, m[lv]);
    if (mq > 0) atomicMax(out + 6, mq);
    if (mr > 0) atomicMax(out + 7, mr);
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
    a.codes = nullptr; a.dict = nullptr; a.vcodes = nullptr; a.vdict = nullptr;
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
        // (the value stream is staged interleaved across the lanes in 16-byte chunks, stage_slice_ilv: N=10M SpMV f64 186 -> 165 us,
        // c64 186 -> 166, c128 435 -> 314, f32 109 -> 105; the plain staging has no instance any more)
        // one-byte column codes instead of aCols (build_index_codes; the codes belong to THIS cols array)
        const bool coded_any = nrhs == 1 && plan.codes && plan.codes_for == cols && tune().index_codes != 0;
        const bool coded = coded_any && !plan.codes16, coded16 = coded_any && plan.codes16;
        // ... and one-byte value codes on top (matrices of at most 256 distinct entries; complex128 has none)
        // (its staging covers slices of at most 8 x 256 entries: rows of 8 entries on average)
        const bool vcoded = coded && !rb_list && plan.vcodes && plan.vcodes_for == vals && tune().value_codes != 0 && a.cap <= 8 * kBlock && (unsigned long long)n * sizeof(T) < (1ULL << 30);
        a.codes = coded_any ? plan.codes : nullptr;
        a.dict = coded_any ? plan.dict : nullptr;      // (16-bit form: the first column of every row block)
        const size_t lds = coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15)
                                 : coded16 ? (((size_t)a.cap * (sizeof(T) + 2) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
#define CG_RB(NT, UNR)                                                                                                  \
    do {                                                                                                                \
        if (coded16) {                                                                                                  \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, true, UNR, -4>), g5, block, lds, st, a);        \
            else CG_LAUNCH_EV((spmv_rowblock_kernel<T, kBlock, NT, false, UNR, -4>), g5, block, lds, st, a);            \
        } else if (coded) {                                                                                             \
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
        if constexpr (sizeof(T) <= 8) {
            if (vcoded) {
                a.vcodes = plan.vcodes; a.vdict = static_cast<const T *>(plan.vdict);
                const size_t lds2 = (((size_t)a.cap * 4 * kVcRows + 15) & ~(size_t)15) + 16;      // two buffers x kVcRows blocks x two code streams (+ the unclamped reads' slack)
                const int supers = (plan.row_blocks + kVcBlocks - 1) / kVcBlocks, cyc = std::max(1, a.cycle / kVcBlocks);
                a.cycle = cyc;
                const dim3 gvc(rowblock_grid(supers, cyc));
                // rows that fit one batch (max_row <= unroll): the form with the gathers pipelined across the row blocks
                const bool pipe = plan.max_row > 0 && plan.max_row <= unroll && tune().dev_vc_pipe != 0;
                // ... and where at most 256 (offset, value) pairs occur: one joint code byte per non-zero
                const bool joint = pipe && plan.jcodes && tune().dev_joint_codes != 0;
                if (joint) { a.codes = plan.jcodes; a.dict = plan.jdict_off; a.vdict = static_cast<const T *>(plan.jdict_val); a.vcodes = nullptr; }
#define CG_VC(NT, UNR)                                                                                                  \
    do {                                                                                                                \
        if (joint) {                                                                                                    \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_vcp_kernel<T, kBlock, NT, true, UNR, true>), gvc, block, lds2, st, a); \
            else CG_LAUNCH_EV((spmv_rowblock_vcp_kernel<T, kBlock, NT, false, UNR, true>), gvc, block, lds2, st, a);     \
        } else if (pipe) {                                                                                              \
            if (fuse) CG_LAUNCH_EV((spmv_rowblock_vcp_kernel<T, kBlock, NT, true, UNR, false>), gvc, block, lds2, st, a); \
            else CG_LAUNCH_EV((spmv_rowblock_vcp_kernel<T, kBlock, NT, false, UNR, false>), gvc, block, lds2, st, a);    \
        } else if (fuse) CG_LAUNCH_EV((spmv_rowblock_vc_kernel<T, kBlock, NT, true, UNR>), gvc, block, lds2, st, a);     \
        else CG_LAUNCH_EV((spmv_rowblock_vc_kernel<T, kBlock, NT, false, UNR>), gvc, block, lds2, st, a);                \
    } while (0)
                if (unroll == 4) { if (nt) CG_VC(true, 4); else CG_VC(false, 4); }
                else if (unroll == 5) { if (nt) CG_VC(true, 5); else CG_VC(false, 5); }
                else if (unroll == 7) { if (nt) CG_VC(true, 7); else CG_VC(false, 7); }
                else { if (nt) CG_VC(true, 8); else CG_VC(false, 8); }
#undef CG_VC
                return check_launch("spmv_rowblock_vc");
            }
        }
        if (unroll == 4) { if (nt) CG_RB(true, 4); else CG_RB(false, 4); }
        else if (unroll == 5) { if (nt) CG_RB(true, 5); else CG_RB(false, 5); }
        else if (unroll == 7) { if (nt) CG_RB(true, 7); else CG_RB(false, 7); }
        else { if (nt) CG_RB(true, 8); else CG_RB(false, 8); }
#undef CG_RB
        return check_launch("spmv_rowblock");
    }
    if (vec && nrhs == 1 && plan.kind == 7 && !rb_list) {
        const int span = plan.chunk_span[plan.lpr == 2 ? 0 : plan.lpr == 4 ? 1 : plan.lpr == 8 ? 2 : plan.lpr == 16 ? 3 : 4];
        a.cap = (span + 3) & ~3;
        a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
        const bool coded_any = plan.codes && plan.codes_for == cols && tune().index_codes != 0;
        const bool coded = coded_any && !plan.codes16, coded16 = coded_any && plan.codes16;
        a.codes = coded_any ? plan.codes : nullptr;
        a.dict = coded_any ? plan.dict : nullptr;
        const size_t lds = coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15)
                                 : coded16 ? (((size_t)a.cap * (sizeof(T) + 2) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
        const dim3 g7(rowblock_grid(plan.row_blocks, a.cycle));
        const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
        constexpr int U = sizeof(T) > 8 ? 4 : 8;
#define CG_CH(NT, L)                                                                                                     \
    do {                                                                                                                  \
        if (coded16) {                                                                                                    \
            if (fuse) hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, true, L, U, 2>), g7, block, lds, st, a);      \
            else hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, false, L, U, 2>), g7, block, lds, st, a);          \
        } else if (coded) {                                                                                               \
            if (fuse) hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, true, L, U, 1>), g7, block, lds, st, a);      \
            else hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, false, L, U, 1>), g7, block, lds, st, a);          \
        } else if (fuse) hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, true, L, U>), g7, block, lds, st, a);   \
        else hipLaunchKernelGGL((spmv_rowblock_chunked_kernel<T, kBlock, NT, false, L, U>), g7, block, lds, st, a);       \
    } while (0)
        if (plan.lpr == 2) { if (nt) CG_CH(true, 2); else CG_CH(false, 2); }
        else if (plan.lpr == 4) { if (nt) CG_CH(true, 4); else CG_CH(false, 4); }
        else if (plan.lpr == 8) { if (nt) CG_CH(true, 8); else CG_CH(false, 8); }
        else if (plan.lpr == 16) { if (nt) CG_CH(true, 16); else CG_CH(false, 16); }
        else { if (nt) CG_CH(true, 32); else CG_CH(false, 32); }
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
    CG_HIP(hipMemsetAsync(scratch_dev, 0, 8 * sizeof(int), st));
    int g = (row_blocks + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL((spmv_span_kernel<kBlock>), dim3(g), dim3(256), 0, st, n, ptr_dev, row_blocks, scratch_dev);
    if (int rc = check_launch("spmv_span")) return rc;
    int spans[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CG_HIP(hipMemcpyAsync(spans, scratch_dev, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    plan->max_span = spans[0];
    plan->max_row = spans[7];
    for (int lv = 0; lv < 5; ++lv) plan->chunk_span[lv] = spans[lv + 1];
    plan->max_quad = spans[6];
    (void)cols_dev;
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
    int kind = tune().dev_generic_spmv ? 0 : 5;
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
            // (16 and 32 lanes per row -- chunks of 16 / 8 rows -- take rows of ~100 entries, the upstream report's m_t1 shape: the
            // generic kernel ran that at 6.5 % of the roofline, profiles/r3/configs_irregular.log)
            for (int lv = 0; lv < 5 && kind == 0; ++lv)
                if (plan->chunk_span[lv] > 0 && (size_t)plan->chunk_span[lv] * ebytes <= want) {
                    kind = 7;
                    plan->lpr = 2 << lv;
                }
            if (kind == 0 && plan->chunk_span[4] > 0 && (size_t)plan->chunk_span[4] * ebytes <= (size_t)kMaxChunkBytes) {
                kind = 7;
                plan->lpr = 32;
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

template <typename T> static int transpose_impl(int rows, int cols, const void *in, void *out, hipStream_t st) {
    dim3 g((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL((transpose_kernel<T>), g, dim3(256), 0, st, rows, cols, (const T *)in, (T *)out);
    return check_launch("transpose");
}
int launch_transpose(int dtype, int rows, int cols, const void *in, void *out, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return CGAMD_OK;
    CG_DISPATCH(dtype, transpose_impl, rows, cols, in, out, st);
}

// ---- two-launch iteration: SpMV fused with the previous iteration's beta / aypx ---------------------------------
bool fused2_ok(const SpmvPlan &plan, int dtype, int nrhs, const void *vals, const int *cols) {
    (void)dtype;
    if (tune().two_launch == 0 || !fold_alpha_ok(plan.n_partials, plan.fold_max)) return false;
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
    f.part_rr = static_cast<const A *>(part_rr); f.P = P; f.K = sc.krr;
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
    f.r = nullptr; f.dnew = nullptr; f.part_rr = static_cast<const A *>(part_rr); f.P = P; f.K = sc.krr;
    f.delta = (T *)sc.delta; f.beta = (T *)sc.beta; f.history = (T *)sc.history; f.history_cap = sc.history_cap; f.iter = sc.iter;
    hipLaunchKernelGGL((cg_tail_kernel<T, kBlock>), dim3(nrhs), dim3(kBlock), 0, st, f, nrhs);
    return check_launch("cg_tail");
}
int launch_cg_tail(int dtype, const void *part_rr, int P, int nrhs, const CgScalars &sc, hipStream_t st) {
    CG_DISPATCH(dtype, cg_tail_impl, part_rr, P, nrhs, sc, st);
}

}  // namespace cgamd
