// Peer-to-peer backend of the row-partitioned multi-GPU loop: kernels that push / wait for halo entries and exchange scalar sums
// through IPC-shared mailboxes over xGMI (protocol: p2p_device.h), and their launchers.
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"
#include "spmv_device.h"
#include "reduce_device.h"
#include "launch_util.h"
#include "p2p_device.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

namespace cgamd {

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

// CMODE: 0 = aCols / aValues, 1 = one-byte column codes, 2 = one-byte column AND value codes (index_codes.hip: the same columns and
// the same value bits either way)
template <typename T, int BLOCK, bool NT, int UNROLL, int CMODE = 0>
__global__ __launch_bounds__(BLOCK) void spmv_rowblock_p2p_kernel(SpmvP2pArgs<T> g) {
    using A = typename VT<T>::acc;
    constexpr bool CODED = CMODE >= 1, VCODED = CMODE == 2;
    const SpmvArgs<T> &a = g.s;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);
    int *sc = reinterpret_cast<int *>(dyn_smem + (VCODED ? 0 : (size_t)a.cap * sizeof(T)));      // CODED: cap bytes of column codes
    [[maybe_unused]] const unsigned char *svc = reinterpret_cast<const unsigned char *>(dyn_smem) + a.cap;      // VCODED: cap bytes of value codes
    __shared__ A red[BLOCK / kWave];
    __shared__ int sdict[CODED ? BLOCK : 1];
    __shared__ T sdictv[VCODED ? BLOCK : 1];
    const int t = threadIdx.x, b = blockIdx.x;
    if constexpr (CODED) sdict[t] = a.dict[t];
    if constexpr (VCODED) sdictv[t] = a.vdict[t];
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
    if constexpr (VCODED) stage_codes2<BLOCK, NT>(a.codes, a.vcodes, cfirst, p1, reinterpret_cast<unsigned char *>(dyn_smem), reinterpret_cast<unsigned char *>(dyn_smem) + a.cap);
    else stage_slice<T, BLOCK, NT, CODED ? -3 : -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc, a.codes);
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
                if constexpr (VCODED) av[j] = sdictv[svc[idx]];
                else av[j] = sv[idx];
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
                if constexpr (VCODED) av[j] = sdictv[svc[idx]];
                else av[j] = sv[idx];
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
    const bool coded = plan.codes && !plan.codes16 && plan.codes_for == cols && tune().index_codes != 0;
    const bool vcoded = coded && sizeof(T) <= 8 && plan.vcodes && plan.vcodes_for == vals && tune().value_codes != 0;
    a.codes = coded ? plan.codes : nullptr;
    a.dict = coded ? plan.dict : nullptr;
    a.vcodes = vcoded ? plan.vcodes : nullptr;
    a.vdict = vcoded ? static_cast<const T *>(plan.vdict) : nullptr;
    const size_t lds = vcoded ? (((size_t)a.cap * 2 + 15) & ~(size_t)15)
                              : coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
    const int grid = rowblock_grid(plan.row_blocks, a.cycle);
    if (grid < e.n_peers * g.push_chunks) return fail(CGAMD_ERR_STATE, "spmv_p2p: fewer work-groups than push chunks");
    const dim3 gd(grid), block(kBlock);
    const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
    constexpr int U = sizeof(T) > 8 ? 4 : 8;
    // batch length of the row walk follows the longest row, as in spmv_impl (5 / 7: one batch per row of a 5- / 7-point stencil)
    const int fit = (sizeof(T) > 8 || tune().spmv_unroll) ? U : plan.max_row == 5 ? 5 : (plan.max_row == 6 || plan.max_row == 7) ? 7 : U;
#define CG_P2P(UU)                                                                                                      \
    do {                                                                                                                \
        if (vcoded) {                                                                                                   \
            if (nt) hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, true, UU, 2>), gd, block, lds, st, g);      \
            else hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, false, UU, 2>), gd, block, lds, st, g);        \
        } else if (coded) {                                                                                             \
            if (nt) hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, true, UU, 1>), gd, block, lds, st, g);      \
            else hipLaunchKernelGGL((spmv_rowblock_p2p_kernel<T, kBlock, false, UU, 1>), gd, block, lds, st, g);        \
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
