// Single-reduction CG (Chronopoulos & Gear 1989) for the row-partitioned multi-GPU loop: ONE global scalar exchange per iteration
// instead of the two of the reference's recurrence (clcg.c:297-419: d.q after the SpMV, r.r after the r update), and two launches.
//
//   reference recurrence (clcg.c:298-416)                 single-reduction form (same iterates in exact arithmetic)
//   q = A d ; alpha = delta / d.q        <- reduction 1    w = A r ; gamma = r.r, dl = w.r            <- the ONE reduction
//   x += alpha d ; r -= alpha q                            beta = gamma / gamma_old ; alpha = gamma / (dl - beta gamma / alpha_old)
//   delta_new = r.r                      <- reduction 2    p = r + beta p ; s = w + beta s  (= A p) ; x += alpha p ; r -= alpha s
//   beta = delta_new / delta ; d = beta d + r
//
// Per iteration a rank runs
//   spmv_cg1_kernel     w = A r for its rows (halo of r pushed / awaited inside the launch with the peer-to-peer backend, read in
//                       place from the mailbox), partials of r.w and r.r per 256-row block
//   cg1_update_kernel   the global sums (RCCL: already all-reduced; peer-to-peer: work-group 0 publishes this rank's sums to every
//                       mailbox, every work-group adds the ranks' slots in rank order), alpha / beta in every work-group's prologue,
//                       then the four vector updates in one pass (9 vector passes)
// Opt-in (CGAMD_DIST_SINGLE_REDUCTION): the rounding differs from the reference's recurrence, so results are held to a stated
// tolerance against its golden iterates (tests/test_gpu_dist_cg1.py), not bit for bit.  history[k] = r_k.r_k as everywhere.
// The unconjugated dot (reference kernel/complex/vdot.cl:15) makes the same algebra hold for complex-symmetric systems (COCG).
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"
#include "spmv_device.h"
#include "reduce_device.h"
#include "launch_util.h"
#include "p2p_device.h"

#include <algorithm>

namespace cgamd {

template <typename T> struct Cg1SpmvArgs {
    SpmvArgs<T> s;          // x = r (extended with the halo), y = w, partials = [2][row_blocks]: [0] r.w, [1] r.r
    P2pExchangeArgs x;      // peer-to-peer form only
    const T *halo;          // my mailbox's halo area: entry h is column n_local + h
    const int *halo_flag;   // per row block: references a halo column (may be null: none does)
    int n_local, rotate, push_chunks;
    int *iter;                          // iterations started (bumped here, read by the update launch)
    unsigned long long *slot_epoch;     // epoch of the scalar slots (bumped here, read by the update launch); may be null
};

// spmv_rowblock_p2p_kernel's structure (p2p.hip) with two fused dots; P2P = false: a rank without the peer-to-peer backend
// (halo already in the extended vector) or a single rank
// CMODE: 0 = aCols / aValues, 1 = one-byte column codes, 2 = one-byte column and value codes (index_codes.hip)
template <typename T, int BLOCK, bool NT, int UNROLL, int CMODE, bool P2P>
__global__ __launch_bounds__(BLOCK) void spmv_cg1_kernel(Cg1SpmvArgs<T> g) {
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
    if constexpr (P2P) {
        if (b < g.x.n_peers * g.push_chunks)
            p2p_push_chunk<T>(g.x, a.x, b / g.push_chunks, b % g.push_chunks, g.push_chunks, *g.x.epoch + 1);
    }
    if (b == 0 && t == 0) {     // nothing in THIS launch reads either word
        *g.iter = *g.iter + 1;
        if (g.slot_epoch) *g.slot_epoch = *g.slot_epoch + 1;
    }
    int rb = rowblock_of(b, a.row_blocks, a.cycle);
    if (rb < 0) return;
    if constexpr (P2P) {
        rb += g.rotate;
        if (rb >= a.row_blocks) rb -= a.row_blocks;
    }
    const int bflag = (P2P && g.halo_flag) ? g.halo_flag[rb] : 0;
    const int r0 = rb * BLOCK, row = r0 + t;
    const int rclamp = min(row, a.n - 1);
    const int s_raw = a.ptr[rclamp], e_raw = a.ptr[rclamp + 1];
    const int p0 = a.ptr[r0], p1 = a.ptr[min(r0 + BLOCK, a.n)];
    const int cfirst = p0 & ~3;
    const T r_own = a.x[rclamp];
    if constexpr (VCODED) stage_codes2<BLOCK, NT>(a.codes, a.vcodes, cfirst, p1, reinterpret_cast<unsigned char *>(dyn_smem), reinterpret_cast<unsigned char *>(dyn_smem) + a.cap);
    else stage_slice<T, BLOCK, NT, CODED ? -3 : -2>(a.vals, a.cols, a.nnz, cfirst, p1, sv, sc, a.codes);
    const int s = s_raw - cfirst, e = (row < a.n) ? e_raw - cfirst : s_raw - cfirst;
    const bool boundary = bflag != 0;
    if constexpr (P2P) {
        if (boundary && t == 0) {
            const char *mb = g.x.mailbox[g.x.rank];
            const unsigned long long ep = *g.x.epoch + 1;
            for (int p = 0; p < g.x.n_peers; ++p) {
                if (g.x.recv_count[p] == 0) continue;
                if (!spin_until(reinterpret_cast<const unsigned long long *>(mb + kMbHaloFlags) + g.x.peer_rank[p], ep, mb))
                    st_sys(reinterpret_cast<unsigned long long *>(const_cast<char *>(mb) + kMbError), 1ULL);
            }
        }
    }
    __syncthreads();
    T sum = vzero<T>();
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
        if (P2P && boundary) {      // block-uniform
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const bool far = cj[j] >= g.n_local;
                xv[j] = a.x[far ? rclamp : cj[j]];
                if (far) xv[j] = ld_sys_val(g.halo + (cj[j] - g.n_local));     // in place from the mailbox, cache-bypassing
            }
        } else {
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) xv[j] = a.x[cj[j]];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const T nxt = vfma(av[j], xv[j], sum);
            sum = vsel(k + j < e, nxt, sum);
        }
    }
    A d1 = vzero<A>(), d2 = vzero<A>();
    if (row < a.n) {
        a.y[row] = sum;
        d1 = to_acc(vmul(r_own, sum));
        d2 = to_acc(vmul(r_own, r_own));
    }
    const A t1 = block_sum<BLOCK>(d1, red);
    if (t == 0) a.partials[rb] = t1;
    const A t2 = block_sum<BLOCK>(d2, red);
    if (t == 0) a.partials[a.row_blocks + rb] = t2;
}

// r.r partials with the SpMV launch's own structure (one per 256-row block, same expression, same tree): the residual norm of the
// LAST iteration of an iterate() call has the bits the next call's SpMV launch will produce for it
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void cg1_rowblock_rr_kernel(int n, const T *__restrict__ r, typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    const int row = blockIdx.x * BLOCK + threadIdx.x;
    A d2 = vzero<A>();
    if (row < n) {
        const T rv = r[row];
        d2 = to_acc(vmul(rv, rv));
    }
    const A t2 = block_sum<BLOCK>(d2, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t2;
}

template <typename T> struct Cg1UpdateArgs {
    int n;
    T *r, *p, *s, *x;               // r: own part of the extended residual
    const T *w;
    const typename VT<T>::acc *red;         // MODE 0: {w.r, r.r} already summed over all ranks
    const typename VT<T>::acc *partials;    // MODE 1: [2][P] local partials of the SpMV launch
    int P;
    char *const *mailbox;
    int rank, nranks;
    const unsigned long long *slot_epoch;
    unsigned long long *halo_epoch;         // advanced here (nobody in this launch reads it); may be null
    T *state;                               // [2][2]: {gamma, alpha} of iteration k in state[k & 1]
    T *alpha, *beta, *delta, *history;
    int history_cap;
    const int *iter;
};

// this rank's two sums in a fixed order (thread-strided, wave tree, 4 wave sums); result valid in thread 0.  Shared by the update
// launch's work-group 0 and the tail kernel, so both produce the same bits for the same partials
template <typename A, int BLOCK> CG_DEV void cg1_local_sums(const A *partials, int P, bool both, A &s_wr, A &s_rr, A *red) {
    A a1 = vzero<A>(), a2 = vzero<A>();
    int i = threadIdx.x;
    for (; i + 7 * BLOCK < P; i += 8 * BLOCK) {      // 8 (x2) loads in flight; the additions keep the thread-strided order
        A v1[8], v2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v2[k] = partials[P + i + k * BLOCK];
            if (both) v1[k] = partials[i + k * BLOCK];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (both) a1 = vadd(a1, v1[k]);
            a2 = vadd(a2, v2[k]);
        }
    }
    for (; i < P; i += BLOCK) {
        if (both) a1 = vadd(a1, partials[i]);
        a2 = vadd(a2, partials[P + i]);
    }
    s_wr = block_sum<BLOCK>(a1, red);
    s_rr = block_sum<BLOCK>(a2, red);
}

// publish (thread s < nranks writes rank s's mailbox) and gather the slots of all ranks from my own mailbox; sums in rank order,
// valid in thread 0.  vals: LDS double[64][4]; own: LDS double[4] holding my sums
CG_DEV bool cg1_exchange(char *const *mailbox, int rank, int nranks, unsigned long long ep, bool publish, const double *own,
                         double (*vals)[4], double2 &g_wr, double2 &g_rr) {
    const int s = threadIdx.x;
    const long long slot_off = kMbCg1 + ((long long)(ep & 1) * 64) * 64;
    if (publish && s < nranks) {
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(mailbox[s] + slot_off + (long long)rank * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) st_sys(slot + i, (unsigned long long)__double_as_longlong(own[i]));
        p2p_stores_done();
        st_sys(slot + 4, ep);
    }
    bool ok = true;
    if (s < nranks) {
        const unsigned long long *in = reinterpret_cast<const unsigned long long *>(mailbox[rank] + slot_off + (long long)s * 64);
        if (!spin_until(in + 4, ep, mailbox[rank])) {
            st_sys(reinterpret_cast<unsigned long long *>(mailbox[rank] + kMbError), 2ULL);
            ok = false;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) vals[s][i] = __longlong_as_double((long long)ld_sys(in + i));
    }
    __syncthreads();
    if (s == 0) {
        g_wr = make_double2(0., 0.);
        g_rr = make_double2(0., 0.);
        for (int k = 0; k < nranks; ++k) {
            g_wr.x += vals[k][0]; g_wr.y += vals[k][1];
            g_rr.x += vals[k][2]; g_rr.y += vals[k][3];
        }
    }
    return ok;
}

// MODE 0: sums in `red` (RCCL all-reduce, or a single rank); MODE 1: peer-to-peer exchange in the prologue.
template <typename T, int BLOCK, bool VEC, int VNT, int MODE>
__global__ __launch_bounds__(BLOCK) void cg1_update_kernel(Cg1UpdateArgs<T> u) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ double vals[64][4];
    __shared__ double own[4];
    __shared__ T ab[2];
    const int t = threadIdx.x;
    const int k = *u.iter;              // the iteration this launch completes (1-based): advanced by its SpMV launch
    double2 g_wr = make_double2(0., 0.), g_rr = make_double2(0., 0.);
    if constexpr (MODE == 0) {
        if (t == 0) { g_wr = to_acc2(u.red[0]); g_rr = to_acc2(u.red[1]); }
    } else {
        const unsigned long long ep = *u.slot_epoch;
        const bool pub = blockIdx.x == 0;
        if (pub) {
            A s_wr, s_rr;
            cg1_local_sums<A, BLOCK>(u.partials, u.P, true, s_wr, s_rr, red);
            if (t == 0) {
                const double2 a = to_acc2(s_wr), b = to_acc2(s_rr);
                own[0] = a.x; own[1] = a.y; own[2] = b.x; own[3] = b.y;
            }
            __syncthreads();
        }
        cg1_exchange(u.mailbox, u.rank, u.nranks, ep, pub, own, vals, g_wr, g_rr);
    }
    if (t == 0) {
        const T gT = from_acc<T>(from_acc2<A>(g_rr)), dT = from_acc<T>(from_acc2<A>(g_wr));
        T beta = vzero<T>(), alpha;
        if (k <= 1) {
            alpha = from_acc<T>(acc_div(to_acc(gT), to_acc(dT)));
        } else {
            const T gold = u.state[((k - 1) & 1) * 2], aold = u.state[((k - 1) & 1) * 2 + 1];
            beta = from_acc<T>(acc_div(to_acc(gT), to_acc(gold)));
            const A corr = acc_div(vmul(to_acc(beta), to_acc(gT)), to_acc(aold));
            alpha = from_acc<T>(acc_div(to_acc(gT), vsub(to_acc(dT), corr)));
        }
        ab[0] = alpha; ab[1] = beta;
        if (blockIdx.x == 0) {
            u.state[(k & 1) * 2] = gT;
            u.state[(k & 1) * 2 + 1] = alpha;
            u.alpha[0] = alpha; u.beta[0] = beta; u.delta[0] = gT;
            if (k >= 1 && k - 1 < u.history_cap) u.history[k - 1] = gT;       // gamma is r.r of the residual this iteration started from
            if (u.halo_epoch) *u.halo_epoch = *u.halo_epoch + 1;              // the exchange the SpMV launch just used
        }
    }
    __syncthreads();
    const T al = ab[0], bt = ab[1];
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + t;
    if (VEC) {
        const long long npack = u.n / E;
        for (long long i = i0; i < npack; i += stride) {
            Pack<T> pr = ld_pack(u.r + i * E), pp = ld_pack(u.p + i * E), ps = ld_pack(u.s + i * E);
            const Pack<T> pw = (VNT & 2) ? ld_pack_nt(u.w + i * E) : ld_pack(u.w + i * E);
            Pack<T> px = (VNT & 1) ? ld_pack_nt(u.x + i * E) : ld_pack(u.x + i * E);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                pp.v[e] = vaypx(bt, pp.v[e], pr.v[e]);
                ps.v[e] = vaypx(bt, ps.v[e], pw.v[e]);
                px.v[e] = vadd(px.v[e], vmul(al, pp.v[e]));
                pr.v[e] = vsub(pr.v[e], vmul(al, ps.v[e]));
            }
            st_pack(u.p + i * E, pp);
            st_pack(u.s + i * E, ps);
            if (VNT & 1) st_pack_nt(u.x + i * E, px); else st_pack(u.x + i * E, px);
            st_pack(u.r + i * E, pr);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < u.n; i += stride) {
        const T pn = vaypx(bt, u.p[i], u.r[i]), sn = vaypx(bt, u.s[i], u.w[i]);
        u.p[i] = pn;
        u.s[i] = sn;
        u.x[i] = vadd(u.x[i], vmul(al, pn));
        u.r[i] = vsub(u.r[i], vmul(al, sn));
    }
}

// history[*iter] = r.r of the current residual (end of an iterate() call; no state of the recurrence is touched).
// MODE 0: `red[1]` holds the global sum; MODE 1: one work-group sums `partials` ([P] r.r partials at partials + P) and exchanges.
template <typename T, int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void cg1_tail_kernel(Cg1UpdateArgs<T> u, unsigned long long *slot_epoch_rw) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ double vals[64][4];
    __shared__ double own[4];
    double2 g_wr = make_double2(0., 0.), g_rr = make_double2(0., 0.);
    if constexpr (MODE == 0) {
        if (threadIdx.x == 0) g_rr = to_acc2(u.red[1]);
    } else {
        const unsigned long long ep = *slot_epoch_rw + 1;
        A s_wr, s_rr;
        cg1_local_sums<A, BLOCK>(u.partials, u.P, false, s_wr, s_rr, red);
        if (threadIdx.x == 0) {
            const double2 b = to_acc2(s_rr);
            own[0] = 0.; own[1] = 0.; own[2] = b.x; own[3] = b.y;
        }
        __syncthreads();
        cg1_exchange(u.mailbox, u.rank, u.nranks, ep, true, own, vals, g_wr, g_rr);
        if (threadIdx.x == 0) *slot_epoch_rw = ep;
    }
    if (threadIdx.x == 0) {
        const int k = *u.iter;
        const T gT = from_acc<T>(from_acc2<A>(g_rr));
        u.delta[0] = gT;
        if (k < u.history_cap) u.history[k] = gT;
    }
}

// =================================================================================================
// launchers
// =================================================================================================
template <typename T>
static int spmv_cg1_impl(const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols, const void *r_ext,
                         void *w, void *partials, const int *halo_flag, int rotate, const P2pExchange *e, int *iter,
                         unsigned long long *slot_epoch, hipStream_t st) {
    Cg1SpmvArgs<T> g;
    SpmvArgs<T> &a = g.s;
    a.n = n; a.nrhs = 1; a.nnz = nnz;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<const T *>(r_ext); a.ldx = 0;
    a.y = static_cast<T *>(w); a.ldy = 0;
    a.dvec = nullptr;
    a.partials = static_cast<typename VT<T>::acc *>(partials);
    a.row_blocks = plan.row_blocks; a.rb_list = nullptr; a.rb_count = 0;
    a.cap = (plan.max_span + 3) & ~3;
    a.cycle = tune().spmv_cycle > 0 ? tune().spmv_cycle : 1;
    const bool p2p = e != nullptr;
    g.halo = nullptr; g.halo_flag = halo_flag; g.n_local = n; g.rotate = 0; g.push_chunks = 1;
    g.iter = iter; g.slot_epoch = slot_epoch;
    g.x = P2pExchangeArgs{};
    if (p2p) {
        g.x.mailbox = e->mailbox; g.x.rank = e->rank; g.x.n_peers = e->n_peers; g.x.n_local = e->n_local;
        g.x.peer_rank = e->peer_rank; g.x.send_off = e->send_off; g.x.send_count = e->send_count; g.x.dst_off = e->dst_off;
        g.x.recv_off = e->recv_off; g.x.recv_count = e->recv_count; g.x.send_index = e->send_index; g.x.epoch = e->epoch;
        g.x.counters = e->counters; g.x.max_count = e->max_count;
        g.halo = static_cast<const T *>(e->my_halo);
        g.n_local = e->n_local; g.rotate = rotate; g.push_chunks = p2p_push_chunks(*e);
    }
    const bool coded = plan.codes && !plan.codes16 && plan.codes_for == cols && tune().index_codes != 0;
    const bool vcoded = coded && sizeof(T) <= 8 && plan.vcodes && plan.vcodes_for == vals && tune().value_codes != 0;
    a.codes = coded ? plan.codes : nullptr;
    a.dict = coded ? plan.dict : nullptr;
    a.vcodes = vcoded ? plan.vcodes : nullptr;
    a.vdict = vcoded ? static_cast<const T *>(plan.vdict) : nullptr;
    const size_t lds = vcoded ? (((size_t)a.cap * 2 + 15) & ~(size_t)15)
                              : coded ? (((size_t)a.cap * (sizeof(T) + 1) + 15) & ~(size_t)15) : (size_t)a.cap * (sizeof(T) + 4);
    const int grid = rowblock_grid(plan.row_blocks, a.cycle);
    if (p2p && grid < e->n_peers * g.push_chunks) return fail(CGAMD_ERR_STATE, "spmv_cg1: fewer work-groups than push chunks");
    const dim3 gd(grid), block(kBlock);
    const bool nt = tune().spmv_nt >= 0 ? (tune().spmv_nt != 0) : (plan.nt != 0);
    constexpr int U = sizeof(T) > 8 ? 4 : 8;
    const int fit = (sizeof(T) > 8 || tune().spmv_unroll) ? U : plan.max_row == 5 ? 5 : (plan.max_row == 6 || plan.max_row == 7) ? 7 : U;
#define CG1_L(UU, CO, PP)                                                                                              \
    do {                                                                                                                \
        if (nt) hipLaunchKernelGGL((spmv_cg1_kernel<T, kBlock, true, UU, CO, PP>), gd, block, lds, st, g);             \
        else hipLaunchKernelGGL((spmv_cg1_kernel<T, kBlock, false, UU, CO, PP>), gd, block, lds, st, g);               \
    } while (0)
#define CG1_U(UU)                                                                                                      \
    do {                                                                                                                \
        if (vcoded) { if (p2p) CG1_L(UU, 2, true); else CG1_L(UU, 2, false); }                                         \
        else if (coded) { if (p2p) CG1_L(UU, 1, true); else CG1_L(UU, 1, false); }                                     \
        else { if (p2p) CG1_L(UU, 0, true); else CG1_L(UU, 0, false); }                                                \
    } while (0)
    if (fit == 5) CG1_U(5);
    else if (fit == 7) CG1_U(7);
    else CG1_U(U);
#undef CG1_U
#undef CG1_L
    return check_launch("spmv_cg1");
}
int launch_spmv_cg1(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                    const void *r_ext, void *w, void *partials, const int *halo_flag, int rotate, const P2pExchange *e, int *iter,
                    unsigned long long *slot_epoch, hipStream_t st) {
    if (plan.kind != 5 || !aligned16(vals) || !aligned16(cols)) return fail(CGAMD_ERR_STATE, "spmv_cg1: needs the row-block kernel");
    CG_DISPATCH(dtype, spmv_cg1_impl, plan, n, nnz, vals, ptr, cols, r_ext, w, partials, halo_flag, rotate, e, iter, slot_epoch, st);
}
bool cg1_supported(const SpmvPlan &plan, const void *vals, const int *cols) { return plan.kind == 5 && aligned16(vals) && aligned16(cols); }

template <typename T> static int cg1_rr_impl(int n, const void *r, void *partials, hipStream_t st) {
    hipLaunchKernelGGL((cg1_rowblock_rr_kernel<T, kBlock>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const T *)r,
                       (typename VT<T>::acc *)partials);
    return check_launch("cg1_rowblock_rr");
}
int launch_cg1_rowblock_rr(int dtype, int n, const void *r, void *partials, hipStream_t st) {
    CG_DISPATCH(dtype, cg1_rr_impl, n, r, partials, st);
}

template <typename T> static Cg1UpdateArgs<T> cg1_args(const Cg1Update &c) {
    Cg1UpdateArgs<T> u;
    u.n = c.n; u.r = (T *)c.r; u.p = (T *)c.p; u.s = (T *)c.s; u.x = (T *)c.x; u.w = (const T *)c.w;
    u.red = (const typename VT<T>::acc *)c.red; u.partials = (const typename VT<T>::acc *)c.partials; u.P = c.P;
    u.mailbox = c.mailbox; u.rank = c.rank; u.nranks = c.nranks; u.slot_epoch = c.slot_epoch; u.halo_epoch = c.halo_epoch;
    u.state = (T *)c.state; u.alpha = (T *)c.sc.alpha; u.beta = (T *)c.sc.beta; u.delta = (T *)c.sc.delta;
    u.history = (T *)c.sc.history; u.history_cap = c.sc.history_cap; u.iter = c.sc.iter;
    return u;
}
// the peer-to-peer form spins in every work-group until all ranks' slots have arrived, and this rank's own slot is published by
// work-group 0 of the same launch: the grid is capped at what is resident at once (p2p.hip: aypx_beta_p2p_kernel)
template <typename K> static int cg1_resident_cap(K kernel) {
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, kBlock, 0) != hipSuccess) return 0;
    if (per > 8) per = 8;
    if (per > 1) per -= 1;
    return cus * per;
}
template <typename T> static int cg1_update_impl(const Cg1Update &c, bool vec, int vnt, hipStream_t st) {
    const Cg1UpdateArgs<T> u = cg1_args<T>(c);
    const int want = vec_grid(c.n, VT<T>::dtype);
#define CG1_UP(V, N, M)                                                                                                 \
    do {                                                                                                                 \
        int g = want;                                                                                                    \
        if (M == 1) {                                                                                                    \
            static const int cap = cg1_resident_cap(cg1_update_kernel<T, kBlock, V, N, M>);                             \
            if (cap < 1) return fail(CGAMD_ERR_HIP, "cg1_update: occupancy query failed; refusing an all-work-group spin"); \
            g = std::min(g, cap);                                                                                        \
        }                                                                                                                \
        hipLaunchKernelGGL((cg1_update_kernel<T, kBlock, V, N, M>), dim3(g), dim3(kBlock), 0, st, u);                    \
    } while (0)
    if (c.mailbox) {
        if (vec && vnt == 3) CG1_UP(true, 3, 1); else if (vec) CG1_UP(true, 0, 1); else CG1_UP(false, 0, 1);
    } else {
        if (vec && vnt == 3) CG1_UP(true, 3, 0); else if (vec) CG1_UP(true, 0, 0); else CG1_UP(false, 0, 0);
    }
#undef CG1_UP
    return check_launch("cg1_update");
}
int launch_cg1_update(int dtype, const Cg1Update &c, hipStream_t st, int vec_nt) {
    if (c.n <= 0) return CGAMD_OK;
    if (c.nranks > 64) return fail(CGAMD_ERR_INVALID, "single-reduction loop: at most 64 ranks");
    const bool v = vec_ok(dtype, c.n, 1, {c.r, c.p, c.s, c.x, c.w});
    const int vnt = (tune().vec_nt >= 0 ? tune().vec_nt : vec_nt) == 3 ? 3 : 0;
    CG_DISPATCH(dtype, cg1_update_impl, c, v, vnt, st);
}
template <typename T> static int cg1_tail_impl(const Cg1Update &c, unsigned long long *slot_epoch_rw, hipStream_t st) {
    const Cg1UpdateArgs<T> u = cg1_args<T>(c);
    if (c.mailbox) hipLaunchKernelGGL((cg1_tail_kernel<T, kBlock, 1>), dim3(1), dim3(kBlock), 0, st, u, slot_epoch_rw);
    else hipLaunchKernelGGL((cg1_tail_kernel<T, kBlock, 0>), dim3(1), dim3(kBlock), 0, st, u, slot_epoch_rw);
    return check_launch("cg1_tail");
}
int launch_cg1_tail(int dtype, const Cg1Update &c, unsigned long long *slot_epoch_rw, hipStream_t st) {
    CG_DISPATCH(dtype, cg1_tail_impl, c, slot_epoch_rw, st);
}

}  // namespace cgamd
