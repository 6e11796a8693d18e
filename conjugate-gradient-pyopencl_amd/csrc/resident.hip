// Resident CG loop for small systems (gfx950): ALL iterations of an iterate() call inside ONE launch.
//
// The reference solves many small systems per time step (one sub-domain per right-hand side: 16k rows x 9 complex64 RHS x
// 256 iterations, p_h-PY_C-CL.py:1925-1950 -> clcg.c:297-419); at that size an iteration is a handful of microseconds of
// work behind two kernel launches.  Here a GROUP of G = ceil(n / 1024) work-groups (one per CU) owns one
// right-hand side for the whole solve:
//   * member m (512 threads, two rows each) keeps rows [1024 m, 1024 m + 1024) of the matrix in LDS for the whole launch
//     and its 16-byte packs of x, r and d in registers;
//   * the recurrence is the two-launch loop's (spmv.hip spmv_fused_kernel / vector.hip axpy2_dot_alpha_kernel): beta and
//     d = beta d + r are recomputed for every gathered column, d ping-pongs between two buffers; only r and d are
//     exchanged through memory (write + gather), q stays in LDS;
//   * the two scalar reductions of an iteration are "granule" all-gathers: every partial sum is published as 8-byte
//     words {tag, 32 payload bits}; a member polls the P words of its group and has barrier and data in one round trip.
//     Tags count (claim, iteration, phase) within the launch; the words are zeroed before every launch.
// Bit-identity: per-row accumulation order, the 256-row / 256-pack partial sums (wave tree, then 4 wave sums) and the order
// the partials are added are those of the two-launch kernels, and the library is built with -ffp-contract=off, so x,
// the residual history and the device scalars equal the other loops' bit for bit (tests/test_gpu_resident.py).
//
// Placement: groups are formed INSIDE the launch from the work-groups that are actually running -- a work-group takes a
// ticket from the counter of its XCD (s_getreg XCC_ID); tickets [g G, g G + G) are group g of that XCD; the member that
// draws the last ticket knows the group is complete and claims right-hand sides for it from a global counter until none is
// left.  Nothing depends on the dispatch order or on a work-group -> XCD mapping.  With all members of a group on ONE XCD
// (`LOCAL`) that XCD's L2 is the point of coherence: plain stores, L1-bypassing (sc1) loads.  Groups wider than an XCD
// (`!LOCAL`, G > 32) use write-through (sc1) stores and sc1 loads, the cross-XCD form of MI355X_MICROARCH.md "Valid forms".
// Every spin is bounded (wall clock); a time-out sets the error word, all work-groups drain and the host reports it.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "cgamd_internal.h"
#include "device_mem.h"
#include "resident_device.h"

namespace cgamd {
namespace {


template <typename T> struct ResArgs {
    int n, nrhs, G, row_blocks, P_rr, npack;
    int it0, K, history_cap, cap;   // cap: LDS entries reserved for the slice (multiple of 4)
    int wcap;                       // LDS entries for the window of d / r a member stages per iteration (0 = gather from L2)
    const T *vals;
    const int *ptr, *cols;
    T *x, *r, *d0, *d1;             // RHS-major [nrhs][n]; d of iteration k lives in (k & 1 ? d1 : d0)
    typename VT<T>::acc *part_rr;   // [nrhs][P_rr]: read when it0 > 0, written by the last iteration
    T *alpha, *beta, *delta, *history;
    int *iter;
    unsigned *hdr;                  // kHdrWords
    u64 *slot_word;                 // [slots] {claim sequence, rhs}; slot = xcc * lg + group of that XCD
    u64 *gran;                      // [slots][2][gran_stride] granule words
    int gran_stride, lg;            // lg: groups per ticket counter (per XCD when LOCAL)
    long long claim_ticks;          // bound (100 MHz ticks) of the wait for a group to fill while nothing has started
    double tol;                     // > 0 (one right-hand side): stop before the first iteration whose sqrt|r.r| < tol (or is NaN)
    long long *prof;                // diagnostics (CGAMD_RESIDENT_PROF=1): phase times of one work-group, s_memtime ticks
};
#define RES_STAMP(i)                                                                   \
    if (a.prof && t == 0 && m == 0 && rhs == 0) {                                        \
        const long long now__ = clock64();                                               \
        a.prof[i] += now__ - stamp;                                                      \
        stamp = now__;                                                                   \
    }


// column range [w0, w0 + wlen) a member stages per iteration: the columns of its rows and the rows themselves, w0 a multiple of E
CG_DEV void window_of(int cmin, int cmax, int R0, int n, int E, int &w0, int &wlen) {
    w0 = min(cmin, R0) & ~(E - 1);
    wlen = max(cmax, min(R0 + kResRows, n) - 1) - w0 + 1;
}

// largest window over the members (host: does it fit the LDS left over?): out[0] = max wlen
__global__ __launch_bounds__(kResThreads) void resident_window_kernel(int n, int E, const int *__restrict__ ptr, const int *__restrict__ cols, int *out) {
    __shared__ int smin, smax;
    const int t = threadIdx.x, R0 = blockIdx.x * kResRows;
    if (t == 0) { smin = 0x7fffffff; smax = 0; }
    __syncthreads();
    const int p0 = ptr[R0], p1 = ptr[min(R0 + kResRows, n)];
    int cmin = 0x7fffffff, cmax = 0;
    for (int i = p0 + t; i < p1; i += kResThreads) { cmin = min(cmin, cols[i]); cmax = max(cmax, cols[i]); }
    atomicMin(&smin, cmin);
    atomicMax(&smax, cmax);
    __syncthreads();
    if (t == 0) {
        int w0, wlen;
        window_of(smin, smax, R0, n, E, w0, wlen);
        atomicMax(out, wlen);
    }
}

template <typename T, bool LOCAL, int UNROLL, bool WINDOW>
__global__ __launch_bounds__(kResThreads) void cg_resident_kernel(ResArgs<T> a) {
    using A = typename VT<T>::acc;
    constexpr int E = Pack<T>::N;                    // 4, 2, or 1 (complex128: one value per 16-byte pack)
    constexpr int W = sizeof(A) / 4;                 // granule words per partial
    constexpr int PPT = E >= 2 ? 1 : 2;              // 16-byte packs of every vector per thread: 1024 rows / E packs on 512 threads
    constexpr int VT_ = kResRows / E / PPT;          // threads with packs (<= kResThreads); pack j of thread t: local pack t + j VT_
    static_assert(VT_ <= kResThreads && VT_ * PPT * E == kResRows, "the packs cover the member's rows");
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *sv = reinterpret_cast<T *>(dyn_smem);
    int *sc = reinterpret_cast<int *>(dyn_smem + (size_t)a.cap * sizeof(T));
    T *qs = reinterpret_cast<T *>(dyn_smem + (size_t)a.cap * (sizeof(T) + sizeof(int)));
    T *win = qs + kResRows;
    __shared__ ResShared sh;
    const int t = threadIdx.x;

    // ---- group formation
    if (t == 0) {
        const unsigned xcc = LOCAL ? xcc_id() : 0u;
        sh.ctl[0] = (int)xcc;
        sh.ctl[1] = (int)atomicAdd(a.hdr + kHdrTicket + xcc, 1u);
        sh.fail = 0;
        sh.cmin = 0x7fffffff;
        sh.cmax = 0;
    }
    __syncthreads();
    const int xcc = __builtin_amdgcn_readfirstlane(sh.ctl[0]), ticket = __builtin_amdgcn_readfirstlane(sh.ctl[1]);
    const int lg = ticket / a.G, m = ticket - lg * a.G;
    if (lg >= a.lg) return;                          // beyond the groups this launch has words for (several work-groups per CU)
    const int slot = xcc * a.lg + lg;
    const bool leader = m == a.G - 1;                // drew the group's last ticket: every member is running

    // ---- my 1024 rows of the matrix -> LDS, once; thread t walks rows t and t + 512 of them
    const int R0 = m * kResRows;
    const int p0 = a.ptr[R0], p1 = a.ptr[min(R0 + kResRows, a.n)];
    const int cfirst = p0 & ~3;
    int cmin = 0x7fffffff, cmax = 0;
    for (int i = cfirst + t; i < p1; i += kResThreads) {
        const int c = a.cols[i];
        sv[i - cfirst] = a.vals[i];
        sc[i - cfirst] = c;
        if (i >= p0) { cmin = min(cmin, c); cmax = max(cmax, c); }
    }
    atomicMin(&sh.cmin, cmin);
    atomicMax(&sh.cmax, cmax);
    __syncthreads();
    // WINDOW (the host checked that every member's range fits the LDS left over): the columns my rows reference, and the
    // rows themselves, lie in [w0, w0 + wlen); each iteration has beta d + r of that range in LDS -- my own rows straight from
    // the registers that hold them, the halo by coalesced loads, one per entry -- and the row walk reads LDS.  !WINDOW:
    // every non-zero gathers d and r from L2 (each entry ~ row-length times per work-group: 1 us more at 9 per row).
    int w0, wlen;
    window_of(sh.cmin, sh.cmax, R0, a.n, E, w0, wlen);
    constexpr bool windowed = WINDOW;
    const int wpacks = (wlen + E - 1) / E;
    const int nlow = (R0 - w0) / E;                                             // halo packs below my rows
    const int hi0 = (min(R0 + kResRows, a.n) - w0) / E;                         // first pack above them
    const int nhalo = nlow + (wpacks - hi0);
    __syncthreads();
    for (int i = cfirst + t; i < p1; i += kResThreads)         // byte offsets: into the window, or into the global vectors (n <= 65536)
        sc[i - cfirst] = (windowed ? sc[i - cfirst] - w0 : sc[i - cfirst]) * (int)sizeof(T);
    int rs[2], re[2];
    unsigned own_off[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = R0 + t + h * kResThreads, rc = min(row, a.n - 1);
        live[h] = row < a.n;
        rs[h] = a.ptr[rc] - cfirst;
        re[h] = live[h] ? a.ptr[rc + 1] - cfirst : rs[h];
        own_off[h] = (unsigned)(windowed ? rc - w0 : rc) * (unsigned)sizeof(T);
    }
    bool packer[PPT];                                // my 16-byte packs of every vector (t < VT_)
    unsigned pack_off[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int pack = m * (kResRows / E) + t + j * VT_;
        packer[j] = t < VT_ && pack < a.npack;
        pack_off[j] = (unsigned)pack * 16u;
    }
    u64 *g_dq = a.gran + (size_t)(slot * 2) * a.gran_stride, *g_rr = g_dq + a.gran_stride;
    __syncthreads();
    // WINDOW: the first UNROLL entries of my two rows stay in registers for the whole launch (longer rows: the rest from LDS)
    T mv[2][WINDOW ? UNROLL : 1];
    int mo[2][WINDOW ? UNROLL : 1];
    if constexpr (WINDOW) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const int idx = max(min(rs[h] + j, re[h] - 1), 0);
                const bool real = rs[h] + j < re[h];
                mv[h][j] = vsel(real, sv[idx], vzero<T>());                         // beyond the row's end: 0 times the window's zero cell
                mo[h][j] = real ? sc[idx] : (a.wcap - 1) * (int)sizeof(T);
            }
        if (t == 0) win[a.wcap - 1] = vzero<T>();                                   // never touched by the staging (pack of slack)
    }

    for (unsigned seq = 1;; ++seq) {
        // ---- which right-hand side: the leader claims, the others wait for its word (or for the end of all solves)
        if (t == 0) {
            int rhs = -1;
            if (leader) {
                rhs = (int)atomicAdd(a.hdr + kHdrNextRhs, 1u);
                __hip_atomic_store(a.slot_word + slot, ((u64)seq << 32) | (unsigned)rhs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const long long t0 = wall_clock64();
                for (unsigned spins = 0;; ++spins) {
                    const u64 w = ld_word(a.slot_word + slot);
                    if ((unsigned)(w >> 32) == seq) { rhs = (int)(unsigned)w; break; }
                    if (ld_word(a.hdr + kHdrSolved) >= (unsigned)a.nrhs) break;           // every solve is done: nothing left for anyone
                    // gives up only while NO group has claimed anything: then no vector has been touched and the host can take the
                    // launched loops instead; once a group works, the others wait for it however long its solves take
                    if ((spins & 63) == 63 && ((wall_clock64() - t0 > a.claim_ticks && ld_word(a.hdr + kHdrNextRhs) == 0) || ld_word(a.hdr + kHdrError) != 0)) {
                        atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrClaim);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            sh.ctl[2] = rhs;
        }
        __syncthreads();
        const int rhs = __builtin_amdgcn_readfirstlane(sh.ctl[2]);
        __syncthreads();
        if (rhs < 0 || rhs >= a.nrhs) return;

        const long long off = (long long)rhs * a.n;
        T *xr = a.x + off, *rr = a.r + off, *d0r = a.d0 + off, *d1r = a.d1 + off;
        Pack<T> px[PPT], pr[PPT], pd[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (packer[j]) {
                px[j] = ld_pack(at_off(xr, pack_off[j]));
                pr[j] = ld_pack(at_off(rr, pack_off[j]));
                pd[j] = ld_pack(at_off((a.it0 & 1) ? d1r : d0r, pack_off[j]));
            }
        T dlt = a.delta[rhs];                        // delta of the iteration being run (it0 == 0: delta_0 of set_rhs)
        const unsigned tag0 = seq << 20;

        long long stamp = clock64();
        int stopped_at = -1;
        for (int k = 0; k < a.K; ++k) {
            const int it = a.it0 + k;
            RES_STAMP(0)
            // ---- beta from the previous iteration's r.r partials (spmv_fused_kernel's prologue); the poll is barrier 2
            T bt = vzero<T>();
            const T *dold_p = (it & 1) ? d1r : d0r;
            T *dnew_p = (it & 1) ? d0r : d1r;
            // WINDOW: the first round of halo loads (d_old and r of my neighbours) needs barrier 2 but not beta: issued between the
            // poll and wave 0's divisions (4.57 -> 4.33 us per iteration at 16k rows x 9 complex64, scripts/resident_ab.py)
            Pack<T> h_da, h_ra, h_db, h_rb;
            int h_pa = 0, h_pb = 0;
            bool h_two = false;
            bool halo_prefetch_late = false;
            const int ht = kResThreads - 1 - t;       // the LAST waves load the halo: wave 0 is busy with the divisions meanwhile
            auto halo_prefetch = [&]() {
                if constexpr (WINDOW) {
                    if (ht < nhalo) {
                        const int hq = ht + kResThreads;
                        h_two = hq < nhalo;
                        h_pa = ht < nlow ? ht : hi0 + (ht - nlow);
                        h_pb = h_two ? (hq < nlow ? hq : hi0 + (hq - nlow)) : h_pa;
                        h_da = ld_pack_coh(dold_p + w0 + h_pa * E); h_ra = ld_pack_coh(rr + w0 + h_pa * E);
                        h_db = ld_pack_coh(dold_p + w0 + h_pb * E); h_rb = ld_pack_coh(rr + w0 + h_pb * E);
                    }
                }
            };
            if (it == 0) halo_prefetch_late = true;      // no prologue in the very first iteration
            if (it > 0) {
                const T dold = k == 0 ? a.history[(long long)(it - 1) * a.nrhs + rhs] : dlt;
                auto fin = [&](A tot, T &b, T &dn) {
                    dn = from_acc<T>(tot);
                    b = from_acc<T>(acc_div(to_acc(dn), to_acc(dold)));
                };
                T dnT;
                bool ok;
                if (k == 0) {
                    const A *p = a.part_rr + (long long)rhs * a.P_rr;
                    ok = group_scalars<A, T>(a.P_rr, sh, a.hdr, bt, dnT, [&](int i, A &v) { v = p[i]; return true; }, fin, WINDOW, halo_prefetch);
                } else {
                    ok = group_scalars<A, T>(a.P_rr, sh, a.hdr, bt, dnT, [&](int i, A &v) { return get_granule(g_rr + (size_t)i * W, tag0 + 2 * k, v); }, fin,
                                             WINDOW, halo_prefetch);
                }
                if (!ok) return;
                dlt = dnT;
                if (leader && t == 0) {
                    a.beta[rhs] = bt;
                    a.delta[rhs] = dnT;
                    if (it < a.history_cap) a.history[(long long)it * a.nrhs + rhs] = dnT;
                }
            }
            if (a.tol > 0. && it > 0 && !(res_norm(dlt) >= a.tol)) {     // (the reference tests after an iteration: never before the first)      // uniform: every member holds the same delta
                stopped_at = it;
                break;
            }
            if (halo_prefetch_late) halo_prefetch();
            RES_STAMP(1)
            // ---- my pack of d_new = beta d + r, published for the NEXT iteration's gathers
#pragma unroll
            for (int jp = 0; jp < PPT; ++jp)
                if (packer[jp]) {
#pragma unroll
                    for (int j = 0; j < E; ++j) pd[jp].v[j] = vaypx(bt, pd[jp].v[j], pr[jp].v[j]);
                    st_pack_coh<LOCAL>(at_off(dnew_p, pack_off[jp]), pd[jp]);
                }
            // ---- q[row] = sum a_ij (beta d_old[j] + r[j]) for my two rows, matrix from LDS
            T dn_own[2], sum[2];
            sum[0] = sum[1] = vzero<T>();
            const int rounds = max(re[0] - rs[0], re[1] - rs[1]);
            if constexpr (WINDOW) {
                char *wb = reinterpret_cast<char *>(win);
#pragma unroll
                for (int jp = 0; jp < PPT; ++jp)      // my rows: already beta d + r
                    if (packer[jp]) *reinterpret_cast<Pack<T> *>(wb + (size_t)(R0 - w0) * sizeof(T) + (size_t)(t + jp * VT_) * 16) = pd[jp];
                if (ht < nhalo) {                                           // the prefetched first round of the halo
                    Pack<T> oa, ob;
#pragma unroll
                    for (int j = 0; j < E; ++j) { oa.v[j] = vaypx(bt, h_da.v[j], h_ra.v[j]); ob.v[j] = vaypx(bt, h_db.v[j], h_rb.v[j]); }
                    *reinterpret_cast<Pack<T> *>(wb + (size_t)h_pa * 16) = oa;
                    if (h_two) *reinterpret_cast<Pack<T> *>(wb + (size_t)h_pb * 16) = ob;
                }
                for (int hp = ht + 2 * kResThreads; hp < nhalo; hp += 2 * kResThreads) {     // wider halos: two packs of d and of r in flight per thread
                    const int hq = hp + kResThreads;
                    const bool two = hq < nhalo;
                    const int pa = hp < nlow ? hp : hi0 + (hp - nlow), pb = two ? (hq < nlow ? hq : hi0 + (hq - nlow)) : pa;
                    const Pack<T> da_ = ld_pack_coh(dold_p + w0 + pa * E), ra_ = ld_pack_coh(rr + w0 + pa * E);
                    const Pack<T> db_ = ld_pack_coh(dold_p + w0 + pb * E), rb_ = ld_pack_coh(rr + w0 + pb * E);
                    Pack<T> oa, ob;
#pragma unroll
                    for (int j = 0; j < E; ++j) { oa.v[j] = vaypx(bt, da_.v[j], ra_.v[j]); ob.v[j] = vaypx(bt, db_.v[j], rb_.v[j]); }
                    *reinterpret_cast<Pack<T> *>(wb + (size_t)pa * 16) = oa;
                    if (two) *reinterpret_cast<Pack<T> *>(wb + (size_t)pb * 16) = ob;
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h) dn_own[h] = *reinterpret_cast<const T *>(wb + own_off[h]);
                {
                    T dn[2][UNROLL];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) dn[h][j] = *reinterpret_cast<const T *>(wb + mo[h][j]);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) sum[h] = vfma(mv[h][j], dn[h][j], sum[h]);     // padding adds 0 * 0 (sum is never -0)
                }
                for (int kk = UNROLL; kk < rounds; kk += UNROLL) {          // rows longer than UNROLL
                    T dn[2][UNROLL], av[2][UNROLL];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) {
                            const int idx = max(min(rs[h] + kk + j, re[h] - 1), 0);
                            dn[h][j] = *reinterpret_cast<const T *>(wb + sc[idx]);
                            av[h][j] = sv[idx];
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) {
                            const T nxt = vfma(av[h][j], dn[h][j], sum[h]);
                            sum[h] = vsel(rs[h] + kk + j < re[h], nxt, sum[h]);
                        }
                }
            } else {
                T d_own[2], r_own[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d_own[h] = ld_coh_at(dold_p, own_off[h]);
                    r_own[h] = ld_coh_at(rr, own_off[h]);
                }
                for (int kk = 0; kk < rounds; kk += UNROLL) {
                    T dv[2][UNROLL], rv[2][UNROLL], av[2][UNROLL];
                    unsigned go[2][UNROLL];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) go[h][j] = (unsigned)sc[max(min(rs[h] + kk + j, re[h] - 1), 0)];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) {      // uniform base + 32-bit byte offset: one address register per gather
                            dv[h][j] = ld_coh_at(dold_p, go[h][j]);
                            rv[h][j] = ld_coh_at(rr, go[h][j]);
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) av[h][j] = sv[max(min(rs[h] + kk + j, re[h] - 1), 0)];     // LDS reads behind the gathers
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < UNROLL; ++j) {
                            const T nxt = vfma(av[h][j], vaypx(bt, dv[h][j], rv[h][j]), sum[h]);
                            sum[h] = vsel(rs[h] + kk + j < re[h], nxt, sum[h]);
                        }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) dn_own[h] = vaypx(bt, d_own[h], r_own[h]);
            }
            A dot0 = vzero<A>(), dot1 = vzero<A>();
            if (live[0]) dot0 = to_acc(vmul(dn_own[0], sum[0]));
            if (live[1]) dot1 = to_acc(vmul(dn_own[1], sum[1]));
            RES_STAMP(2)
            qs[t] = sum[0];
            qs[t + kResThreads] = sum[1];
            vblock_sum2(dot0, dot1, sh);              // drains this wave's d_new stores; its barrier also orders qs
            if ((t & 255) == 0) {
                const int rb = m * 4 + (t >> 8);
                if (rb < a.row_blocks) put_granule<LOCAL>(g_dq + (size_t)rb * W, tag0 + 2 * k + 1, dot0);
                if (rb + 2 < a.row_blocks) put_granule<LOCAL>(g_dq + (size_t)(rb + 2) * W, tag0 + 2 * k + 1, dot1);
            }
            RES_STAMP(3)
            // ---- alpha = delta / d.q (axpy2_dot_alpha_kernel's prologue); the poll is barrier 1
            T al, al_unused;
            if (!group_scalars<A, T>(a.row_blocks, sh, a.hdr, al, al_unused,
                                     [&](int i, A &v) { return get_granule(g_dq + (size_t)i * W, tag0 + 2 * k + 1, v); },
                                     [&](A dq, T &o, T &u) {
                                         const T dqT = from_acc<T>(dq);      // the reference rounds d.q to the value type before dividing (clcg.c:318-327)
                                         o = from_acc<T>(acc_div(to_acc(dlt), to_acc(dqT)));
                                         u = o;
                                     })) return;
            RES_STAMP(4)
            if (leader && t == 0) a.alpha[rhs] = al;
            // ---- x += alpha d ; r -= alpha q ; r.r partial of my 256-pack block (axpy2_dot_body)
            A acc = vzero<A>(), acc1 = vzero<A>();      // acc1: the second pack's 256-pack block (complex128)
#pragma unroll
            for (int jp = 0; jp < PPT; ++jp)
                if (packer[jp]) {
                    A part = vzero<A>();
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        const T qv = qs[(t + jp * VT_) * E + j];
                        px[jp].v[j] = vadd(px[jp].v[j], vmul(al, pd[jp].v[j]));
                        pr[jp].v[j] = vsub(pr[jp].v[j], vmul(al, qv));
                        part = vadd(part, to_acc(vmul(pr[jp].v[j], pr[jp].v[j])));
                    }
                    st_pack_coh<LOCAL>(at_off(rr, pack_off[jp]), pr[jp]);
                    if (jp == 0) acc = part; else acc1 = part;
                }
            RES_STAMP(5)
            if constexpr (PPT == 1) vblock_sum1(acc, sh);      // drains the r stores
            else vblock_sum2(acc, acc1, sh);
            RES_STAMP(6)
            if ((t & 255) == 0 && t < VT_) {
#pragma unroll
                for (int jp = 0; jp < PPT; ++jp) {
                    const int v = m * (4 / E) + (t >> 8) + jp * (VT_ / 256);      // 256-pack blocks: packs t + jp VT_ of the member's 1024 / E
                    if (v < a.P_rr) {
                        const A val = jp == 0 ? acc : acc1;
                        put_granule<LOCAL>(g_rr + (size_t)v * W, tag0 + 2 * k + 2, val);
                        if (k == a.K - 1) a.part_rr[(long long)rhs * a.P_rr + v] = val;
                    }
                }
            }
            // no barrier here: qs and sh.ws are rewritten only behind the barrier of the next prologue's group_sum
        }
#pragma unroll
        for (int jp = 0; jp < PPT; ++jp)
            if (packer[jp]) st_pack(at_off(xr, pack_off[jp]), px[jp]);
        if (stopped_at >= 0) {       // the tolerance is met after `stopped_at` iterations: delta / beta / history are already recorded
            if (leader && t == 0) {
                if (rhs == 0) *a.iter = stopped_at;
                a.hdr[kHdrStop] = (unsigned)stopped_at + 1u;
                atomicAdd(a.hdr + kHdrSolved, 1u);
            }
            continue;
        }
        // ---- delta / beta / history of the last iteration (cg_tail_kernel), by the leader's work-group
        if (leader) {
            T bfin, dnT;
            if (!group_scalars<A, T>(a.P_rr, sh, a.hdr, bfin, dnT, [&](int i, A &v) { return get_granule(g_rr + (size_t)i * W, tag0 + 2 * a.K, v); },
                                     [&](A tot, T &b, T &dn) {
                                         dn = from_acc<T>(tot);
                                         b = from_acc<T>(acc_div(to_acc(dn), to_acc(dlt)));
                                     })) return;
            if (t == 0) {
                const int it = a.it0 + a.K;
                a.beta[rhs] = bfin;
                a.delta[rhs] = dnT;
                if (it < a.history_cap) a.history[(long long)it * a.nrhs + rhs] = dnT;
                if (rhs == 0) *a.iter = it;
                atomicAdd(a.hdr + kHdrSolved, 1u);
            }
        }
    }
}

// =================================================================================================
// Wide resident loop: ONE group across the whole chip for a single right-hand side of up to 512 RPT x 256 rows (RPT = 4 or 8
// rows per thread: 2048 / 4096 rows per work-group), for systems the one-XCD loop above cannot hold -- BASELINE configs 2 and 3
// (1M rows fp64 5-point, 250k rows complex64 7-point).  The matrix lives in REGISTERS (rows of at most U entries), the window
// of beta d + r and q in LDS; members exchange through memory (write-through stores, sc1 loads: the cross-XCD form).  One
// partial sum per work-group and reduction, so this loop is NOT bit-identical to the launched loops (it is held to the oracle
// like them); the host recomputes the launched loops' r.r partials after a call, so the two can alternate on one handle.
// =================================================================================================
template <typename T> struct ResWideArgs {
    int n, nrhs, G, NG, npack, it0, K, history_cap, wcap;     // NG groups of G members; a group solves one right-hand side at a time
    long long claim_ticks;          // bound of the wait at the start line
    double tol;                     // > 0 (one right-hand side): stop before the first iteration whose sqrt|r.r| < tol (or is NaN)
    int d_ready;                    // the caller's d already is beta d + r (state of the three / four-launch loops): iteration it0 + 1 takes it as is
    const T *vals;
    const int *ptr, *cols;
    T *x, *r, *d0, *d1;
    T *alpha, *beta, *delta, *history;
    int *iter;
    unsigned *hdr;
    u64 *slot_word, *gran;          // slot_word[NG]; gran: [NG][2][G * W]
    u64 *xres;                      // [NG][16 XCDs][2 reductions][8 words]: the scalars an XCD's first work-group publishes for the others
    unsigned *xcnt;                 // [NG][8] arrivals of a group's members per XCD
    unsigned *marks;                // [NG] members whose column marks are in
    unsigned *need;                 // bitmap over the rows: somebody's row references this column from another member's slice
    const T *mdiag;                 // PCG: the diagonal preconditioner (z = m .* r), helmFE_var.py:546-586
    T *rho2;                        // PCG: [2][nrhs] rho of the last two iterations (the launched loop's parity buffer)
    long long *prof;                // diagnostics (CGAMD_RESIDENT_PROF=1)
};

// The member's value in the launched loops' member-blocked prologue sums (reduce_device.h thread_partials, K = 2 N): the
// 256-thread block sums of its N per-thread values -- block 2 i + (t >> 8): wave tree, then ((w0 + w1) + w2) + w3, exactly
// block_sum<256> of the launched kernels -- added in block order starting from zero.  Result valid in thread 0.  Every wave's
// outstanding stores are acknowledged before the barrier (see wg_sum).
template <int L> CG_DEV double lane_of(double v) {     // lane L's value in every lane (v_readlane: no LDS trip)
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), L);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int L> CG_DEV double2 lane_of(double2 v) { return make_double2(lane_of<L>(v.x), lane_of<L>(v.y)); }
template <typename A, int C, int N2> struct BlockChain {       // tot = (((0 + b_0) + b_1) + ...) + b_{N2-1}
    static CG_DEV A run(A tot, A b) { return BlockChain<A, C + 1, N2>::run(vadd(tot, lane_of<C>(b)), b); }
};
template <typename A, int N2> struct BlockChain<A, N2, N2> { static CG_DEV A run(A tot, A) { return tot; } };
CG_DEV double acc_component(double v, int) { return v; }
CG_DEV double acc_component(double2 v, int c) { return c ? v.y : v.x; }
template <typename A, int N> CG_DEV A member_sum(A (&v)[N], A (*wsn)[kResThreads / kWave]) {
    static_assert(kResThreads == 512 && kResThreads / 256 == kResWideBlocksPerRpt, "two 256-thread blocks per value");
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    // the N (complex: 2 N) wave sums four at a time (device_types.h wave_sum4: the same tree per value, a third of the adds)
    constexpr int C = (int)(sizeof(A) / sizeof(double)), ND = N * C;
    double *wd = reinterpret_cast<double *>(&wsn[0][0]);              // value i, wave w, component c at ((i * 8 + w) * C + c)
    // value i, component c = element i * C + c (no pointer cast: reading a double2 array through a double pointer is what the
    // compiler may, and did, treat as unrelated storage)
    double vd[ND];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (C == 1) vd[i] = acc_component(v[i], 0);
        else { vd[2 * i] = acc_component(v[i], 0); vd[2 * i + 1] = acc_component(v[i], 1); }
    }
    auto put = [&](int k, double g) {                                  // k: index into vd
        wd[((k / C) * (kResThreads / kWave) + wave) * C + (k % C)] = g;
    };
    // (real accumulators only: with complex ones the complex64 two-row instance returned wrong, run-to-run different sums on the
    // banded test system -- scripts/dev/wide_c64_diag.py -- although the primitives check out, scripts/microbench/wave_sum4_probe.hip;
    // not understood, so complex types keep the one-value-at-a-time tree)
    if constexpr (ND % 4 == 0 && C == 1) {
#pragma unroll
        for (int k = 0; k < ND; k += 4) {
            const double g = wave_sum4(vd[k], vd[k + 1], vd[k + 2], vd[k + 3]);
            if ((lane & 15) == 0) put(k + (lane == 0 ? 0 : lane == 16 ? 2 : lane == 32 ? 1 : 3), g);
        }
    } else if constexpr (ND % 2 == 0 && C == 1) {
#pragma unroll
        for (int k = 0; k < ND; k += 2) {
            const double g = wave_sum2(vd[k], vd[k + 1]);
            if ((lane & 31) == 0) put(k + (lane >> 5), g);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v[i] = wave_sum(v[i]);
            if (lane == 0) wsn[i][wave] = v[i];
        }
    }
    drain_stores();
    __syncthreads();
    A tot = vzero<A>();
    if (t < kWave) {
        A b = vzero<A>();
        if (lane < 2 * N) {
            const A *w = &wsn[lane >> 1][4 * (lane & 1)];
            b = vadd(vadd(vadd(w[0], w[1]), w[2]), w[3]);
        }
        tot = BlockChain<A, 0, 2 * N>::run(tot, b);
    }
    return tot;
}

// PCG: the Jacobi-preconditioned recurrence of the launched four-launch loop (vector.hip pcg_*): z = m r, rho = r.z instead of
// delta = r.r in alpha and beta, p = beta p + z; r.r is still summed (history, stopping test).  The caller's d is always "already
// beta p + z" (the launched loop's convention), so the first iteration of a launch takes it as is.
template <typename T, int RPT, int U, bool PCG>
__global__ __launch_bounds__(kResThreads) void cg_resident_wide_kernel(ResWideArgs<T> a) {
    using A = typename VT<T>::acc;
    constexpr int E = Pack<T>::N;
    constexpr int W = sizeof(A) / 4;
    constexpr int ROWS = kResThreads * RPT;          // rows per member
    constexpr int PPT = RPT / E;                     // 16-byte packs of every vector per thread
    static_assert(RPT % E == 0, "whole packs per thread");
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *qs = reinterpret_cast<T *>(dyn_smem);         // q of my rows
    T *win = qs + ROWS;                              // beta d + r of the column range of my rows; last entry: the zero cell
    __shared__ ResShared sh;
    __shared__ A wsn[RPT][kResThreads / kWave];      // wave sums of the per-block partials (member_sum)
    const int t = threadIdx.x;

    if (t == 0) {
        sh.ctl[1] = (int)atomicAdd(a.hdr + kHdrTicket, 1u);
        sh.fail = 0;
        sh.cmin = 0x7fffffff;
        sh.cmax = 0;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(sh.ctl[1]);
    const int grp = ticket / a.G, m = ticket - grp * a.G;     // tickets [g G, g G + G) are group g: one right-hand side at a time each
    if (grp >= a.NG) return;
    const bool leader = m == a.G - 1;                // drew the group's last ticket: every member is running
    if (t == 0) {                                    // the first member of the group on an XCD polls for the others there (xcd_scalars)
        const unsigned xcc = xcc_id();
        sh.ctl[0] = (int)xcc;
        sh.ctl[2] = (int)atomicAdd(a.xcnt + grp * 8 + (xcc & 7u), 1u);
    }
    __syncthreads();
    // small groups: every member polls for itself (the extra hop costs more than the pollers: 90k rows, G = 22: 5.9 against 7.1 us)
    const bool xlead = a.G <= 32 || __builtin_amdgcn_readfirstlane(sh.ctl[2]) == 0;
    const bool xpublish = a.G > 32;
    u64 *xs_rr = a.xres + ((size_t)grp * 16 + (size_t)(__builtin_amdgcn_readfirstlane(sh.ctl[0]) & 7)) * 24, *xs_dq = xs_rr + 8;   // same & 7 as the counter
    [[maybe_unused]] u64 *xs_r2 = xs_rr + 16;        // PCG: the r.r reduction (xs_rr carries rho there)
    // ---- my rows: entries into registers (rows have at most U entries: the host checked)
    const int R0 = m * ROWS;
    constexpr int UP = (U + 1) / 2;
    T mv[RPT][U];
    int mo[RPT][U];                                  // set-up only; the loop reads the 16-bit window offsets packed in mo2
    unsigned mo2[RPT][UP];
    bool live[RPT];
    int cmin = 0x7fffffff, cmax = 0;
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
        const int row = R0 + t + h * kResThreads;
        live[h] = row < a.n;
        const int rc = min(row, a.n - 1);
        const int ps = a.ptr[rc], len = live[h] ? a.ptr[rc + 1] - ps : 0;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool real = j < len;
            mv[h][j] = real ? a.vals[ps + j] : vzero<T>();
            mo[h][j] = real ? a.cols[ps + j] : -1;
            if (real) {
                cmin = min(cmin, mo[h][j]);
                cmax = max(cmax, mo[h][j]);
                // a column in another member's slice: that member must publish the entry every iteration (the others never leave its registers)
                if (mo[h][j] < R0 || mo[h][j] >= R0 + ROWS) atomicOr(a.need + (mo[h][j] >> 5), 1u << (mo[h][j] & 31));
            }
        }
    }
    atomicMin(&sh.cmin, cmin);
    atomicMax(&sh.cmax, cmax);
    drain_stores();                                  // this wave's marks are performed
    __syncthreads();
    if (t == 0) {                                    // ... and every member's, before anybody reads the bitmap
        atomicAdd(a.marks + grp, 1u);
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
            if (ld_word(a.marks + grp) >= (unsigned)a.G) break;
            if ((spins & 63) == 63 && ((wall_clock64() - t0 > a.claim_ticks && ld_word(a.hdr + kHdrNextRhs) == 0) || ld_word(a.hdr + kHdrError) != 0)) {
                atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrClaim);
                sh.fail = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (sh.fail) return;
    const int rows_m = min(ROWS, a.n - R0);
    const int w0 = min(sh.cmin, R0) & ~(E - 1);
    const int wlen = max(sh.cmax, R0 + rows_m - 1) - w0 + 1;
    const int wpacks = (wlen + E - 1) / E;
    const int nlow = (R0 - w0) / E, hi0 = (R0 + rows_m - w0) / E, nhalo = nlow + (wpacks - hi0);
    if (wlen + 1 > a.wcap) {                         // cannot happen: the host sized the window from the same pass
        if (t == 0) atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrSweep);
        return;
    }
    unsigned own_off[RPT];
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
        own_off[h] = (unsigned)(min(R0 + t + h * kResThreads, a.n - 1) - w0) * (unsigned)sizeof(T);
#pragma unroll
        for (int j = 0; j < U; ++j) mo[h][j] = mo[h][j] >= 0 ? mo[h][j] - w0 : a.wcap - 1;      // window entry, < 65536 (LDS holds fewer)
#pragma unroll
        for (int j = 0; j < UP; ++j) mo2[h][j] = (unsigned)mo[h][2 * j] | ((2 * j + 1 < U ? (unsigned)mo[h][2 * j + 1] : 0u) << 16);
    }
    if (t == 0) win[a.wcap - 1] = vzero<T>();
    u64 *g_dq = a.gran + (size_t)grp * 3 * a.G * W, *g_rr = g_dq + (size_t)a.G * W;      // g_rr: r.r partials; PCG: r.z partials
    [[maybe_unused]] u64 *g_r2 = g_rr + (size_t)a.G * W;                                    // PCG: r.r partials
    u64 *slot = a.slot_word + grp;
    char *wb = reinterpret_cast<char *>(win);
    const int ht = kResThreads - 1 - t;              // the last waves load the halo (wave 0 does the divisions)

    for (unsigned seq = 1;; ++seq) {
    // ---- which right-hand side: the leader claims, the others wait for its word.  The first word is also the start line: nobody
    // touches a vector before the whole group runs (a work-group queued behind other kernels)
    if (t == 0) {
        int rhs = -1;
        if (leader) {
            rhs = (int)atomicAdd(a.hdr + kHdrNextRhs, 1u);
            __hip_atomic_store(slot, ((u64)seq << 32) | (unsigned)rhs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                const u64 w = ld_word(slot);
                if ((unsigned)(w >> 32) == seq) { rhs = (int)(unsigned)w; break; }
                if (ld_word(a.hdr + kHdrSolved) >= (unsigned)a.nrhs) break;
                if ((spins & 63) == 63 && ((wall_clock64() - t0 > a.claim_ticks && ld_word(a.hdr + kHdrNextRhs) == 0) || ld_word(a.hdr + kHdrError) != 0)) {
                    atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrClaim);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        sh.ctl[3] = rhs;
    }
    __syncthreads();
    const int rhs = __builtin_amdgcn_readfirstlane(sh.ctl[3]);
    __syncthreads();
    if (rhs < 0 || rhs >= a.nrhs) return;
    const long long voff = (long long)rhs * a.n;
    T *xr = a.x + voff, *rr = a.r + voff, *d0r = a.d0 + voff, *d1r = a.d1 + voff;

    Pack<T> px[PPT], pr[PPT], pd[PPT];
    [[maybe_unused]] Pack<T> pm[PPT];
    bool pk[PPT], pub[PPT];
    unsigned poff[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int pl = t + j * kResThreads;          // local pack: local rows E pl .. E pl + E - 1
        const int pack = m * (ROWS / E) + pl;
        pk[j] = pack < a.npack;
        poff[j] = (unsigned)pack * 16u;
        // published every iteration only if another member reads one of its rows (E <= 4 consecutive bits of one bitmap word)
        pub[j] = pk[j] && ((ld_word(a.need + ((pack * E) >> 5)) >> ((pack * E) & 31)) & ((1u << E) - 1u)) != 0;
        if (pk[j]) {
            px[j] = ld_pack(at_off(xr, poff[j]));
            pr[j] = ld_pack(at_off(rr, poff[j]));
            pd[j] = ld_pack(at_off((a.it0 & 1) ? d1r : d0r, poff[j]));
            if constexpr (PCG) pm[j] = ld_pack(at_off(a.mdiag, poff[j]));
        }
    }
    T dlt = a.delta[rhs];                            // PCG: rho = r.z of the state the launch starts from
    [[maybe_unused]] T rrT = PCG ? a.history[(long long)min(a.it0, a.history_cap - 1) * a.nrhs + rhs] : vzero<T>();      // PCG: r.r of that state
    const unsigned tag0 = seq << 20;

    long long stamp = clock64();
    int it_end = a.it0 + a.K;                        // iterations run when this solve ends (earlier if the tolerance is met)
    bool stopped = false;
    for (int k = 0; k < a.K; ++k) {
        const int it = a.it0 + k;
        RES_STAMP(0)
        const T *dold_p = (it & 1) ? d1r : d0r;
        T *dnew_p = (it & 1) ? d0r : d1r;
        Pack<T> h_da, h_ra, h_db, h_rb;
        [[maybe_unused]] Pack<T> h_ma, h_mb;
        int h_pa = 0, h_pb = 0;
        bool h_two = false;
        auto halo_prefetch = [&]() {
            if (ht < nhalo) {
                const int hq = ht + kResThreads;
                h_two = hq < nhalo;
                h_pa = ht < nlow ? ht : hi0 + (ht - nlow);
                h_pb = h_two ? (hq < nlow ? hq : hi0 + (hq - nlow)) : h_pa;
                h_da = ld_pack_coh(dold_p + w0 + h_pa * E); h_ra = ld_pack_coh(rr + w0 + h_pa * E);
                h_db = ld_pack_coh(dold_p + w0 + h_pb * E); h_rb = ld_pack_coh(rr + w0 + h_pb * E);
                if constexpr (PCG) { h_ma = ld_pack(a.mdiag + w0 + h_pa * E); h_mb = ld_pack(a.mdiag + w0 + h_pb * E); }
            }
        };
        T bt = vzero<T>();
        const bool as_is = k == 0 && (PCG || a.d_ready);      // uniform
        if (PCG && k == 0) {                         // the caller's d is the direction: no beta
            halo_prefetch();
        } else if (it > 0 && k == 0) {               // beta of a continued run from the stored scalars
            bt = from_acc<T>(acc_div(to_acc(dlt), to_acc(a.history[(long long)(it - 1) * a.nrhs + rhs])));
            halo_prefetch();
        } else if (it > 0) {
            T dnT;
            if (!xcd_scalars<A, T>(a.G, sh, a.hdr, bt, dnT, [&](int i, A &v) { return get_granule(g_rr + (size_t)i * W, tag0 + 2 * k, v); },
                                   [&](A tot, T &b, T &dn) {
                                       dn = from_acc<T>(tot);
                                       b = from_acc<T>(acc_div(to_acc(dn), to_acc(dlt)));
                                   }, xlead, xpublish, xs_rr, tag0 + 2 * k, true, halo_prefetch)) return;
            dlt = dnT;
            if constexpr (PCG) {                     // ... and r.r of the same update: what the history records and the stopping test reads
                T r2, r2_unused;
                __syncthreads();                     // (the shared words of the reduction above are read by everybody first)
                if (!xcd_scalars<A, T>(a.G, sh, a.hdr, r2, r2_unused, [&](int i, A &v) { return get_granule(g_r2 + (size_t)i * W, tag0 + 2 * k, v); },
                                       [&](A tot, T &o, T &u) { o = from_acc<T>(tot); u = o; }, xlead, xpublish, xs_r2, tag0 + 2 * k)) return;
                rrT = r2;
            }
            if (leader && t == 0) {
                a.beta[rhs] = bt;
                a.delta[rhs] = dnT;
                if constexpr (PCG) a.rho2[(long long)(it & 1) * a.nrhs + rhs] = dnT;
                if (it < a.history_cap) a.history[(long long)it * a.nrhs + rhs] = PCG ? rrT : dnT;
            }
        } else {
            halo_prefetch();
        }
        if (a.tol > 0. && it > 0 && !(res_norm(PCG ? rrT : dlt) >= a.tol)) {     // (the reference tests after an iteration: never before the first)      // uniform: every member holds the same delta
            it_end = it;
            stopped = true;
            break;
        }
        RES_STAMP(1)
        // ---- d_new = beta d + r: my packs to memory (the neighbours' next halo) and into the window
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (pk[j]) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if constexpr (PCG) pd[j].v[e] = as_is ? pd[j].v[e] : vadd(vmul(bt, pd[j].v[e]), vmul(pm[j].v[e], pr[j].v[e]));      // p = beta p + m r (pcg_aypx_beta_kernel)
                    else pd[j].v[e] = as_is ? pd[j].v[e] : vaypx(bt, pd[j].v[e], pr[j].v[e]);
                }
                if (pub[j]) st_pack_coh<false>(at_off(dnew_p, poff[j]), pd[j]);
                *reinterpret_cast<Pack<T> *>(wb + (size_t)(R0 - w0) * sizeof(T) + (size_t)(t + j * kResThreads) * 16) = pd[j];
            }
        if (ht < nhalo) {
            Pack<T> oa, ob;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if constexpr (PCG) {
                    oa.v[e] = as_is ? h_da.v[e] : vadd(vmul(bt, h_da.v[e]), vmul(h_ma.v[e], h_ra.v[e]));
                    ob.v[e] = as_is ? h_db.v[e] : vadd(vmul(bt, h_db.v[e]), vmul(h_mb.v[e], h_rb.v[e]));
                } else {
                    oa.v[e] = as_is ? h_da.v[e] : vaypx(bt, h_da.v[e], h_ra.v[e]);
                    ob.v[e] = as_is ? h_db.v[e] : vaypx(bt, h_db.v[e], h_rb.v[e]);
                }
            }
            *reinterpret_cast<Pack<T> *>(wb + (size_t)h_pa * 16) = oa;
            if (h_two) *reinterpret_cast<Pack<T> *>(wb + (size_t)h_pb * 16) = ob;
        }
        for (int hp = ht + 2 * kResThreads; hp < nhalo; hp += 2 * kResThreads) {
            const int hq = hp + kResThreads;
            const bool two = hq < nhalo;
            const int pa = hp < nlow ? hp : hi0 + (hp - nlow), pb = two ? (hq < nlow ? hq : hi0 + (hq - nlow)) : pa;
            const Pack<T> da_ = ld_pack_coh(dold_p + w0 + pa * E), ra_ = ld_pack_coh(rr + w0 + pa * E);
            const Pack<T> db_ = ld_pack_coh(dold_p + w0 + pb * E), rb_ = ld_pack_coh(rr + w0 + pb * E);
            [[maybe_unused]] Pack<T> ma_, mb_;
            if constexpr (PCG) { ma_ = ld_pack(a.mdiag + w0 + pa * E); mb_ = ld_pack(a.mdiag + w0 + pb * E); }
            Pack<T> oa, ob;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if constexpr (PCG) {
                    oa.v[e] = as_is ? da_.v[e] : vadd(vmul(bt, da_.v[e]), vmul(ma_.v[e], ra_.v[e]));
                    ob.v[e] = as_is ? db_.v[e] : vadd(vmul(bt, db_.v[e]), vmul(mb_.v[e], rb_.v[e]));
                } else {
                    oa.v[e] = as_is ? da_.v[e] : vaypx(bt, da_.v[e], ra_.v[e]);
                    ob.v[e] = as_is ? db_.v[e] : vaypx(bt, db_.v[e], rb_.v[e]);
                }
            }
            *reinterpret_cast<Pack<T> *>(wb + (size_t)pa * 16) = oa;
            if (two) *reinterpret_cast<Pack<T> *>(wb + (size_t)pb * 16) = ob;
        }
        __syncthreads();
        RES_STAMP(2)
        // ---- q = A d_new for my rows, d.q (one partial per 256 rows, as the launched SpMV forms them)
        A dotv[RPT];
#pragma unroll
        for (int h = 0; h < RPT; ++h) {
            T sum = vzero<T>();
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const unsigned idx = (j & 1) ? mo2[h][j >> 1] >> 16 : mo2[h][j >> 1] & 0xffffu;
                sum = vfma(mv[h][j], win[idx], sum);
            }
            qs[t + h * kResThreads] = sum;
            dotv[h] = live[h] ? to_acc(vmul(*reinterpret_cast<const T *>(wb + own_off[h]), sum)) : vzero<A>();
        }
        RES_STAMP(3)
        A tot = member_sum<A, RPT>(dotv, wsn);
        if (t == 0) put_granule<false>(g_dq + (size_t)m * W, tag0 + 2 * k + 1, tot);
        RES_STAMP(4)
        T al, al_unused;
        if (!xcd_scalars<A, T>(a.G, sh, a.hdr, al, al_unused, [&](int i, A &v) { return get_granule(g_dq + (size_t)i * W, tag0 + 2 * k + 1, v); },
                               [&](A dq, T &o, T &u) {
                                   const T dqT = from_acc<T>(dq);
                                   o = from_acc<T>(acc_div(to_acc(dlt), to_acc(dqT)));
                                   u = o;
                               }, xlead, xpublish, xs_dq, tag0 + 2 * k + 1)) return;
        RES_STAMP(5)
        if (leader && t == 0) a.alpha[rhs] = al;
        A accv[PPT];                                 // r.r per pack = per thread of the launched vector kernel (one pack per thread)
        [[maybe_unused]] A accz[PPT];                // PCG: r.z per pack (pcg_axpy2_dot2_kernel: z = m r, r.z and r.r element by element)
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            accv[j] = vzero<A>();
            if constexpr (PCG) accz[j] = vzero<A>();
            if (pk[j]) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const T qv = qs[(t + j * kResThreads) * E + e];
                    px[j].v[e] = vadd(px[j].v[e], vmul(al, pd[j].v[e]));
                    pr[j].v[e] = vsub(pr[j].v[e], vmul(al, qv));
                    if constexpr (PCG) {
                        const T z = vmul(pm[j].v[e], pr[j].v[e]);
                        accz[j] = vadd(accz[j], to_acc(vmul(pr[j].v[e], z)));
                    }
                    accv[j] = vadd(accv[j], to_acc(vmul(pr[j].v[e], pr[j].v[e])));
                }
                if (pub[j]) st_pack_coh<false>(at_off(rr, poff[j]), pr[j]);
            }
        }
        RES_STAMP(6)
        if constexpr (PCG) {
            tot = member_sum<A, PPT>(accz, wsn);
            if (t == 0) put_granule<false>(g_rr + (size_t)m * W, tag0 + 2 * k + 2, tot);
            __syncthreads();                         // wsn is rewritten
            tot = member_sum<A, PPT>(accv, wsn);
            if (t == 0) put_granule<false>(g_r2 + (size_t)m * W, tag0 + 2 * k + 2, tot);
        } else {
            tot = member_sum<A, PPT>(accv, wsn);
            if (t == 0) put_granule<false>(g_rr + (size_t)m * W, tag0 + 2 * k + 2, tot);
        }
        RES_STAMP(7)
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j)
        if (pk[j]) {                                 // the whole state back to memory: x, and what was not published on the way
            st_pack(at_off(xr, poff[j]), px[j]);
            if (!pub[j]) {
                st_pack(at_off(rr, poff[j]), pr[j]);
                st_pack(at_off((it_end & 1) ? d1r : d0r, poff[j]), pd[j]);
            }
        }
    if (stopped) {                                   // delta / beta / history of iteration it_end are already recorded
        if (leader && t == 0) {
            if (rhs == 0) *a.iter = it_end;
            a.hdr[kHdrStop] = (unsigned)it_end + 1u;
            atomicAdd(a.hdr + kHdrSolved, 1u);
        }
    } else if (leader) {
        T bfin, dnT;
        if (!group_scalars<A, T>(a.G, sh, a.hdr, bfin, dnT, [&](int i, A &v) { return get_granule(g_rr + (size_t)i * W, tag0 + 2 * a.K, v); },
                                 [&](A tot, T &b, T &dn) {
                                     dn = from_acc<T>(tot);
                                     b = from_acc<T>(acc_div(to_acc(dn), to_acc(dlt)));
                                 })) return;
        [[maybe_unused]] T r2fin = vzero<T>(), r2_unused;
        if constexpr (PCG) {
            __syncthreads();
            if (!group_scalars<A, T>(a.G, sh, a.hdr, r2fin, r2_unused, [&](int i, A &v) { return get_granule(g_r2 + (size_t)i * W, tag0 + 2 * a.K, v); },
                                     [&](A tot, T &o, T &u) { o = from_acc<T>(tot); u = o; })) return;
        }
        if (t == 0) {
            const int it = a.it0 + a.K;
            a.beta[rhs] = bfin;
            a.delta[rhs] = dnT;
            if constexpr (PCG) a.rho2[(long long)(it & 1) * a.nrhs + rhs] = dnT;
            if (it < a.history_cap) a.history[(long long)it * a.nrhs + rhs] = PCG ? r2fin : dnT;
            if (rhs == 0) *a.iter = it;
            atomicAdd(a.hdr + kHdrSolved, 1u);
        }
    }
    __syncthreads();                                 // qs / sh words of this solve are done with before the next claim
    }
}

// widest column range of a `rows`-row slice (see resident_window_kernel) and the longest row: out[0], out[1]
__global__ __launch_bounds__(kResThreads) void resident_wide_scan_kernel(int n, int E, int rows, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                                         int *out) {
    __shared__ int smin, smax, slen;
    const int t = threadIdx.x, R0 = blockIdx.x * rows;
    if (t == 0) { smin = 0x7fffffff; smax = 0; slen = 0; }
    __syncthreads();
    const int r1 = min(R0 + rows, n), p0 = ptr[R0], p1 = ptr[r1];
    int cmin = 0x7fffffff, cmax = 0, len = 0;
    for (int i = p0 + t; i < p1; i += kResThreads) { cmin = min(cmin, cols[i]); cmax = max(cmax, cols[i]); }
    for (int r = R0 + t; r < r1; r += kResThreads) len = max(len, ptr[r + 1] - ptr[r]);
    atomicMin(&smin, cmin);
    atomicMax(&smax, cmax);
    atomicMax(&slen, len);
    __syncthreads();
    if (t == 0) {
        const int w0 = min(smin, R0) & ~(E - 1);
        atomicMax(out, max(smax, r1 - 1) - w0 + 1);
        atomicMax(out + 1, slen);
    }
}

// One resident launch at a time per GPU.  Inside a process: a timed mutex per device (threads that drive different GPUs -- the
// reference's RHS-sharded multi-device mode -- do not wait for each other).  Across the processes of a host (MPI ranks of the
// reference's driver sharing one GPU each call cg()): an advisory lock on a file named after the device's PCI bus id, held from
// launch to completion.  Every wait is bounded by `wait_ms` (cgamd_tune "resident_claim_ms"): a peer that holds the GPU longer --
// or a stopped / hung one -- makes the caller fall back to the launched loops instead of blocking (flock LOCK_NB, retried).
// The lock file is opened O_NOFOLLOW and must be a regular file owned by the caller; without a usable temp directory the launch
// proceeds unlocked (a group that cannot fill is then reported after its bounded wait).
}  // namespace

static std::timed_mutex g_resident_mutex[64];
ResidentLock::ResidentLock(int device, int wait_ms) {
    using clock = std::chrono::steady_clock;
    if (tune().resident_lock == 0) { held_ = true; return; }
    const auto deadline = clock::now() + std::chrono::milliseconds(wait_ms > 0 ? wait_ms : 1);
    std::timed_mutex &mx = g_resident_mutex[(unsigned)device & 63u];
    if (!mx.try_lock_until(deadline)) return;
    mutex_ = &mx;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device) != hipSuccess) { held_ = true; return; }
    for (char *c = bus; *c; ++c)
        if (*c == ':' || *c == '.' || *c == '/') *c = '_';
    const char *dir = getenv("TMPDIR");
    const std::string path = std::string(dir && *dir ? dir : "/tmp") + "/cgamd-resident-" + std::to_string((long long)getuid()) + "-" + bus + ".lock";
    int fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
    struct stat sb;
    if (fd >= 0 && (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_uid != getuid())) { close(fd); fd = -1; }
    if (fd < 0) { held_ = true; return; }          // best effort: no usable lock file
    for (;;) {
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) { fd_ = fd; held_ = true; return; }
        if (errno != EWOULDBLOCK && errno != EINTR) { close(fd); held_ = true; return; }      // locking unsupported here: proceed unlocked
        if (clock::now() >= deadline) break;
        usleep(500);
    }
    close(fd);
    mx.unlock();
    mutex_ = nullptr;
}
ResidentLock::~ResidentLock() {
    if (fd_ >= 0) { (void)flock(fd_, LOCK_UN); close(fd_); }
    if (mutex_) static_cast<std::timed_mutex *>(mutex_)->unlock();
}

// largest 4-aligned span of a 1024-row slice and the longest row (host copy of the row pointers)
static int resident_max_span(int n, const int *ptr_host, int *max_row_len) {
    int best = 0, longest = 0;
    for (int r0 = 0; r0 < n; r0 += kResRows) {
        const int p0 = ptr_host[r0] & ~3, p1 = ptr_host[std::min(r0 + kResRows, n)];
        best = std::max(best, p1 - p0);
    }
    for (int i = 0; i < n; ++i) longest = std::max(longest, ptr_host[i + 1] - ptr_host[i]);
    *max_row_len = longest;
    return best;
}

// Does the resident loop apply?  The two-launch loop's partial sums must be the ones it reproduces: 256-row d.q partials
// (row_blocks) and one 16-byte pack per thread in the vector launch (vgrid covers the packs once).
bool resident_plan(int dtype, int n, int vgrid, int row_blocks, int n_cus, const int *ptr_host, int max_window, ResidentPlan *out) {
    const int mode = tune().resident;
    if (mode == 0 || !ptr_host) return false;
    const int E = (int)(16 / dtype_size(dtype));
    if (n < 1 || n > kResMaxRows || n % E || n_cus < 8) return false;
    const int npack = n / E;
    if (vgrid != (npack + kBlock - 1) / kBlock || row_blocks != (n + kBlock - 1) / kBlock) return false;
    int max_len = 0;
    const int span = resident_max_span(n, ptr_host, &max_len);
    ResidentPlan rp;
    rp.cap = (span + 3) & ~3;
    rp.lds_bytes = (size_t)rp.cap * (dtype_size(dtype) + 4) + (size_t)kResRows * dtype_size(dtype);
    if (rp.lds_bytes > 150 * 1024) return false;
    // what is left of 150 KB holds the per-iteration window of beta d + r (at most the whole vector)
    const size_t left = (150 * 1024 - rp.lds_bytes) / dtype_size(dtype);
    rp.wcap = 0;
    if (max_window > 0 && tune().resident_window != 0 && (size_t)max_window + 4 <= left) rp.wcap = (max_window + 3 + 4) & ~3;      // + a pack of slack
    rp.lds_bytes += (size_t)rp.wcap * dtype_size(dtype);
    if (dtype == 3 && rp.wcap == 0) return false;      // complex128 only with the LDS window (the per-non-zero form would spill)
    rp.G = (n + kResRows - 1) / kResRows;
    rp.unroll = (max_len > 10 && max_len <= 12 && dtype == 0) ? 12 : (max_len > 8 && max_len <= 10 && dtype != 3) ? 10 : dtype == 3 ? 4 : 8;
    const int per_xcd = n_cus / 8;
    rp.local = mode == 2 ? 0 : (rp.G <= per_xcd ? 1 : 0);
    // groups wider than an XCD exchange through memory (write-through stores): measured SLOWER than two launches per iteration
    // (65536 rows complex64 11.4 vs 9.4 us; with four partials per polling thread 250k rows 17.8 vs 15.2,
    // profiles/r2_experiments/resident_ab.log) -- only on request
    if (!rp.local && mode != 2) return false;
    rp.lg = rp.local ? per_xcd / rp.G : n_cus / rp.G;
    if (rp.lg < 1) return false;
    rp.slots = (rp.local ? 16 : 1) * rp.lg;
    const size_t W = acc_size(dtype) / 4;
    rp.sync_bytes = (size_t)kHdrWords * 4 + (size_t)rp.slots * 8 + (size_t)rp.slots * 2 * ((size_t)row_blocks * W) * 8;
    rp.sync_bytes = (rp.sync_bytes + 15) & ~(size_t)15;
    *out = rp;
    return true;
}

static long long *g_prof_dev = nullptr;     // diagnostics only (CGAMD_RESIDENT_PROF=1; single device, single thread)
static void resident_print_prof(int K, hipStream_t st) {
    long long h[8] = {0};
    if (!g_prof_dev || hipMemcpyAsync(h, g_prof_dev, 64, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
    fprintf(stderr, "resident prof (s_memtime ticks per iteration, %d iterations): loop-top %.0f  beta-poll %.0f  gather+fma %.0f  reduce+put %.0f  "
                    "alpha-poll %.0f  update %.0f  reduce+put %.0f\n", K, (double)h[0] / K, (double)h[1] / K, (double)h[2] / K, (double)h[3] / K,
            (double)h[4] / K, (double)h[5] / K, (double)h[6] / K);
}

static void resident_print_prof_wide(int K, hipStream_t st) {
    long long h[8] = {0};
    if (!g_prof_dev || hipMemcpyAsync(h, g_prof_dev, 64, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
    fprintf(stderr, "wide resident prof (s_memtime ticks per iteration, %d iterations): loop-top %.0f  beta %.0f  window %.0f  rows %.0f  sum+put %.0f  "
                    "alpha %.0f  update %.0f  sum+put %.0f\n", K, (double)h[0] / K, (double)h[1] / K, (double)h[2] / K, (double)h[3] / K,
            (double)h[4] / K, (double)h[5] / K, (double)h[6] / K, (double)h[7] / K);
}

template <typename T, bool LOCAL, int UNROLL, bool WINDOW>
static int resident_launch_inst(const ResArgs<T> &a, size_t lds, int grid, hipStream_t st) {
    auto kern = cg_resident_kernel<T, LOCAL, UNROLL, WINDOW>;
    if (lds > 64 * 1024) {      // per launch: the attribute belongs to the current device's copy of the function (a thread may serve several)
        const size_t want = std::min<size_t>((lds + 8191) & ~(size_t)8191, 152 * 1024);      // + the static words: below the 160 KB of a CU
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("resident loop: hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kResThreads), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("resident loop launch: ") + hipGetErrorString(e));
    return CGAMD_OK;
}
template <typename T, int UNROLL>
static int resident_launch_u(const ResidentPlan &rp, const ResArgs<T> &a, int grid, hipStream_t st) {
    if (rp.wcap > 0) return rp.local ? resident_launch_inst<T, true, UNROLL, true>(a, rp.lds_bytes, grid, st) : resident_launch_inst<T, false, UNROLL, true>(a, rp.lds_bytes, grid, st);
    return rp.local ? resident_launch_inst<T, true, UNROLL, false>(a, rp.lds_bytes, grid, st) : resident_launch_inst<T, false, UNROLL, false>(a, rp.lds_bytes, grid, st);
}

template <typename T>
static int resident_impl(const ResidentPlan &rp, int n, int nrhs, const void *vals, const int *ptr, const int *cols, void *x, void *r,
                         void *d0, void *d1, void *part_rr, int P_rr, int row_blocks, const CgScalars &sc, int it0, int K, void *sync,
                         int grid, hipStream_t st, double tol) {
    using A = typename VT<T>::acc;
    constexpr int W = sizeof(A) / 4;
    ResArgs<T> a;
    a.n = n; a.nrhs = nrhs; a.G = rp.G; a.row_blocks = row_blocks; a.P_rr = P_rr; a.npack = n / Pack<T>::N;
    a.it0 = it0; a.K = K; a.history_cap = sc.history_cap; a.cap = rp.cap; a.wcap = rp.wcap;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<T *>(x); a.r = static_cast<T *>(r); a.d0 = static_cast<T *>(d0); a.d1 = static_cast<T *>(d1);
    a.part_rr = static_cast<A *>(part_rr);
    a.alpha = (T *)sc.alpha; a.beta = (T *)sc.beta; a.delta = (T *)sc.delta; a.history = (T *)sc.history; a.iter = sc.iter;
    a.hdr = static_cast<unsigned *>(sync);
    a.slot_word = reinterpret_cast<u64 *>(static_cast<char *>(sync) + kHdrWords * 4);
    a.gran = a.slot_word + rp.slots;
    a.gran_stride = row_blocks * W;
    a.lg = rp.lg;
    a.claim_ticks = (long long)std::max(1, tune().resident_claim_ms) * 100000;
    a.tol = tol;
    if (getenv("CGAMD_RESIDENT_PROF") && !g_prof_dev) CG_HIP(hipMalloc(&g_prof_dev, 64));
    a.prof = g_prof_dev;
    if (g_prof_dev) CG_HIP(hipMemsetAsync(g_prof_dev, 0, 64, st));
    // every polled word is zero at the start of every launch (tags count within the launch)
    CG_HIP(hipMemsetAsync(sync, 0, rp.sync_bytes, st));
    if constexpr (sizeof(T) == 16) {        // complex128: windowed instances, 4 entries of a row in registers (the rest of the row from LDS)
        return rp.local ? resident_launch_inst<T, true, 4, true>(a, rp.lds_bytes, grid, st) : resident_launch_inst<T, false, 4, true>(a, rp.lds_bytes, grid, st);
    } else {
        if constexpr (sizeof(T) == 4)
            if (rp.unroll == 12) return resident_launch_u<T, 12>(rp, a, grid, st);
        if (rp.unroll == 10) return resident_launch_u<T, 10>(rp, a, grid, st);
        return resident_launch_u<T, 8>(rp, a, grid, st);
    }
}

// the widest column range a member's rows touch (device pass over the column indices; synchronises `st`)
int resident_max_window(int dtype, int n, const int *ptr, const int *cols, int *scratch_dev, hipStream_t st, int *out) {
    const int E = (int)(16 / dtype_size(dtype));
    CG_HIP(hipMemsetAsync(scratch_dev, 0, 4, st));
    hipLaunchKernelGGL(resident_window_kernel, dim3((n + kResRows - 1) / kResRows), dim3(kResThreads), 0, st, n, E, ptr, cols, scratch_dev);
    CG_HIP(hipMemcpyAsync(out, scratch_dev, 4, hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    return CGAMD_OK;
}

// K iterations of every right-hand side in one launch; synchronises `st` and reports a time-out inside the launch
int run_cg_resident(int dtype, const ResidentPlan &rp, int n, int nrhs, const void *vals, const int *ptr, const int *cols, void *x, void *r,
                    void *d0, void *d1, void *part_rr, int P_rr, int row_blocks, const CgScalars &sc, int it0, int K, void *sync,
                    int n_cus, hipStream_t st, bool *untouched, double tol, int *stopped_at) {
    if (K < 1 || K >= (1 << 18)) return fail(CGAMD_ERR_INVALID, "resident loop: iteration count per launch out of range");
    if (tol > 0. && nrhs != 1) return fail(CGAMD_ERR_INVALID, "resident loop: the tolerance stop handles one right-hand side");
    if (stopped_at) *stopped_at = -1;
    int device = 0;
    CG_HIP(hipGetDevice(&device));
    ResidentLock lock(device, tune().resident_claim_ms);
    if (!lock.held()) {             // another thread / process keeps this GPU's resident launches: nothing was touched
        if (untouched) *untouched = true;
        return fail(CGAMD_ERR_STATE, "resident loop: the GPU's resident-launch lock was not free within " + std::to_string(tune().resident_claim_ms) + " ms");
    }
    // one work-group per CU; groups form from whatever is running (see the header comment).  Test hook: a grid too small for any group.
    const int grid = tune().resident_test_short_grid ? std::max(1, rp.G - 1) : n_cus;
    if (untouched) *untouched = false;
    int rc;
    switch (dtype) {
    case 0: rc = resident_impl<float>(rp, n, nrhs, vals, ptr, cols, x, r, d0, d1, part_rr, P_rr, row_blocks, sc, it0, K, sync, grid, st, tol); break;
    case 1: rc = resident_impl<double>(rp, n, nrhs, vals, ptr, cols, x, r, d0, d1, part_rr, P_rr, row_blocks, sc, it0, K, sync, grid, st, tol); break;
    case 2: rc = resident_impl<float2>(rp, n, nrhs, vals, ptr, cols, x, r, d0, d1, part_rr, P_rr, row_blocks, sc, it0, K, sync, grid, st, tol); break;
    case 3: rc = resident_impl<double2>(rp, n, nrhs, vals, ptr, cols, x, r, d0, d1, part_rr, P_rr, row_blocks, sc, it0, K, sync, grid, st, tol); break;
    default: return fail(CGAMD_ERR_INVALID, "resident loop: bad dtype");
    }
    if (rc) return rc;
    unsigned hdr[kHdrWords];
    CG_HIP(hipMemcpyAsync(hdr, sync, sizeof(hdr), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    if (getenv("CGAMD_RESIDENT_PROF")) resident_print_prof(K, st);
    if (untouched && hdr[kHdrError] == kErrClaim && hdr[kHdrSolved] == 0 && hdr[kHdrNextRhs] == 0) {
        *untouched = true;          // no group ever filled: nothing was read or written
        return fail(CGAMD_ERR_STATE, "resident loop: no group of work-groups became resident within " + std::to_string(tune().resident_claim_ms) + " ms");
    }
    if (stopped_at && hdr[kHdrStop] != 0) *stopped_at = (int)hdr[kHdrStop] - 1;
    if (hdr[kHdrError] != 0 || hdr[kHdrSolved] != (unsigned)nrhs)
        return fail(CGAMD_ERR_HIP, "resident loop: " + std::string(hdr[kHdrError] == kErrSweep ? "a partial sum" : hdr[kHdrError] == kErrClaim ? "a group's claim" : "completion") +
                                       " timed out (solved " + std::to_string(hdr[kHdrSolved]) + " of " + std::to_string(nrhs) +
                                       " right-hand sides); cgamd_tune(\"resident\", 0) selects the two-launch loop");
    return CGAMD_OK;
}


// ---- wide resident loop: host side ---------------------------------------------------------------------------------------
template <typename T, int RPT, int U, bool PCG>
static int resident_wide_launch_p(const ResWideArgs<T> &a, size_t lds, int grid, hipStream_t st) {
    auto kern = cg_resident_wide_kernel<T, RPT, U, PCG>;
    if (lds > 64 * 1024) {
        const size_t want = std::min<size_t>((lds + 8191) & ~(size_t)8191, 152 * 1024);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("wide resident loop: hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kResThreads), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("wide resident loop launch: ") + hipGetErrorString(e));
    return CGAMD_OK;
}
template <typename T, int RPT, int U>
static int resident_wide_launch(const ResWideArgs<T> &a, size_t lds, int grid, hipStream_t st) {
    return a.mdiag ? resident_wide_launch_p<T, RPT, U, true>(a, lds, grid, st) : resident_wide_launch_p<T, RPT, U, false>(a, lds, grid, st);
}

// Does the wide loop apply?  Rows of at most 8 entries, a right-hand side on G <= n_cus work-groups of 2048 / 4096 rows (1024:
// complex128), the column range of a slice (+ q) within LDS; several right-hand sides run as NG = n_cus / G concurrent groups, each
// solving one after the other where that beats the launched loops' streaming cost per right-hand side (a small model, below).
// Synchronises `st` (a pass over the matrix per candidate).
int resident_wide_plan(int dtype, int n, long long nnz, int nrhs, int n_cus, const int *ptr_dev, const int *cols_dev, int *scratch_dev,
                       hipStream_t st, ResidentWidePlan *out, bool small_ok) {
    out->ok = false;
    const int mode = tune().resident_wide;
    // systems the one-XCD loop could hold by size: only when the caller found that it does not (small_ok)
    if (mode == 0 || nrhs < 1 || n_cus < 8 || (n <= 32768 && !small_ok) || n < 2048) return CGAMD_OK;
    const int E = (int)(16 / dtype_size(dtype));
    if (n % E) return CGAMD_OK;
    const int forced = tune().resident_wide_rpt;
    ResidentWidePlan best;
    int best_rounds = 1 << 30;
    for (int rpt : {2, 4, 8}) {
        // complex128 (16-byte values): two rows per thread, the others 4 or 8; complex64 also 2 for systems of up to 32768 rows (its
        // 4-row instances spill: 34 registers in the 7-entry one, 151 with the preconditioner).  NOT beyond: with more than 32 members
        // the complex64 two-row instance returned wrong, run-to-run different results in some builds (any matrix; scripts/dev/
        // wide_c64_diag.py) -- a race I could not find; every other instance, and this one up to 32 members, is bit-stable across
        // builds and runs (tests/test_gpu_resident_wide.py)
        if (rpt % E) continue;
        if (dtype == 3 ? rpt != 2 : (rpt == 2 && !(dtype == 2 && n <= 32768))) continue;
        if (forced > 0 && dtype != 3 && rpt != forced) continue;
        const int rows = kResThreads * rpt, G = (n + rows - 1) / rows;
        if (G > std::min(n_cus, 256)) continue;
        const int NG = std::min(std::min(n_cus, 256) / G, nrhs), rounds = (nrhs + NG - 1) / NG;
        // Several rounds: a group's iteration costs t_w ~ 8 + 0.011 G us whatever the size (nothing streams from memory), i.e. t_w / NG
        // per right-hand side, while a launched iteration streams ~10 vector passes per right-hand side plus its share of the matrix at
        // ~5 TB/s (90k rows x 20 fp64: 0.75 against 2.0 us measured; 1M rows x 32 fp64: 10.7 against 16.3; 250k x 9 complex64: 6.1
        // against 5.5 -> launched).
        if (rounds > 1) {
            const double t_wide = (8.0 + 0.011 * G) / NG;
            const double t_launched = (10.0 * n * (double)dtype_size(dtype) + (double)nnz * (dtype_size(dtype) + 4) / nrhs) / 5.0e6;
            if (t_wide * 1.1 >= t_launched) continue;
        }
        if (rounds >= best_rounds) continue;
        int h[2] = {0, 0};
        CG_HIP(hipMemsetAsync(scratch_dev, 0, 8, st));
        hipLaunchKernelGGL(resident_wide_scan_kernel, dim3(G), dim3(kResThreads), 0, st, n, E, rows, ptr_dev, cols_dev, scratch_dev);
        CG_HIP(hipMemcpyAsync(h, scratch_dev, 8, hipMemcpyDeviceToHost, st));
        CG_HIP(hipStreamSynchronize(st));
        const int unroll = h[1] <= 5 ? 5 : h[1] <= 7 ? 7 : h[1] <= 8 ? 8 : 10;
        if (h[1] > 10 || (rpt == 8 && unroll != 5)) continue;          // instances: (2 | 4, 5 | 7 | 8 | 10), (8, 5)
        if (dtype == 2 && unroll == 10) continue;                      // (complex64 with 10 entries per row spills: launched loops)
        const size_t lds = ((size_t)rows + (size_t)h[0] + 8) * dtype_size(dtype);
        if (lds > 150 * 1024 || (size_t)h[0] + 8 >= 65536) continue;      // 16-bit window indices
        best.rpt = rpt; best.unroll = unroll; best.G = G; best.NG = NG;
        best.wcap = (h[0] + 3 + 4) & ~3;
        best.lds_bytes = ((size_t)rows + best.wcap) * dtype_size(dtype);
        const size_t W = acc_size(dtype) / 4;
        // header | slot words [NG] | granules [NG][3][G W] | XCD results [NG][16][3][8] | XCD arrival counters [NG][8]
        //   | marks [NG] | bitmap of published rows [n / 32]
        best.sync_bytes = (((size_t)kHdrWords * 4 + (size_t)NG * (8 + 3 * (size_t)G * W * 8 + 16 * 24 * 8 + 8 * 4) + (size_t)((NG + 3) & ~3) * 4 +
                            ((size_t)n / 32 + 2) * 4) + 15) & ~(size_t)15;
        best.ok = true;
        best_rounds = rounds;
    }
    if (best.ok) *out = best;
    return CGAMD_OK;
}

template <typename T>
static int resident_wide_impl(const ResidentWidePlan &wp, int n, int nrhs, const void *vals, const int *ptr, const int *cols, void *x, void *r,
                              void *d0, void *d1, bool d_ready, const CgScalars &sc, int it0, int K, void *sync, int grid, hipStream_t st, double tol) {
    using A = typename VT<T>::acc;
    ResWideArgs<T> a;
    a.mdiag = static_cast<const T *>(sc.pcg_m); a.rho2 = static_cast<T *>(sc.pcg_rho2);      // set: the Jacobi-preconditioned recurrence
    a.n = n; a.nrhs = nrhs; a.G = wp.G; a.NG = wp.NG; a.npack = n / Pack<T>::N; a.it0 = it0; a.K = K; a.history_cap = sc.history_cap;
    a.wcap = wp.wcap;
    a.d_ready = d_ready ? 1 : 0;
    a.tol = tol;
    a.claim_ticks = (long long)std::max(1, tune().resident_claim_ms) * 100000;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.cols = cols;
    a.x = static_cast<T *>(x); a.r = static_cast<T *>(r); a.d0 = static_cast<T *>(d0); a.d1 = static_cast<T *>(d1);
    a.alpha = (T *)sc.alpha; a.beta = (T *)sc.beta; a.delta = (T *)sc.delta; a.history = (T *)sc.history; a.iter = sc.iter;
    a.hdr = static_cast<unsigned *>(sync);
    a.slot_word = reinterpret_cast<u64 *>(static_cast<char *>(sync) + kHdrWords * 4);
    a.gran = a.slot_word + wp.NG;
    a.xres = a.gran + (size_t)wp.NG * 3 * (size_t)wp.G * (sizeof(A) / 4);
    a.xcnt = reinterpret_cast<unsigned *>(a.xres + (size_t)wp.NG * 16 * 24);
    a.marks = a.xcnt + (size_t)wp.NG * 8;
    a.need = a.marks + ((wp.NG + 3) & ~3);
    if (getenv("CGAMD_RESIDENT_PROF") && !g_prof_dev) CG_HIP(hipMalloc(&g_prof_dev, 64));
    a.prof = g_prof_dev;
    if (g_prof_dev) CG_HIP(hipMemsetAsync(g_prof_dev, 0, 64, st));
    CG_HIP(hipMemsetAsync(sync, 0, wp.sync_bytes, st));
    if constexpr (sizeof(T) == 16) {
        if (wp.unroll == 5) return resident_wide_launch<T, 2, 5>(a, wp.lds_bytes, grid, st);
        if (wp.unroll == 7) return resident_wide_launch<T, 2, 7>(a, wp.lds_bytes, grid, st);
        if (wp.unroll == 8) return resident_wide_launch<T, 2, 8>(a, wp.lds_bytes, grid, st);
        return resident_wide_launch<T, 2, 10>(a, wp.lds_bytes, grid, st);
    } else if (wp.rpt == 8) {
        return resident_wide_launch<T, 8, 5>(a, wp.lds_bytes, grid, st);
    } else if (wp.rpt == 2) {
        if constexpr (sizeof(T) == 8 && Pack<T>::N == 2 && !std::is_same<T, double>::value) {      // complex64
            if (wp.unroll == 5) return resident_wide_launch<T, 2, 5>(a, wp.lds_bytes, grid, st);
            if (wp.unroll == 7) return resident_wide_launch<T, 2, 7>(a, wp.lds_bytes, grid, st);
            if (wp.unroll == 8) return resident_wide_launch<T, 2, 8>(a, wp.lds_bytes, grid, st);
            return resident_wide_launch<T, 2, 10>(a, wp.lds_bytes, grid, st);
        }
        return fail(CGAMD_ERR_INVALID, "wide resident loop: no two-row instance for this type");
    } else {
        if (wp.unroll == 5) return resident_wide_launch<T, 4, 5>(a, wp.lds_bytes, grid, st);
        if (wp.unroll == 7) return resident_wide_launch<T, 4, 7>(a, wp.lds_bytes, grid, st);
        if (wp.unroll == 8) return resident_wide_launch<T, 4, 8>(a, wp.lds_bytes, grid, st);
        return resident_wide_launch<T, 4, 10>(a, wp.lds_bytes, grid, st);
    }
    return fail(CGAMD_ERR_INVALID, "wide resident loop: no instance");
}

// K iterations of the single right-hand side in one chip-wide launch; synchronises `st`
int run_cg_resident_wide(int dtype, const ResidentWidePlan &wp, int n, int nrhs, const void *vals, const int *ptr, const int *cols, void *x,
                         void *r, void *d0, void *d1, bool d_ready, const CgScalars &sc, int it0, int K, void *sync, int n_cus, hipStream_t st,
                         bool *untouched, double tol, int *stopped_at) {
    if (K < 1 || K >= (1 << 18)) return fail(CGAMD_ERR_INVALID, "wide resident loop: iteration count per launch out of range");
    if (tol > 0. && nrhs != 1) return fail(CGAMD_ERR_INVALID, "wide resident loop: the tolerance stop handles one right-hand side");
    if (stopped_at) *stopped_at = -1;
    if (untouched) *untouched = false;
    if (tune().resident_test_short_grid) n_cus = std::max(1, wp.G - 1);      // test hook: no group can ever fill
    else n_cus = wp.NG * wp.G;                                               // exactly the groups' work-groups
    int device = 0;
    CG_HIP(hipGetDevice(&device));
    ResidentLock lock(device, tune().resident_claim_ms);
    if (!lock.held()) {             // another thread / process keeps this GPU's resident launches: nothing was touched
        if (untouched) *untouched = true;
        return fail(CGAMD_ERR_STATE, "resident loop: the GPU's resident-launch lock was not free within " + std::to_string(tune().resident_claim_ms) + " ms");
    }
    int rc;
    switch (dtype) {
    case 0: rc = resident_wide_impl<float>(wp, n, nrhs, vals, ptr, cols, x, r, d0, d1, d_ready, sc, it0, K, sync, n_cus, st, tol); break;
    case 1: rc = resident_wide_impl<double>(wp, n, nrhs, vals, ptr, cols, x, r, d0, d1, d_ready, sc, it0, K, sync, n_cus, st, tol); break;
    case 2: rc = resident_wide_impl<float2>(wp, n, nrhs, vals, ptr, cols, x, r, d0, d1, d_ready, sc, it0, K, sync, n_cus, st, tol); break;
    case 3: rc = resident_wide_impl<double2>(wp, n, nrhs, vals, ptr, cols, x, r, d0, d1, d_ready, sc, it0, K, sync, n_cus, st, tol); break;
    default: return fail(CGAMD_ERR_INVALID, "wide resident loop: bad dtype");
    }
    if (rc) return rc;
    unsigned hdr[kHdrWords];
    CG_HIP(hipMemcpyAsync(hdr, sync, sizeof(hdr), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    if (getenv("CGAMD_RESIDENT_PROF")) resident_print_prof_wide(K, st);
    if (untouched && hdr[kHdrError] == kErrClaim && hdr[kHdrSolved] == 0 && hdr[kHdrNextRhs] == 0) {
        *untouched = true;          // no group ever passed its start line: nothing was read or written
        return fail(CGAMD_ERR_STATE, "wide resident loop: the group did not become resident within " + std::to_string(tune().resident_claim_ms) + " ms");
    }
    if (stopped_at && hdr[kHdrStop] != 0) *stopped_at = (int)hdr[kHdrStop] - 1;
    if (hdr[kHdrError] != 0 || hdr[kHdrSolved] != (unsigned)nrhs)
        return fail(CGAMD_ERR_HIP, "wide resident loop: " + std::string(hdr[kHdrError] == kErrSweep ? "a partial sum" : hdr[kHdrError] == kErrClaim ? "the start line" : "completion") +
                                       " timed out; cgamd_tune(\"resident_wide\", 0) selects the launched loops");
    return CGAMD_OK;
}

}  // namespace cgamd
