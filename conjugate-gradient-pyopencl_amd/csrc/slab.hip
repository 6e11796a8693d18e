// Resident CG loop with a STREAMED matrix ("slab loop"): every iteration of an iterate() call inside ONE launch, for systems whose
// VECTORS fit the chip's registers while the matrix does not -- one rank's slab of the row-partitioned headline system (1.25M rows,
// 8.7M non-zeros: the launched loops move 183 MB per iteration there, all of it through the Infinity Cache, and sit at the 36 us
// that costs) and single-GPU systems between the chip-wide resident loop's reach (~1M rows) and ~3M rows.
//
//   * One work-group (512 threads) per CU; member m owns rows [m ROWS, (m + 1) ROWS), ROWS a multiple of 1024, and keeps its entries of
//     x, r, d and q IN REGISTERS for the whole launch (RPT rows per thread).  Per iteration the only vector traffic is d: written
//     through once (the other members' gathers), gathered where a row references another member's column.
//   * The matrix is streamed every iteration, 1024 rows per step: a step's slice of aValues and of the column codes / aCols is
//     fetched with the row-block kernels' coalesced 16-byte loads (spmv_device.h) into REGISTERS one step ahead, parked in LDS,
//     and every thread walks two rows out of LDS.  Columns inside the member's own row range are served from an LDS copy of its
//     d; the others ("far": the z-neighbours of a 3-D stencil, the rows next to the member's boundary) by sc1 loads from memory.
//   * d ping-pongs between two buffers by iteration parity; a member publishes "my d of iteration k is in memory" in a tagged
//     flag word after its stores are acknowledged, and a member waits only for the members that own one of its far columns (a
//     bit mask found at set-up) -- no grid-wide barrier; the two reductions of the iteration are the resident loops' granule
//     all-gathers with one poller per XCD (resident_device.h), which are the only chip-wide synchronisations.
//   * Same recurrence, same per-row summation order as every other loop (clcg.c:297-419); partial sums are per member, so the
//     results are held to the oracle like the chip-wide resident loop's (fp64 1e-10 on delta_k), not bit-identical to the launched loops,
//     and are bitwise reproducible run to run (member m owns chunk m whatever CU it runs on).
//   * Every wait is bounded; a launch whose members cannot all become resident gives up BEFORE any vector is read or written
//     and the handle continues with the launched loops.
// Hand-offs follow MI355X_MICROARCH.md "Valid forms" row 1: all stores of handed-off bytes sc1, every storing wave drains
// (s_waitcnt vmcnt(0)) before the work-group barrier, ONE lane stores the sc1 flag; consumers poll with sc1 loads, pass a
// work-group barrier, and load the bytes with sc1 loads only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "cgamd_internal.h"
#include "device_mem.h"
#include "launch_util.h"
#include "p2p_device.h"
#include "resident_device.h"
#include "spmv_device.h"

namespace cgamd {
namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kSlabStep = kResThreads;      // rows per step: one per thread
constexpr int kSlabRpt = 12;                // steps at most: 6144 rows per member

template <typename T> struct SlabArgs {
    int n, G, rows_m, nsteps, it0, K, history_cap;     // rows_m / nsteps: of the LARGEST member (LDS layout); uniform members when mstart is null
    const int *mstart;              // [G + 1] first row of every member (multiples of 1024), or null = m rows_m
    const int *gmember;             // [ceil(n / 1024)] member that owns a 1024-row granule (with mstart)
    int vcap;                       // LDS entries of one value slice (multiple of 4)
    int ccap;                       // LDS bytes of the member's column codes
    long long nnz;
    long long claim_ticks;
    const T *vals;
    const int *ptr;
    const unsigned char *codes;     // one byte per non-zero: aCols[j] = row + dict[codes[j]] (index_codes.hip); padded by 64 bytes
    const int *dict;
    const unsigned char *vcodes;    // one byte per non-zero: aValues[j] == vdict[vcodes[j]] (build_value_codes), or null = the values are streamed
    const T *vdict;
    T *x, *r;
    T *din;                         // the caller's d (already beta d + r: state of the three / four-launch loops); rewritten at the end
    T *ds0, *ds1;                   // [n + n_halo] each: where iteration k's d is published (k & 1); with peers: uncached, IPC-shared, the peers
                                    // write their boundary entries into the tail [n, n + n_halo)
    // row-partitioned run (nranks > 1 or a rank that is its own peer): peer-to-peer mailboxes (p2p_device.h)
    int n_halo, nranks, rank, n_peers;
    char *const *mailbox;           // [nranks] mapped mailbox bases
    const int *peer_rank, *send_off, *send_count, *recv_count;     // [n_peers]
    const int *send_index;          // concatenated local rows to send, ascending per peer
    T *const *push_dst;             // [2][n_peers]: where my entries for peer p land in ITS ds0 / ds1 tail
    unsigned *pushcnt, *contrib;    // [n_peers] members that have pushed (running) / that push at all
    u64 *halo_epoch, *red_seq;      // exchange epoch of the handle (advanced by K at the end), launch sequence of the scalar slots
    T *alpha, *beta, *delta, *history;
    int *iter;
    unsigned *hdr;
    u64 *gran;                      // [2][G * W]
    u64 *xres;                      // [16 XCDs][2 reductions][8 words]
    unsigned *xcnt;                 // [8]
    unsigned *marks;                // [1] members past their set-up (+ abort bit)
    u64 *dflag;                     // [G] iteration whose d a member has published
    long long *prof;                // diagnostics (CGAMD_RESIDENT_PROF=<member>): phase times of that member, s_memtime ticks
    int prof_m;
};
#define SLAB_STAMP(i)                                                                  \
    if (a.prof && t == 0 && m == a.prof_m) {                                             \
        const long long now__ = clock64();                                               \
        a.prof[i] += now__ - stamp;                                                      \
        stamp = now__;                                                                   \
    }

// one step's VALUE slice (512 rows: at most 4096 entries, one staging round) on its way from memory to LDS: the lane-interleaved
// 16-byte chunks of stage_slice_ilv (spmv_device.h), held in registers between issue and the LDS store
template <typename T> struct SliceRegs {
    using V = typename Chunk16<T>::V;
    static constexpr int EPC = 16 / (int)sizeof(T), NV = 4 / EPC;
    V ch[2][NV];
};
// All loads of the step loop are UNCONDITIONAL buffer loads: a lane with nothing to fetch passes an offset beyond the buffer's
// num_records and the hardware returns 0 without touching memory.  Loads inside branches (per lane `if (far)`, tail handling at the
// end of the matrix) made the compiler's wait-count pass fall back to s_waitcnt vmcnt(0) at every use -- nothing stayed in flight
// and the product ran at 3 TB/s (profiles/r3_experiments/slab_waitcnt.md).
constexpr unsigned kOob = 0x80000000u;
CG_DEV __amdgpu_buffer_rsrc_t slab_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <typename T> CG_DEV void slab_stage_load(__amdgpu_buffer_rsrc_t rv, int cfirst, int p1, SliceRegs<T> &g) {
    using V = typename Chunk16<T>::V;
    constexpr int EPC = SliceRegs<T>::EPC, NV = SliceRegs<T>::NV, BLOCK = kResThreads;
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int ev = cfirst + rg * 4 * BLOCK + wave * 4 * kWave + (k * kWave + lane) * EPC;
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rv, ev < p1 ? (unsigned)ev * (unsigned)sizeof(T) : kOob, 0, 0);
            g.ch[rg][k] = __builtin_bit_cast(V, w);
        }
    }
}
// the whole staging round goes to LDS (8 x 512 entries: the buffer is that large), no per-chunk condition
template <typename T> CG_DEV void slab_stage_store(const SliceRegs<T> &g, T *sv) {
    using V = typename Chunk16<T>::V;
    constexpr int EPC = SliceRegs<T>::EPC, NV = SliceRegs<T>::NV, BLOCK = kResThreads;
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            *reinterpret_cast<V *>(sv + rg * 4 * BLOCK + wave * 4 * kWave + (k * kWave + lane) * EPC) = g.ch[rg][k];
    }
}

CG_DEV void st_coh(float *p, float v) { __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CG_DEV void st_coh(double *p, double v) { __hip_atomic_store(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CG_DEV void st_coh(float2 *p, float2 v) {
    __hip_atomic_store(reinterpret_cast<u64 *>(p), ((u64)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The far entries of one row (columns outside the member's own rows), issued one step ahead of the row walk that uses them
template <typename T, int UNROLL> struct FarRegs {
    T v[UNROLL];
};
CG_DEV void slab_ld_far(__amdgpu_buffer_rsrc_t rd, unsigned off, float &v) { v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, off, 0, 16)); }
CG_DEV void slab_ld_far(__amdgpu_buffer_rsrc_t rd, unsigned off, double &v) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rd, off, 0, 16);
    v = __builtin_bit_cast(double, w);
}
CG_DEV void slab_ld_far(__amdgpu_buffer_rsrc_t rd, unsigned off, float2 &v) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rd, off, 0, 16);
    v = make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
}
// row: global row; s, e: its entries relative to the member's first entry (the LDS-resident codes); one sc1 buffer load per slot:
// far entries fetch their column, the others (and every slot when !valid) pass an out-of-range offset
template <typename T, int UNROLL>
CG_DEV void slab_far_issue(const unsigned char *scode, const int *sdict, int s, int e, int row, int R0, int rows_m, __amdgpu_buffer_rsrc_t rd, bool valid,
                           FarRegs<T, UNROLL> &f) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        const int idx = min(s + j, e - 1);
        const int c = row + sdict[scode[max(idx, 0)]];
        const bool far = valid && s + j < e && (unsigned)(c - R0) >= (unsigned)rows_m;
        slab_ld_far(rd, far ? (unsigned)c * (unsigned)sizeof(T) : kOob, f.v[j]);
    }
}
// the row's sum, left to right in CSR order like every other kernel: values from the staged slice (vs: the row's first value),
// near columns from the member's LDS copy of d, far ones from `f`
template <typename T, int UNROLL>
CG_DEV T slab_row(const T *vs, const unsigned char *scode, const int *sdict, int s, int e, int row, int R0, int rows_m, const T *dl,
                  const FarRegs<T, UNROLL> &f) {
    T sum = vzero<T>();
    T av[UNROLL], xv[UNROLL];
    bool far[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        const int k = min(j, e - s - 1);
        const int c = row + sdict[scode[max(s + k, 0)]];
        const unsigned off = (unsigned)(c - R0);
        far[j] = off >= (unsigned)rows_m;
        av[j] = vs[max(k, 0)];
        xv[j] = dl[far[j] ? 0u : off];
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        const T xx = vsel(far[j], f.v[j], xv[j]);
        const T nxt = vfma(av[j], xx, sum);
        sum = vsel(s + j < e, nxt, sum);
    }
    return sum;
}

// the same with the values from the LDS-resident value codes and their dictionary (nothing of the matrix is streamed)
template <typename T, int UNROLL>
CG_DEV T slab_row_vc(const unsigned char *svcode, const T *sdictv, const unsigned char *scode, const int *sdict, int s, int e, int row, int R0,
                     int rows_m, const T *dl, const FarRegs<T, UNROLL> &f) {
    T sum = vzero<T>();
    T av[UNROLL], xv[UNROLL];
    bool far[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        const int idx = max(s + min(j, e - s - 1), 0);
        const int c = row + sdict[scode[idx]];
        const unsigned off = (unsigned)(c - R0);
        far[j] = off >= (unsigned)rows_m;
        av[j] = sdictv[svcode[idx]];
        xv[j] = dl[far[j] ? 0u : off];
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        const T xx = vsel(far[j], f.v[j], xv[j]);
        const T nxt = vfma(av[j], xx, sum);
        sum = vsel(s + j < e, nxt, sum);
    }
    return sum;
}

// The sum of `local` over all ranks in rank order (bitwise identical everywhere), for a whole wave: lane s publishes this rank's
// value in rank s's mailbox and waits for rank s's value in its own (p2p_device.h: scalar slots of the single-reduction region,
// parity = tag & 1, the epoch word carries the launch sequence and the tag).  Every polling work-group of a rank publishes the
// same bits, so it does not matter whose store a peer sees.
template <typename T, typename A> CG_DEV A slab_rank_sum(const SlabArgs<T> &a, u64 seq, unsigned tag, A local, int *fail) {
    if (a.nranks <= 1 && a.n_peers == 0) return local;
    // granule form (resident_device.h): each 8-byte word carries 32 payload bits under a 32-bit tag, so a consumer has data and
    // validity in ONE round trip and the producer needs no store ordering (payload + epoch word cost three dependent trips)
    const int lane = threadIdx.x & (kWave - 1);
    const u64 tg = (u64)((unsigned)(seq << 21) | tag) << 32;
    const long long slot_off = kMbCg1 + ((long long)(tag & 1u) * 64) * 64;
    const double2 v = to_acc2(local);
    double vx = 0., vy = 0.;
    if (lane < a.nranks) {
        u64 *slot = reinterpret_cast<u64 *>(a.mailbox[lane] + slot_off + (long long)a.rank * 64);
        const u64 bx = (u64)__double_as_longlong(v.x), by = (u64)__double_as_longlong(v.y);
        st_sys(slot, tg | (bx & 0xffffffffull));
        st_sys(slot + 1, tg | (bx >> 32));
        st_sys(slot + 2, tg | (by & 0xffffffffull));
        st_sys(slot + 3, tg | (by >> 32));
        const u64 *in = reinterpret_cast<const u64 *>(a.mailbox[a.rank] + slot_off + (long long)lane * 64);
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
            const u64 w0 = ld_sys(in), w1 = ld_sys(in + 1), w2 = ld_sys(in + 2), w3 = ld_sys(in + 3);
            if ((w0 >> 32) == (tg >> 32) && (w1 >> 32) == (tg >> 32) && (w2 >> 32) == (tg >> 32) && (w3 >> 32) == (tg >> 32)) {
                vx = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
                vy = __longlong_as_double((long long)((w2 & 0xffffffffull) | (w3 << 32)));
                break;
            }
            if ((spins & 63) == 63 && (wall_clock64() - t0 > kResSpinTicks || ld_word(a.hdr + kHdrError) != 0)) {
                st_sys(reinterpret_cast<u64 *>(a.mailbox[a.rank] + kMbError), 2ULL);
                atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrSweep);
                *fail = 1;
                break;
            }
        }
    }
    double2 tot = make_double2(0., 0.);
    for (int r = 0; r < a.nranks; ++r) { tot.x += __shfl(vx, r, kWave); tot.y += __shfl(vy, r, kWave); }
    return from_acc2<A>(tot);
}

// VC: the matrix values are one-byte codes into a dictionary of at most 256 entries (build_value_codes), held in LDS like the column
// codes: the member streams NOTHING of the matrix, the value slices, their staging registers and the barrier per step are gone.
template <typename T, int RPT, int UNROLL, bool VC>
__global__ __launch_bounds__(kResThreads) void cg_slab_kernel(SlabArgs<T> a) {
    using A = typename VT<T>::acc;
    constexpr int W = sizeof(A) / 4;
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    T *dl = reinterpret_cast<T *>(dyn_smem);                    // d of my rows [rows_m]
    T *sval = dl + a.rows_m;                                    // two value slices [2][vcap]; VC: the value codes of all my rows [ccap]
    unsigned char *svcode = reinterpret_cast<unsigned char *>(sval);
    unsigned char *scode = VC ? svcode + a.ccap : reinterpret_cast<unsigned char *>(sval + 2 * (size_t)a.vcap);     // the column codes of all my rows [ccap]
    __shared__ T sdictv[VC ? 256 : 1];
    __shared__ ResShared sh;
    __shared__ int sdict[256];
    __shared__ unsigned depmask[8];
    __shared__ int sbound[2 * RPT + 2];     // per step: first entry (4-aligned) and end of its slice
    __shared__ int sendlo[64], sendhi[64];  // my part of every peer's send list
    __shared__ unsigned scontrib[64];
    __shared__ u64 sepoch[2];               // [0] exchange epoch at the start of the launch, [1] launch sequence of the scalar slots
    __shared__ int has_halo;
    const int t = threadIdx.x;

    if (t == 0) {
        sh.ctl[1] = (int)atomicAdd(a.hdr + kHdrTicket, 1u);
        sh.fail = 0;
    }
    if (t < 8) depmask[t] = 0u;
    if (t < 256) sdict[t] = a.dict[t];
    if (t == 0) {
        has_halo = 0;
        sepoch[0] = a.halo_epoch ? *a.halo_epoch : 0;
        sepoch[1] = a.red_seq ? *a.red_seq + 1 : 1;
    }
    __syncthreads();
    const int m = __builtin_amdgcn_readfirstlane(sh.ctl[1]);
    if (m >= a.G) return;
    const bool leader = m == a.G - 1;
    if (t == 0) {
        const unsigned xcc = xcc_id();
        sh.ctl[0] = (int)xcc;
        sh.ctl[2] = (int)atomicAdd(a.xcnt + (xcc & 7u), 1u);
    }
    __syncthreads();
    const bool xlead = a.G <= 32 || __builtin_amdgcn_readfirstlane(sh.ctl[2]) == 0;
    const bool xpublish = a.G > 32;
    u64 *xs_rr = a.xres + (size_t)(__builtin_amdgcn_readfirstlane(sh.ctl[0]) & 7) * 16, *xs_dq = xs_rr + 8;

    // ---- my rows: the codes of all of them into LDS, slice bounds per step, the members that own a far column of mine
    // members of unequal size (slab_partition): the ones that push to / wait for a peer rank own fewer rows, so that their SpMV ends
    // with the others' although it starts a push later
    const int R0 = a.mstart ? a.mstart[m] : m * a.rows_m, R1 = min(a.mstart ? a.mstart[m + 1] : R0 + a.rows_m, a.n);
    const int nsteps = a.mstart ? (a.mstart[m + 1] - a.mstart[m]) / kSlabStep : a.nsteps;      // even (members start at multiples of 1024)
    const int c0 = a.ptr[R0] & ~3, cend = a.ptr[R1];            // my entries are [c0, cend)
    for (int o = 4 * t; o < cend - c0; o += 4 * kResThreads) {
        *reinterpret_cast<unsigned *>(scode + o) = *reinterpret_cast<const unsigned *>(a.codes + c0 + o);
        if constexpr (VC) *reinterpret_cast<unsigned *>(svcode + o) = *reinterpret_cast<const unsigned *>(a.vcodes + c0 + o);
    }
    if constexpr (VC) {
        if (t < 256) sdictv[t] = a.vdict[t];
    }
    if (t <= RPT) {
        const int ra = min(R0 + t * kSlabStep, R1);
        sbound[2 * t] = a.ptr[ra] & ~3;
        sbound[2 * t + 1] = a.ptr[min(ra + kSlabStep, R1)];
    }
    __syncthreads();
    // (start, length) of my row in every step, packed: start relative to the member's first entry c0 (< 2^26), length (< 32)
    unsigned rowinfo[RPT];
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
        const int row = R0 + h * kSlabStep + t;
        unsigned info = 0;
        if (h < nsteps && row < R1) {
            const int ps = a.ptr[row], pe = a.ptr[row + 1];
            info = ((unsigned)(ps - c0) << 5) | (unsigned)(pe - ps);
            for (int j = ps; j < pe; ++j) {
                const int c = row + sdict[scode[j - c0]];
                if ((unsigned)(c - R0) >= (unsigned)(R1 - R0)) {       // (the last member's R0 + rows_m may lie beyond n: halo columns start at n)
                    if (c >= a.n) {
                        has_halo = 1;                        // a column another RANK owns: it arrives in the tail of ds0 / ds1
                    } else {
                        const int owner = a.gmember ? a.gmember[c >> 10] : c / a.rows_m;
                        atomicOr(&depmask[owner >> 5], 1u << (owner & 31));
                    }
                }
            }
        }
        rowinfo[h] = info;
    }
    if (t < a.n_peers) {                     // the entries of peer t's send list that are rows of mine (the list is ascending)
        const int *idx = a.send_index + a.send_off[t];
        const int cnt = a.send_count[t];
        int lo = 0, hi = cnt;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (idx[mid] < R0) lo = mid + 1; else hi = mid; }
        const int first = lo;
        hi = cnt;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (idx[mid] < R1) lo = mid + 1; else hi = mid; }
        sendlo[t] = first;
        sendhi[t] = lo;
        if (lo > first) atomicAdd(a.contrib + t, 1u);
    }
    drain_stores();                          // my count is performed before thread 0 announces this member at the start line
    __syncthreads();
    // ---- start line: nobody touches a vector before every member runs (a work-group queued behind other kernels).  One word
    // holds the count of members that are ready (low 16 bits) and an ABORT bit that can only be set while the count is short:
    // either every member sees the full count without the bit and goes on, or every member sees the bit and leaves -- never both.
    if (t == 0) {
        constexpr unsigned kAbort = 0x80000000u;
        unsigned v = atomicAdd(a.marks, 1u) + 1u;
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
            if (v & kAbort) { sh.fail = 1; break; }
            if ((v & 0xffffu) >= (unsigned)a.G) break;
            if ((spins & 63) == 63 && wall_clock64() - t0 > a.claim_ticks) {
                const unsigned old = atomicCAS(a.marks, v, v | kAbort);         // only while the count is the short one just read
                if (old == v) { atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrClaim); sh.fail = 1; break; }
                v = old;
                continue;
            }
            __builtin_amdgcn_s_sleep(8);
            v = ld_word(a.marks);
        }
        if (!sh.fail && leader) atomicAdd(a.hdr + kHdrNextRhs, 1u);       // past the start line: vectors are in use from here on
    }
    if (t < a.n_peers) scontrib[t] = ld_word(a.contrib + t);       // every member is past its set-up: the counts are final
    __syncthreads();
    if (sh.fail) return;
    const u64 E0 = sepoch[0], seq = sepoch[1];

    // x, r, q of my rows in registers; d of my rows in LDS (dl) -- the row walks read it there anyway
    T px[RPT], pr[RPT], pq[RPT];
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
        const int row = R0 + h * kSlabStep + t;
        const bool lv = h < nsteps && row < R1;
        px[h] = lv ? a.x[row] : vzero<T>();
        pr[h] = lv ? a.r[row] : vzero<T>();
        pq[h] = vzero<T>();
        if (h < nsteps) dl[h * kSlabStep + t] = lv ? a.din[row] : vzero<T>();
    }
    u64 *g_dq = a.gran, *g_rr = a.gran + (size_t)a.G * W;
    T dlt = a.delta[0];
    // Value slices: global step G (= k nsteps + h) walks slice G mod nsteps out of LDS buffer G & 1; the slices of steps G + 1 and
    // G + 2 are in flight in registers (set (G + 1) & 1 and set G & 1): a step stores the older one into the other LDS buffer and
    // issues the loads of step G + 3 into the freed set.  Two slices (~56 KB per CU) in flight are what the memory latency under
    // load (~2.3 us measured) needs for the chip's streaming rate; one (the first version) ran the product at 3 TB/s.
    SliceRegs<T> slA, slB;          // slA: slices of odd global steps, slB: of even ones (nsteps is even: the parity of a step is that of h)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rv = slab_rsrc(a.vals, (unsigned)((a.nnz * (long long)sizeof(T) + 15) & ~15LL));
    const unsigned dbytes = (unsigned)(a.n + a.n_halo) * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rd0 = slab_rsrc(a.ds0, dbytes), rd1 = slab_rsrc(a.ds1, dbytes);
    if constexpr (!VC) {
        slab_stage_load<T>(rv, sbound[0], sbound[1], slB);
        slab_stage_store<T>(slB, sval);
        slab_stage_load<T>(rv, sbound[2], sbound[3], slA);                                     // step 1
        slab_stage_load<T>(rv, sbound[2 * (2 % nsteps)], sbound[2 * (2 % nsteps) + 1], slB);   // step 2
    }
    __syncthreads();

    long long stamp = clock64();
    for (int k = 0; k < a.K; ++k) {
        const int it = a.it0 + k;
        SLAB_STAMP(0)
        T *dpub = (k & 1) ? a.ds1 : a.ds0;           // where iteration k's d is published
        T bt = vzero<T>();
        if (k > 0) {
            T dnT;
            if (!xcd_scalars<A, T>(a.G, sh, a.hdr, bt, dnT, [&](int i, A &v) { return get_granule(g_rr + (size_t)i * W, 2u * k, v); },
                                   [&](A tot, T &b, T &dn) {
                                       dn = from_acc<T>(slab_rank_sum<T, A>(a, seq, 2u * k, tot, &sh.fail));
                                       b = from_acc<T>(acc_div(to_acc(dn), to_acc(dlt)));
                                   }, xlead, xpublish, xs_rr, 2u * k)) return;
            dlt = dnT;
            if (leader && t == 0) {
                a.beta[0] = bt;
                a.delta[0] = dnT;
                if (it < a.history_cap) a.history[it] = dnT;
            }
        }
        SLAB_STAMP(1)
        // ---- d = beta d + r (k = 0: the caller's d as it is): to memory for the members that gather it, and into my LDS copy
#pragma unroll
        for (int h = 0; h < RPT; ++h) {
            const int li = h * kSlabStep + t;
            if (h < nsteps && R0 + li < R1) {
                const T dn = k > 0 ? vaypx(bt, dl[li], pr[h]) : dl[li];
                dl[li] = dn;
                st_coh(dpub + (R0 + li), dn);
            }
        }
        if (a.n_peers > 0) {                         // my rows on the peers' send lists: straight into their buffer of this parity
            __syncthreads();
            for (int p = 0; p < a.n_peers; ++p) {
                T *dst = a.push_dst[(k & 1) * a.n_peers + p];
                const int *idx = a.send_index + a.send_off[p];
                const int lo = sendlo[p], hi = sendhi[p];
                if (hi <= lo) continue;              // uniform
                int rows[RPT];                       // at most rows_m entries are mine: all index loads in flight together (one by
                                                     // one, a member that pushes a whole plane was 5 us behind the others)
#pragma unroll
                for (int u = 0; u < RPT; ++u) rows[u] = idx[min(lo + t + u * kResThreads, hi - 1)];
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const int i = lo + t + u * kResThreads;
                    if (i < hi) st_sys_val(dst + i, dl[rows[u] - R0]);
                }
            }
        }
        drain_stores();                              // (also lands the value slices in flight: they are stored to LDS next anyway)
        __syncthreads();
        if (t == 0) st_word<false>(a.dflag + m, (u64)k + 1);
        if (t < a.n_peers && sendhi[t] > sendlo[t]) {   // the last member to have pushed its part raises my flag in the peer's mailbox
            const unsigned old = atomicAdd(a.pushcnt + t, 1u);
            if ((old + 1u) % scontrib[t] == 0u)
                st_sys(reinterpret_cast<u64 *>(a.mailbox[a.peer_rank[t]] + kMbHaloFlags) + a.rank, E0 + (u64)k + 1);
        }
        SLAB_STAMP(2)
        if (t < 256) {                               // the members whose d I gather have published iteration k
            const bool dep = t < a.G && ((depmask[t >> 5] >> (t & 31)) & 1u) != 0;
            const long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                const bool ok = !dep || ld_word(a.dflag + t) >= (u64)k + 1;
                if (__all(ok)) break;
                if ((spins & 63) == 63 && (wall_clock64() - t0 > kResSpinTicks || ld_word(a.hdr + kHdrError) != 0)) {
                    if ((t & (kWave - 1)) == 0) { atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrSweep); sh.fail = 1; }
                    break;
                }
            }
        } else if (t < 256 + kWave && has_halo) {    // ... and the peers whose entries my rows reference
            const int p = t - 256;
            if (p < a.n_peers && a.recv_count[p] > 0) {
                const char *mb = a.mailbox[a.rank];
                if (!spin_until(reinterpret_cast<const u64 *>(mb + kMbHaloFlags) + a.peer_rank[p], E0 + (u64)k + 1, mb)) {
                    st_sys(reinterpret_cast<u64 *>(const_cast<char *>(mb) + kMbError), 1ULL);
                    atomicCAS(a.hdr + kHdrError, 0u, (unsigned)kErrSweep);
                    sh.fail = 1;
                }
            }
        }
        __syncthreads();
        if (sh.fail) return;
        SLAB_STAMP(3)
        // ---- q = A d, 512 rows per step.  Entering step h: value slice h is in LDS buffer h & 1, slice h + 1 in flight (registers), the
        // far entries of step h are in flight in `far[h & 1]`.  Program order per step keeps every load one step ahead of its use
        // and older than what the step's own waits let pass (vmcnt retires in order).
        // The step loop is a REAL loop over pairs of steps (the fully unrolled form let the scheduler interleave the steps and spill
        // hundreds of registers); the per-step registers are picked / updated with uniform selects.  One step, in program order:
        // (a) far entries of step h + 1 leave memory, (b) the older slice in flight -> the other LDS buffer, (c) the slice of
        // global step G + 3 leaves memory into the freed registers, (d) my row of step h: every load is issued at least one step
        // before its use and nothing is conditional, so the wait counts the compiler derives keep all of it in flight.
        const __amdgpu_buffer_rsrc_t rd = (k & 1) ? rd1 : rd0;
        FarRegs<T, UNROLL> fA, fB;       // far entries of even / odd steps
        {
            const unsigned i0 = rowinfo[0];
            slab_far_issue<T, UNROLL>(scode, sdict, (int)(i0 >> 5), (int)(i0 >> 5) + (int)(i0 & 31u), R0 + t, R0, R1 - R0, rd, true, fA);
        }
        auto one_step = [&](int h, int bcur, SliceRegs<T> &sl, FarRegs<T, UNROLL> &fcur, FarRegs<T, UNROLL> &fnext) {
            unsigned info = 0, info2 = 0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                info = h == i ? rowinfo[i] : info;
                info2 = h + 1 == i ? rowinfo[i] : info2;
            }
            const int s = (int)(info >> 5), e = s + (int)(info & 31u);
            const int s2 = (int)(info2 >> 5), e2 = s2 + (int)(info2 & 31u);
            slab_far_issue<T, UNROLL>(scode, sdict, s2, e2, R0 + (h + 1) * kSlabStep + t, R0, R1 - R0, rd, h + 1 < nsteps, fnext);
            T qv;
            if constexpr (VC) {
                qv = slab_row_vc<T, UNROLL>(svcode, sdictv, scode, sdict, s, e, R0 + h * kSlabStep + t, R0, R1 - R0, dl, fcur);
            } else {
                slab_stage_store<T>(sl, sval + (size_t)(bcur ^ 1) * a.vcap);
                const int j3 = (h + 3) % nsteps;           // (past the last iteration: a redundant load, never stored)
                slab_stage_load<T>(rv, sbound[2 * j3], sbound[2 * j3 + 1], sl);
                qv = slab_row<T, UNROLL>(sval + (size_t)bcur * a.vcap + (e > s ? (c0 + s) - sbound[2 * h] : 0), scode, sdict, s, e,
                                         R0 + h * kSlabStep + t, R0, R1 - R0, dl, fcur);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) pq[i] = vsel(h == i, qv, pq[i]);
            if constexpr (!VC) __syncthreads();      // slice h is done with; slice h + 1 is complete in the other buffer
        };
#pragma unroll 1
        for (int h = 0; h < nsteps; h += 2) {
            one_step(h, 0, slA, fA, fB);
            one_step(h + 1, 1, slB, fB, fA);
        }
        SLAB_STAMP(4)
        // ---- alpha = delta / d.q
        A dot = vzero<A>();
#pragma unroll
        for (int h = 0; h < RPT; ++h)
            if (h < nsteps) dot = vadd(dot, to_acc(vmul(dl[h * kSlabStep + t], pq[h])));
        A tot = wg_sum(dot, sh);
        if (t == 0) put_granule<false>(g_dq + (size_t)m * W, 2u * k + 1, tot);
        SLAB_STAMP(5)
        T al, al_unused;
        if (!xcd_scalars<A, T>(a.G, sh, a.hdr, al, al_unused, [&](int i, A &v) { return get_granule(g_dq + (size_t)i * W, 2u * k + 1, v); },
                               [&](A dq, T &o, T &u2) {
                                   const T dqT = from_acc<T>(slab_rank_sum<T, A>(a, seq, 2u * k + 1, dq, &sh.fail));
                                   o = from_acc<T>(acc_div(to_acc(dlt), to_acc(dqT)));
                                   u2 = o;
                               }, xlead, xpublish, xs_dq, 2u * k + 1)) return;
        if (leader && t == 0) a.alpha[0] = al;
        SLAB_STAMP(6)
        // ---- x += alpha d ; r -= alpha q ; r.r
        A acc = vzero<A>();
#pragma unroll
        for (int h = 0; h < RPT; ++h)
            if (h < nsteps) {
                px[h] = vadd(px[h], vmul(al, dl[h * kSlabStep + t]));
                pr[h] = vsub(pr[h], vmul(al, pq[h]));
                acc = vadd(acc, to_acc(vmul(pr[h], pr[h])));
            }
        tot = wg_sum(acc, sh);
        if (t == 0) put_granule<false>(g_rr + (size_t)m * W, 2u * k + 2, tot);
        SLAB_STAMP(7)
    }
    // ---- the last iteration's beta, d = beta d + r (the launched loops' convention), state back to memory
    {
        T bt, dnT;
        if (!xcd_scalars<A, T>(a.G, sh, a.hdr, bt, dnT, [&](int i, A &v) { return get_granule(g_rr + (size_t)i * W, 2u * a.K, v); },
                               [&](A tot, T &b, T &dn) {
                                   dn = from_acc<T>(slab_rank_sum<T, A>(a, seq, 2u * a.K, tot, &sh.fail));
                                   b = from_acc<T>(acc_div(to_acc(dn), to_acc(dlt)));
                               }, xlead, xpublish, xs_rr, 2u * a.K)) return;
#pragma unroll
        for (int h = 0; h < RPT; ++h) {
            const int row = R0 + h * kSlabStep + t;
            if (h < nsteps && row < R1) {
                a.x[row] = px[h];
                a.r[row] = pr[h];
                a.din[row] = vaypx(bt, dl[h * kSlabStep + t], pr[h]);
            }
        }
        if (leader && t == 0) {
            const int it = a.it0 + a.K;
            a.beta[0] = bt;
            a.delta[0] = dnT;
            if (it < a.history_cap) a.history[it] = dnT;
            *a.iter = it;
            if (a.halo_epoch) *a.halo_epoch = E0 + (u64)a.K;     // K exchanges happened
            if (a.red_seq) *a.red_seq = seq;
            atomicAdd(a.hdr + kHdrSolved, 1u);
        }
    }
}

template <typename T, int RPT, int UNROLL, bool VC>
int slab_launch_vc(const SlabArgs<T> &a, size_t lds, hipStream_t st) {
    auto kern = cg_slab_kernel<T, RPT, UNROLL, VC>;
    if (lds > 64 * 1024) {
        const size_t want = std::min<size_t>((lds + 8191) & ~(size_t)8191, 152 * 1024);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("slab loop: hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(a.G), dim3(kResThreads), lds, st, a);
    return check_launch("cg_slab");
}
template <typename T, int RPT, int UNROLL>
int slab_launch(const SlabArgs<T> &a, size_t lds, hipStream_t st) {
    return a.vcodes ? slab_launch_vc<T, RPT, UNROLL, true>(a, lds, st) : slab_launch_vc<T, RPT, UNROLL, false>(a, lds, st);
}

}  // namespace

// rows of a member, members, LDS: false when the loop does not apply (too many rows for the chip's registers, rows too long)
bool slab_plan(int dtype, int n, int n_cus, const SpmvPlan &plan, bool coded, SlabPlan *out) {
    *out = SlabPlan();
    // one-byte column codes are what lets a member keep the pattern of its rows in LDS for the whole launch (and issue the far
    // gathers a step ahead); 16-byte values would spill the staging registers
    if (n_cus < 8 || plan.kind != 5 || dtype == CGAMD_C128 || !coded || plan.max_row < 1 || plan.max_row > 8) return false;
    const size_t vs = dtype_size(dtype);
    // rows per member: the smallest multiple of 1024 (an even number of 512-row steps) that covers n with at most n_cus members
    long long rows_m = ((long long)n + n_cus - 1) / n_cus;
    rows_m = (rows_m + 2 * kSlabStep - 1) / (2 * kSlabStep) * (2 * kSlabStep);
    if (rows_m > (long long)kSlabRpt * kSlabStep) return false;
    const int G = (int)((n + rows_m - 1) / rows_m);
    if (G < 2 || G > 256) return false;
    // a step's value slice: 512 rows = two 256-row blocks (+ alignment slack), one staging round of 8 x 512 entries at most
    if (((long long)plan.max_span * 2 + 8) > 8LL * kResThreads) return false;
    const long long vcap = 8LL * kResThreads;       // the whole staging round is stored
    const long long ccap = (((long long)plan.max_span * (rows_m / 256) + 8 + 15) & ~15LL) + 16;
    // (streamed values: two slices + the column codes; value codes: both code arrays -- the launch takes whichever form the matrix has)
    const size_t lds = (size_t)rows_m * vs + std::max(2 * (size_t)vcap * vs + (size_t)ccap, 2 * (size_t)ccap) + 64;
    if (lds > 150 * 1024) return false;
    out->ok = true; out->rows_m = (int)rows_m; out->G = G; out->cap = (int)vcap; out->ccap = (int)ccap; out->lds_bytes = (lds + 15) & ~(size_t)15;
    out->nsteps = (int)(rows_m / kSlabStep);
    out->unroll = plan.max_row <= 5 ? 5 : plan.max_row <= 7 ? 7 : 8;
    out->sync_bytes = slab_sync_bytes(G);
    return true;
}
size_t slab_sync_bytes(int G) {
    constexpr int W4 = 4;       // granule words per partial (complex: 4, real: 2): sized for the larger
    return (size_t)kHdrWords * 4 + (size_t)2 * G * W4 * 8 + 16 * 16 * 8 + 8 * 4 + 16 + (size_t)G * 8 + 2 * 64 * 4 + 256;
}

// Members of unequal size for a rank with peers.  boundary[g] != 0: the 1024-row granule g holds a row that is pushed to a peer (for
// the symmetric patterns of CG also the rows that read the peer's entries).  Such members publish, push ~rows x 8 bytes with
// system-scope stores and wait for the peer's flag before their SpMV can start (~6 us on the 1/8 slab of the headline system, where
// they were the critical path of every iteration): they get `trim` granules fewer than the others.  Fills mstart [G + 1] / gmember
// [granules]; returns the number of members, 0 when more than max_members would be needed.
int slab_partition(const SlabPlan &sp, int n, const std::vector<char> &boundary, int trim, int max_members, std::vector<int> *mstart,
                   std::vector<int> *gmember) {
    const int granules = (n + 1023) / 1024, cap_i = sp.rows_m / 1024, cap_b = std::max(1, cap_i - trim);
    mstart->clear();
    gmember->assign((size_t)granules, 0);
    int g = 0, members = 0;
    while (g < granules) {
        int take = std::min(cap_i, granules - g);
        bool touches = false;
        for (int k = 0; k < take; ++k) touches = touches || boundary[(size_t)(g + k)] != 0;
        if (touches) {
            // up to cap_b granules; a member that starts in the interior stops short of the boundary instead of swallowing it
            int first_b = 0;
            while (!boundary[(size_t)(g + first_b)]) ++first_b;
            take = first_b > 0 ? first_b : std::min(cap_b, granules - g);
        }
        mstart->push_back(g * 1024);
        for (int k = 0; k < take; ++k) (*gmember)[(size_t)(g + k)] = members;
        g += take;
        if (++members > max_members) return 0;
    }
    mstart->push_back(granules * 1024);
    return members;
}

static long long *g_slab_prof = nullptr;     // diagnostics only (CGAMD_RESIDENT_PROF=<member>; single device, single thread)
template <typename T>
static int slab_impl(const SlabPlan &sp, int n, long long nnz, const void *vals, const int *ptr, const unsigned char *codes, const int *dict,
                     void *x, void *r, void *din, void *ds0, void *ds1, const SlabComm *cm, const CgScalars &sc, int it0, int K, void *sync,
                     hipStream_t st) {
    using A = typename VT<T>::acc;
    SlabArgs<T> a;
    a.n = n; a.G = sp.G; a.rows_m = sp.rows_m; a.nsteps = sp.nsteps; a.it0 = it0; a.K = K; a.history_cap = sc.history_cap;
    a.mstart = sp.mstart; a.gmember = sp.gmember;
    a.vcap = sp.cap; a.ccap = sp.ccap;
    a.nnz = nnz;
    a.claim_ticks = (long long)std::max(1, tune().resident_claim_ms) * 100000;
    a.vals = static_cast<const T *>(vals); a.ptr = ptr; a.codes = codes; a.dict = dict;
    a.vcodes = (sp.vcodes && sp.vdict && tune().value_codes != 0) ? sp.vcodes : nullptr; a.vdict = static_cast<const T *>(sp.vdict);
    a.x = static_cast<T *>(x); a.r = static_cast<T *>(r); a.din = static_cast<T *>(din); a.ds0 = static_cast<T *>(ds0); a.ds1 = static_cast<T *>(ds1);
    a.alpha = (T *)sc.alpha; a.beta = (T *)sc.beta; a.delta = (T *)sc.delta; a.history = (T *)sc.history; a.iter = sc.iter;
    a.n_halo = 0; a.nranks = 1; a.rank = 0; a.n_peers = 0;
    a.mailbox = nullptr; a.peer_rank = a.send_off = a.send_count = a.recv_count = a.send_index = nullptr;
    a.push_dst = nullptr; a.halo_epoch = nullptr; a.red_seq = nullptr;
    if (cm) {
        a.n_halo = cm->n_halo; a.nranks = cm->nranks; a.rank = cm->rank; a.n_peers = cm->n_peers;
        a.mailbox = cm->mailbox; a.peer_rank = cm->peer_rank; a.send_off = cm->send_off; a.send_count = cm->send_count;
        a.recv_count = cm->recv_count; a.send_index = cm->send_index;
        a.push_dst = reinterpret_cast<T *const *>(cm->push_dst);
        a.halo_epoch = reinterpret_cast<u64 *>(cm->halo_epoch); a.red_seq = reinterpret_cast<u64 *>(cm->red_seq);
    }
    char *base = static_cast<char *>(sync);
    a.hdr = reinterpret_cast<unsigned *>(base);
    a.gran = reinterpret_cast<u64 *>(base + kHdrWords * 4);
    a.xres = a.gran + (size_t)2 * sp.G * (sizeof(A) / 4);
    a.xcnt = reinterpret_cast<unsigned *>(a.xres + 16 * 16);
    a.marks = a.xcnt + 8;
    a.dflag = reinterpret_cast<u64 *>(a.marks + 4);
    a.pushcnt = reinterpret_cast<unsigned *>(a.dflag + sp.G);
    a.contrib = a.pushcnt + 64;
    CG_HIP(hipMemsetAsync(sync, 0, sp.sync_bytes, st));
    if (getenv("CGAMD_RESIDENT_PROF") && !g_slab_prof) CG_HIP(hipMalloc(&g_slab_prof, 64));
    a.prof = g_slab_prof;
    a.prof_m = getenv("CGAMD_RESIDENT_PROF") ? atoi(getenv("CGAMD_RESIDENT_PROF")) % sp.G : 0;
    if (g_slab_prof) CG_HIP(hipMemsetAsync(g_slab_prof, 0, 64, st));
    // two register budgets: up to 10 steps (5120 rows per member: one rank's slab of the headline system) without a spill in fp64
    if (sp.nsteps <= 10) {
        if (sp.unroll == 5) return slab_launch<T, 10, 5>(a, sp.lds_bytes, st);
        if (sp.unroll == 7) return slab_launch<T, 10, 7>(a, sp.lds_bytes, st);
        return slab_launch<T, 10, 8>(a, sp.lds_bytes, st);
    }
    if (sp.unroll == 5) return slab_launch<T, kSlabRpt, 5>(a, sp.lds_bytes, st);
    if (sp.unroll == 7) return slab_launch<T, kSlabRpt, 7>(a, sp.lds_bytes, st);
    return slab_launch<T, kSlabRpt, 8>(a, sp.lds_bytes, st);
}

// K iterations (it0 + 1 ... it0 + K) in one launch; state in and out is the three / four-launch loops' (x, r, d already
// beta d + r in `din`, delta / beta / alpha / history / iter); their r.r partials are NOT maintained.  ds0 / ds1: n + n_halo values each
// (with peers: inside the rank's IPC-shared mailbox allocation).  Synchronises `st`.
int run_cg_slab(int dtype, const SlabPlan &sp, int n, long long nnz, const void *vals, const int *ptr, const unsigned char *codes,
                const int *dict, void *x, void *r, void *din, void *ds0, void *ds1, const SlabComm *cm, const CgScalars &sc, int it0, int K,
                void *sync, hipStream_t st, bool *untouched) {
    if (!codes || !dict) return fail(CGAMD_ERR_STATE, "slab loop: needs the one-byte column codes");
    if (K < 1 || K >= (1 << 20)) return fail(CGAMD_ERR_INVALID, "slab loop: iteration count per launch out of range");
    if (untouched) *untouched = false;
    int device = 0;
    CG_HIP(hipGetDevice(&device));
    ResidentLock lock(device, tune().resident_claim_ms);
    if (!lock.held()) {
        if (untouched) *untouched = true;
        return fail(CGAMD_ERR_STATE, "slab loop: the GPU's resident-launch lock was not free within " + std::to_string(tune().resident_claim_ms) + " ms");
    }
    int rc;
    switch (dtype) {
    case 0: rc = slab_impl<float>(sp, n, nnz, vals, ptr, codes, dict, x, r, din, ds0, ds1, cm, sc, it0, K, sync, st); break;
    case 1: rc = slab_impl<double>(sp, n, nnz, vals, ptr, codes, dict, x, r, din, ds0, ds1, cm, sc, it0, K, sync, st); break;
    case 2: rc = slab_impl<float2>(sp, n, nnz, vals, ptr, codes, dict, x, r, din, ds0, ds1, cm, sc, it0, K, sync, st); break;
    default: return fail(CGAMD_ERR_INVALID, "slab loop: bad dtype");
    }
    if (rc) return rc;
    unsigned hdr[kHdrWords];
    CG_HIP(hipMemcpyAsync(hdr, sync, sizeof(hdr), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
    if (g_slab_prof) {
        long long h[8] = {0};
        if (hipMemcpy(h, g_slab_prof, 64, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "slab prof (s_memtime ticks per iteration, %d iterations): top %.0f  beta-wait %.0f  d-update+publish %.0f  neighbours %.0f  spmv %.0f  "
                            "dq-sum %.0f  alpha-wait %.0f  update+rr-sum %.0f\n", K, (double)h[0] / K, (double)h[1] / K, (double)h[2] / K, (double)h[3] / K,
                    (double)h[4] / K, (double)h[5] / K, (double)h[6] / K, (double)h[7] / K);
    }
    if (untouched && hdr[kHdrError] == kErrClaim && hdr[kHdrSolved] == 0 && hdr[kHdrNextRhs] == 0) {
        *untouched = true;          // the members never all ran: nothing was read or written
        return fail(CGAMD_ERR_STATE, "slab loop: the work-groups did not all become resident within " + std::to_string(tune().resident_claim_ms) + " ms");
    }
    if (hdr[kHdrError] != 0 || hdr[kHdrSolved] != 1u)
        return fail(CGAMD_ERR_HIP, "slab loop: " + std::string(hdr[kHdrError] == kErrSweep ? "a partial sum or a neighbour's flag" : hdr[kHdrError] == kErrClaim ? "the start line" : "completion") +
                                       " timed out");
    return CGAMD_OK;
}

}  // namespace cgamd
