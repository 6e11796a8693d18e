// Device-side building blocks of the resident loops (resident.hip, slab.hip): coherent accesses between work-groups of one launch,
// tagged-word partial sums ("granules"), the scalar all-gathers of a group, fixed-order work-group sums.
#pragma once
#include <hip/hip_runtime.h>

#include "cgamd_internal.h"
#include "device_mem.h"

namespace cgamd {

typedef unsigned long long u64;
constexpr int kResThreads = 512;                // threads per work-group
constexpr int kResMaxRows = 65536;              // 256 d.q partials: one per polling thread
constexpr int kResRows = 1024;                  // rows per member: every thread walks two (4 virtual blocks of 256 rows)
constexpr long long kResSpinTicks = 400000000;  // a partial sum that does not arrive: 4 s of the 100 MHz wall clock

// header words (unsigned), zeroed before every launch
enum { kHdrTicket = 0 /* [16] */, kHdrNextRhs = 16, kHdrSolved = 17, kHdrError = 18, kHdrStop = 19 /* iterations run + 1 when the tolerance stopped the solve */,
       kHdrWords = 32 };
// error codes
enum { kErrClaim = 1, kErrSweep = 2 };

CG_DEV unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// ---- coherent accesses -----------------------------------------------------------------------------------------------
CG_DEV u64 ld_word(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // sc1: served by L2, never L1
CG_DEV unsigned ld_word(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool LOCAL> CG_DEV void st_word(u64 *p, u64 v) {
    // LOCAL: a plain store -- the line stays in this XCD's L2, where every reader of the group looks (asm: the compiler may
    // neither sink it below the spin that follows nor widen its scope, as it does for a volatile store)
    if (LOCAL) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
CG_DEV float ld_coh(const float *p) { return __uint_as_float(ld_word(reinterpret_cast<const unsigned *>(p))); }
CG_DEV double ld_coh(const double *p) { return __longlong_as_double((long long)ld_word(reinterpret_cast<const u64 *>(p))); }
CG_DEV float2 ld_coh(const float2 *p) {
    const u64 w = ld_word(reinterpret_cast<const u64 *>(p));
    return make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
}
[[maybe_unused]] CG_DEV double2 ld_coh(const double2 *p) {
    return make_double2(ld_coh(reinterpret_cast<const double *>(p)), ld_coh(reinterpret_cast<const double *>(p) + 1));
}
template <bool LOCAL, typename T> CG_DEV void st_pack_coh(T *p, const Pack<T> &v) {
    union { u32x4 raw; Pack<T> v; } u;
    u.v = v;
    if (LOCAL) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(u.raw) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(u.raw) : "memory");
}
template <typename T> CG_DEV T ld_coh_at(const T *base, unsigned byte_off) {
    return ld_coh(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off));
}
template <typename T> CG_DEV T *at_off(T *base, unsigned byte_off) { return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off); }
template <typename T> CG_DEV const T *at_off(const T *base, unsigned byte_off) { return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off); }
template <typename T> CG_DEV Pack<T> ld_pack_coh(const T *p) {      // 16 bytes past L1, as two 8-byte loads
    union { u64 w[2]; Pack<T> v; } u;
    u.w[0] = ld_word(reinterpret_cast<const u64 *>(p));
    u.w[1] = ld_word(reinterpret_cast<const u64 *>(p) + 1);
    return u.v;
}
// sqrt|delta| as the reference's tolerance test forms it (p_h-PY_C-CL.py:1364-1366: sqrt(abs(vdot(r, r))))
CG_DEV double res_norm(float v) { return sqrt(fabs((double)v)); }
CG_DEV double res_norm(double v) { return sqrt(fabs(v)); }
CG_DEV double res_norm(float2 v) { return sqrt(hypot((double)v.x, (double)v.y)); }
CG_DEV double res_norm(double2 v) { return sqrt(hypot(v.x, v.y)); }
CG_DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- granules: a partial sum as 32-bit pieces, each in an 8-byte word under the tag of its phase -----------------------
template <bool LOCAL> CG_DEV void put_granule(u64 *g, unsigned tag, double v) {
    const u64 b = (u64)__double_as_longlong(v), t = (u64)tag << 32;
    st_word<LOCAL>(g, t | (b & 0xffffffffull));
    st_word<LOCAL>(g + 1, t | (b >> 32));
}
template <bool LOCAL> CG_DEV void put_granule(u64 *g, unsigned tag, double2 v) {
    put_granule<LOCAL>(g, tag, v.x);
    put_granule<LOCAL>(g + 2, tag, v.y);
}
CG_DEV bool get_granule(const u64 *g, unsigned tag, double &v) {
    const u64 lo = ld_word(g), hi = ld_word(g + 1);
    v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
    return (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag;
}
CG_DEV bool get_granule(const u64 *g, unsigned tag, double2 &v) {
    const u64 w0 = ld_word(g), w1 = ld_word(g + 1), w2 = ld_word(g + 2), w3 = ld_word(g + 3);      // one round trip
    v.x = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
    v.y = __longlong_as_double((long long)((w2 & 0xffffffffull) | (w3 << 32)));
    return (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag && (unsigned)(w2 >> 32) == tag && (unsigned)(w3 >> 32) == tag;
}

struct ResShared {
    double2 ws[2 * kResThreads / kWave];   // wave sums of the virtual blocks (real types use .x): [value 0 | value 1][wave]
    double2 gs[4];                     // wave sums of a group_sum
    double2 bcT[2];                    // broadcast of the scalars a group_scalars call produces
    int ctl[4];
    int fail;
    int cmin, cmax;                    // column range of this member's rows
};
CG_DEV double lane0(double v) { return __shfl(v, 0, kWave); }
CG_DEV double2 lane0(double2 v) { return make_double2(__shfl(v.x, 0, kWave), __shfl(v.y, 0, kWave)); }
template <typename A> CG_DEV A &as_acc(double2 &v);
template <> CG_DEV double &as_acc<double>(double2 &v) { return v.x; }
template <> CG_DEV double2 &as_acc<double2>(double2 &v) { return v; }

// A scalar of the recurrence from P (<= 256) partial sums.  The sum is formed in the order of the two-launch kernels'
// prologues -- thread t < 256 holds 0 + p[t], wave tree, then ((w0 + w1) + w2) + w3 -- and `finish(sum, o0, o1)` (the
// divisions: double precision, ~80 instructions) runs in wave 0 only; the results reach the other waves through LDS.
// `fetch(i, v)` returns false while partial i is not there yet (granules): the poll is the barrier between the members; two
// polls are kept in flight, so a word is seen one L2 trip after it lands, not one and a half.  Called by the whole
// work-group; false = timed out (error word set).  The shared words are rewritten only after the next work-group barrier
// (there is one between any two calls).
struct NoMid { CG_DEV void operator()() const {} };
template <typename A, typename T, typename F, typename G, typename M = NoMid>
CG_DEV bool group_scalars(int P, ResShared &sh, unsigned *hdr, T &o0, T &o1, F fetch, G finish, bool HAS_MID = false, M mid = M()) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    A v = vzero<A>();
    if (t < 256 && (t & ~(kWave - 1)) < P) {      // waves 0..3 that own partials poll
        const bool mine = t < P;
        const int i = mine ? t : 0;
        const long long t0 = wall_clock64();
        A g0 = vzero<A>(), g1 = vzero<A>();
        bool ok0 = fetch(i, g0), ok1;
        for (unsigned spins = 0;; ++spins) {
            ok1 = fetch(i, g1);
            if (__all(ok0 || !mine)) { v = g0; break; }
            ok0 = fetch(i, g0);
            if (__all(ok1 || !mine)) { v = g1; break; }
            if ((spins & 63) == 63 && (wall_clock64() - t0 > kResSpinTicks || ld_word(hdr + kHdrError) != 0)) {
                if (lane == 0) { atomicCAS(hdr + kHdrError, 0u, (unsigned)kErrSweep); sh.fail = 1; }
                break;
            }
        }
        v = wave_sum(mine ? vadd(vzero<A>(), v) : vzero<A>());
        if (lane == 0 && wave > 0) as_acc<A>(sh.gs[wave]) = v;
    }
    if (HAS_MID || P > kWave) __syncthreads();    // uniform: the members' words are in (and the other polling waves' sums)
    if (HAS_MID) mid();                           // loads that need the barrier but not the scalars: in flight behind the divisions
    if (wave == 0) {
        A s = lane0(v);
#pragma unroll
        for (int w = 1; w < 4; ++w) s = vadd(s, w * kWave < P ? as_acc<A>(sh.gs[w]) : vzero<A>());
        T r0, r1;
        finish(s, r0, r1);
        if (lane == 0) { *reinterpret_cast<T *>(&sh.bcT[0]) = r0; *reinterpret_cast<T *>(&sh.bcT[1]) = r1; }
    }
    __syncthreads();
    o0 = *reinterpret_cast<const T *>(&sh.bcT[0]);
    o1 = *reinterpret_cast<const T *>(&sh.bcT[1]);
    return sh.fail == 0;
}

// the 256-thread block sums of the two-launch kernels for the virtual blocks of this work-group: thread t holds values of
// virtual blocks (t >> 8) [v0] and 2 + (t >> 8) [v1]; results valid in the first thread of every 256-thread half
// (block_sum<256>: own wave sum, then + the 3 following waves' in order).  Every wave waits for its own outstanding stores
// before the barrier (their acknowledgement overlaps the wave sums): whatever a leader signals afterwards, the
// work-group's d / r stores have reached L2 (memory, in the write-through form).
template <typename A> CG_DEV void vblock_sum2(A &v0, A &v1, ResShared &sh) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    if (lane == 0) { as_acc<A>(sh.ws[wave]) = v0; as_acc<A>(sh.ws[8 + wave]) = v1; }
    drain_stores();
    __syncthreads();
    if ((t & 255) == 0) {
#pragma unroll
        for (int i = 1; i < 4; ++i) { v0 = vadd(v0, as_acc<A>(sh.ws[wave + i])); v1 = vadd(v1, as_acc<A>(sh.ws[8 + wave + i])); }
    }
}

template <typename A> CG_DEV void vblock_sum1(A &v0, ResShared &sh) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    v0 = wave_sum(v0);
    if (lane == 0) as_acc<A>(sh.ws[wave]) = v0;
    drain_stores();
    __syncthreads();
    if ((t & 255) == 0) {
#pragma unroll
        for (int i = 1; i < 4; ++i) v0 = vadd(v0, as_acc<A>(sh.ws[wave + i]));
    }
}

// The two scalars of a reduction for a chip-wide group.  Only ONE work-group per XCD (its first arriver) polls the members'
// partial sums in memory and does the arithmetic; it hands the results to the other work-groups of its XCD through that
// XCD's L2 (plain store, sc1-load poll): G pollers on the fabric become 8 (at G = 245 the all-poll form costs ~10 us of
// an iteration).  The members' vector stores are in memory before their partial sums are (write-through, drained), so a
// work-group that learns the result from its XCD's poller may load them.
template <typename A, typename T, typename F, typename G, typename M = NoMid>
CG_DEV bool xcd_scalars(int P, ResShared &sh, unsigned *hdr, T &o0, T &o1, F fetch, G finish, bool xlead, bool xpublish, u64 *xslot,
                        unsigned tag, bool has_mid = false, M mid = M()) {
    const int t = threadIdx.x;
    if (xlead) {
        if (!group_scalars<A, T>(P, sh, hdr, o0, o1, fetch, finish, has_mid, mid)) return false;
        if (xpublish && t == 0) {            // 32 payload bits per word under the tag: 2 scalars x sizeof(T) / 4 words (<= 8)
            unsigned bits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            __builtin_memcpy(bits, &o0, sizeof(T));
            __builtin_memcpy(bits + 4, &o1, sizeof(T));
            const u64 tg = (u64)tag << 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) st_word<true>(xslot + i, tg | bits[i]);
        }
        return true;
    }
    if (t < kWave) {
        const long long t0 = wall_clock64();
        u64 w = 0;
        for (unsigned spins = 0;; ++spins) {
            w = ld_word(xslot + (t & 7));
            if (__all((unsigned)(w >> 32) == tag)) break;
            if ((spins & 63) == 63 && (wall_clock64() - t0 > kResSpinTicks || ld_word(hdr + kHdrError) != 0)) {
                if (t == 0) { atomicCAS(hdr + kHdrError, 0u, (unsigned)kErrSweep); sh.fail = 1; }
                break;
            }
        }
        unsigned bits[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bits[i] = (unsigned)__shfl(w, i, kWave);
        if (t == 0) {
            __builtin_memcpy(&sh.bcT[0], bits, sizeof(T));
            __builtin_memcpy(&sh.bcT[1], bits + 4, sizeof(T));
        }
    }
    __syncthreads();
    if (has_mid) mid();
    o0 = *reinterpret_cast<const T *>(&sh.bcT[0]);
    o1 = *reinterpret_cast<const T *>(&sh.bcT[1]);
    return sh.fail == 0;
}

// sum over the work-group in a fixed order (wave tree, then the 8 wave sums in order); every wave's outstanding stores are
// acknowledged before the barrier; result valid in thread 0
template <typename A> CG_DEV A wg_sum(A v, ResShared &sh) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    v = wave_sum(v);
    if (lane == 0) as_acc<A>(sh.ws[wave]) = v;
    drain_stores();
    __syncthreads();
    if (t == 0) {
#pragma unroll
        for (int i = 1; i < kResThreads / kWave; ++i) v = vadd(v, as_acc<A>(sh.ws[i]));
    }
    return v;
}


}  // namespace cgamd
