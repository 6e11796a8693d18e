// Device-side pieces of the peer-to-peer mailbox protocol shared by p2p.hip and cg1.hip.
#pragma once
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"

namespace cgamd {

// =================================================================================================
// Peer-to-peer communication over xGMI without RCCL (optional backend of the row-partitioned loop).
// RCCL's latency (10-20 us per small collective) bounds strong scaling of a 40 us iteration; here each rank owns
// an UNCACHED, IPC-shared "mailbox" that its peers write directly:
//     [0,4096)      reduction slots  slot[which in 0..1][source rank] = {value.x, value.y, epoch, pad} (32 B)
//     [4096,6144)   halo flags       flag[source rank] = epoch of the last complete boundary push
//     [6144,8192)   error word
//     [8192,16384)  single-reduction loop (cg1.hip): slot[parity][source rank] = {r.r (x, y), w.r (x, y), epoch, pad} (64 B)
//     [16384,...)   halo entries     laid out exactly like the halo part of d_ext
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
constexpr int kMbSlots = 0, kMbHaloFlags = 4096, kMbError = 6144, kMbCg1 = 8192, kMbHalo = (int)kMailboxHeader;
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

}  // namespace cgamd
