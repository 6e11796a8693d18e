// Fixed-order sum of per-work-group partials by one 1024-thread work-group (shared by the scalar kernels and the
// peer-to-peer all-reduce): thread-strided, wave tree, then the wave sums in order => bitwise reproducible.
#pragma once
#include "device_types.h"

namespace cgamd {

constexpr int kScalarBlock = 1024;

// The per-thread part of every prologue sum of P partials by a BLOCK-thread work-group (BLOCK = 256 or 1024), before the block
// sum.  K = 0: thread-strided (thread t adds p[t], p[t + BLOCK], ...).  K > 0: MEMBER-BLOCKED -- thread t < 256 adds the K
// consecutive partials p[t K .. t K + K - 1] in order, the other threads hold zero.  Handles the chip-wide resident loop can
// take over (resident.hip) use the blocked order everywhere: K = the 256-row (256-pack) blocks of one resident member, so a
// member forms the same "thread" value from its own rows and the group's sum of <= 256 member values has the launched loops'
// bits (wave tree over 64 consecutive members, then the wave sums in order; zeros change nothing).
template <int BLOCK, typename A> CG_DEV A thread_partials(const A *p, int P, int K) {
    A acc = vzero<A>();
    if (K > 0) {
        if ((int)threadIdx.x < 256) {
            const int i0 = (int)threadIdx.x * K, i1 = min(P, i0 + K);
            for (int i = i0; i < i1; ++i) acc = vadd(acc, p[i]);
        }
    } else {
        for (int i = threadIdx.x; i < P; i += BLOCK) acc = vadd(acc, p[i]);
    }
    return acc;
}

template <typename A> CG_DEV A sum_partials_block(const A *p, int grid, A *smem, int K = 0) {
    A acc = vzero<A>();
    int i = threadIdx.x;
    if (K > 0) {
        acc = thread_partials<kScalarBlock>(p, grid, K);
        i = grid;
    }
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

}  // namespace cgamd
