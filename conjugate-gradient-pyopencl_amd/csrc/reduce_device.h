// Fixed-order sum of per-work-group partials by one 1024-thread work-group (shared by the scalar kernels and the
// peer-to-peer all-reduce): thread-strided, wave tree, then the wave sums in order => bitwise reproducible.
#pragma once
#include "device_types.h"

namespace cgamd {

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

}  // namespace cgamd
