// HBM streaming microbenchmark for MI355X: which launch shape / unroll / cache policy reaches the
// achievable read and copy bandwidth.  Build: hipcc -O3 --offload-arch=gfx950 bw.hip -o bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U, bool NT, bool BLOCKED>
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *__restrict__ p, long long n16, unsigned *out) {
    u32x4 acc = {0, 0, 0, 0};
    const long long T = (long long)gridDim.x * 256;
    if (BLOCKED) {
        // each work-group owns one contiguous range, lanes interleave inside it
        const long long per = (n16 + gridDim.x - 1) / gridDim.x;
        const long long b = per * blockIdx.x, e = b + per < n16 ? b + per : n16;
        for (long long i = b + threadIdx.x; i < e; i += 256 * U) {
            u32x4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { long long j = i + 256 * k; v[k] = j < e ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : acc; }
#pragma unroll
            for (int k = 0; k < U; ++k) acc ^= v[k];
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += T * U) {
            u32x4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { long long j = i + T * k; v[k] = j < n16 ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : acc; }
#pragma unroll
            for (int k = 0; k < U; ++k) acc ^= v[k];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ p, u32x4 *__restrict__ q, long long n16) {
    const long long T = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += T * U) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { long long j = i + T * k; if (j < n16) v[k] = NT ? __builtin_nontemporal_load(p + j) : p[j]; }
#pragma unroll
        for (int k = 0; k < U; ++k) { long long j = i + T * k; if (j < n16) { if (NT) __builtin_nontemporal_store(v[k], q + j); else q[j] = v[k]; } }
    }
}

template <typename F> double time_us(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const long long bytes = 1600000000LL, n16 = bytes / 16;
    u32x4 *p, *q; unsigned *out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 2, bytes));
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384, 65536, (int)((n16 + 255) / 256)};
    printf("read %lld bytes\n", bytes);
#define RUN(U, NT, BL) for (int g : grids) { if (BL && g > 65536) continue; double us = time_us([&] { hipLaunchKernelGGL((read_kernel<U, NT, BL>), dim3(g), dim3(256), 0, 0, p, n16, out); }, 7); \
        printf("read  U=%d nt=%d blocked=%d grid=%8d : %8.1f us %7.1f GB/s\n", U, NT, BL, g, us, bytes / us / 1e3); }
    RUN(1, false, false) RUN(2, false, false) RUN(4, false, false) RUN(8, false, false)
    RUN(1, true, false) RUN(4, true, false)
    RUN(1, false, true) RUN(4, false, true) RUN(4, true, true)
#define RUNC(U, NT) for (int g : grids) { double us = time_us([&] { hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(g), dim3(256), 0, 0, p, q, n16 / 2); }, 7); \
        printf("copy  U=%d nt=%d grid=%8d : %8.1f us %7.1f GB/s (r+w)\n", U, NT, g, us, bytes / us / 1e3); }
    RUNC(1, false) RUNC(4, false) RUNC(4, true)
    return 0;
}
