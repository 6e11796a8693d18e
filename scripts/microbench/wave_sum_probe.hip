#include <hip/hip_runtime.h>
__device__ __forceinline__ unsigned dpp_shl(unsigned v, int) { return v; }
template <int N> __device__ __forceinline__ double row_shl(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, 0x100 + N, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), 0x100 + N, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double from_upper32(double v) {
    const unsigned long long b = __double_as_longlong(v);
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ double from_row16(double v) {
    const unsigned long long b = __double_as_longlong(v);
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ double wave_sum_fast(double v) {
    v += from_upper32(v);
    v += from_row16(v);
    v += row_shl<8>(v);
    v += row_shl<4>(v);
    v += row_shl<2>(v);
    v += row_shl<1>(v);
    return v;
}
__device__ double wave_sum_ref(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__global__ void k(const double *in, double *out) {
    const double v = in[threadIdx.x];
    const double a = wave_sum_fast(v), b = wave_sum_ref(v);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}
int main() {
    double h[64], *d, *o, r[2];
    unsigned long long s = 88172645463325252ull;
    int bad = 0;
    hipMalloc(&d, 512); hipMalloc(&o, 16);
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 64; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(long long)s * 1e-19 * ((i * 7 + rep) % 5 + 0.1); }
        hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, o);
        hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        if (r[0] != r[1]) ++bad;
    }
    printf("mismatches %d of 200 (last %.17g %.17g)\n", bad, r[0], r[1]);
    return bad != 0;
}
