// wave_sum4 / wave_sum2 (device_types.h) against wave_sum, bit for bit, on random doubles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../conjugate-gradient-pyopencl_amd/csrc/device_types.h"
using namespace cgamd;
__global__ void probe(const double *in, double *out) {
    const int l = threadIdx.x;
    double v[4];
    for (int i = 0; i < 4; ++i) v[i] = in[i * 64 + l];
    double ref[4];
    for (int i = 0; i < 4; ++i) ref[i] = wave_sum(v[i]);
    const double g4 = wave_sum4(v[0], v[1], v[2], v[3]);
    const double g2 = wave_sum2(v[0], v[1]);
    if (l == 0) { for (int i = 0; i < 4; ++i) out[i] = ref[i]; }
    if ((l & 15) == 0) out[4 + (l == 0 ? 0 : l == 16 ? 2 : l == 32 ? 1 : 3)] = g4;
    if ((l & 31) == 0) out[8 + (l >> 5)] = g2;
}
int main() {
    double h[256], o[16];
    srand(7);
    for (int i = 0; i < 256; ++i) h[i] = (double)rand() / RAND_MAX - 0.5 + 1e-9 * rand();
    double *d, *r;
    hipMalloc(&d, sizeof(h)); hipMalloc(&r, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, r);
    hipMemcpy(o, r, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4; ++i) { if (memcmp(&o[i], &o[4 + i], 8)) { ++bad; printf("wave_sum4 value %d: %.17g vs %.17g\n", i, o[4 + i], o[i]); } }
    for (int i = 0; i < 2; ++i) { if (memcmp(&o[i], &o[8 + i], 8)) { ++bad; printf("wave_sum2 value %d: %.17g vs %.17g\n", i, o[8 + i], o[i]); } }
    printf(bad ? "MISMATCH\n" : "wave_sum4 / wave_sum2 == wave_sum bit for bit\n");
    return bad != 0;
}
