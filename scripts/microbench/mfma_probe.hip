// Layout / timing probe for the small-block matrix-core forms on gfx950 (not product code):
//   v_mfma_f64_4x4x4_4b_f64 : which lane holds A[block][i][k], B[block][k][j], D[block][i][j]
// One-hot A at lane la and one-hot B at lane lb; the D lane that becomes 1 tells (i from la, j from lb) and that
// la, lb share block and k.  Prints the decoded maps, then times back-to-back issue of the 16x16x4 and 4x4x4 f64 forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void probe44(int *hit) {   // hit[la*64+lb] = lane whose D became nonzero, or -1
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) hit[la * 64 + lb] = m ? __ffsll((long long)m) - 1 : -1;
        }
}
template <int FORM> __global__ void rate(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    if (FORM == 0) {
        f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 64 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w;
    } else {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3;
    }
}
int main() {
    int *hit; hipMalloc(&hit, 4096 * sizeof(int));
    probe44<<<1, 64>>>(hit);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), hit, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // for every A lane: the set of B lanes that pair with it and the D lanes produced
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with B lanes -> D lane:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" %d->%d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    double *out; hipMalloc(&out, 1024 * 64 * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (form == 0) rate<0><<<1024, 64>>>(out, iters); else rate<1><<<1024, 64>>>(out, iters);   // 1 wave per SIMD
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 1024 waves over 1024 SIMDs: each SIMD issues 4*iters MFMAs
        printf("form %s: %.3f ms for %d MFMAs per SIMD -> %.1f ns per MFMA (x2.4 GHz = %.1f cycles)\n", form ? "f64 4x4x4_4b" : "f64 16x16x4",
               ms, 4 * iters, ms * 1e6 / (4.0 * iters), ms * 1e6 / (4.0 * iters) * 2.4);
    }
    return 0;
}
