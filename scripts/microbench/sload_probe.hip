// How can waves of one XCD watch each other's progress cheaply?  (rowmajor.hip pace words)
// Producer waves (blocks 8, 16, ..: the same XCD as block 0 under round-robin dispatch) publish a step number into their own
// BYTE of one 256-byte array, `steps` times; a consumer (block 0 = same XCD, block 1 = another XCD) polls until all producer
// bytes reach `steps`.  Store flavours: 0 plain, 1 sc1 (agent scope, write-through).  Load flavours: 0 s_load glc,
// 1 s_dcache_inv + s_load, 2 vector sc1, 3 vector sc0, 4 vector plain.
// Prints: producer clocks per publish (stores serialise at the memory side?), consumer polls / clocks per poll / seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ inline int sload_glc(const int *p) { int v; asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory"); return v; }
__device__ inline int sload_inv(const int *p) { int v; asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory"); return v; }
__device__ inline int vload(const int *p, int f) {
    int v;
    if (f == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (f == 3) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ void probe(unsigned char *bytes, long long *out, int sf, int lf, int steps, int nprod) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= 8 && (b & 7) == 0) {       // producers: block 8 k -> byte k - 1
        const int k = b / 8 - 1;
        if (k >= nprod) return;
        const long long t0 = clock64();
        for (int i = 1; i <= steps; ++i) {
            if (lane == 0) {
                if (sf) asm volatile("global_store_byte %0, %1, off sc1" ::"v"(bytes + k), "v"(i) : "memory");
                else asm volatile("global_store_byte %0, %1, off" ::"v"(bytes + k), "v"(i) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_sleep(16);
        }
        if (lane == 0 && k == 0) out[40] = (clock64() - t0) / steps;
        return;
    }
    if (b > 1) return;
    int polls = 0, seen = 0;
    const long long t0 = clock64();
    const int *w = reinterpret_cast<const int *>(bytes);
    while (polls < 100000) {
        // lane l looks at word l (bytes 4l..4l+3)
        int v;
        if (lf == 0) v = sload_glc(w + (nprod - 1) / 4);
        else if (lf == 1) v = sload_inv(w + (nprod - 1) / 4);
        else v = vload(w + (lane < 64 ? lane : 0), lf);
        ++polls;
        bool ok;
        if (lf <= 1) { seen = (v >> (8 * ((nprod - 1) & 3))) & 0xff; ok = seen >= steps; }     // scalar: only the last producer's byte
        else {
            bool mine = true;
            for (int q = 0; q < 4; ++q) if (4 * lane + q < nprod) mine = mine && ((v >> (8 * q)) & 0xff) >= steps;
            ok = __builtin_amdgcn_ballot_w64(!mine) == 0;
            seen = __builtin_amdgcn_readfirstlane(v) & 0xff;
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(4);
    }
    const long long t1 = clock64();
    if (lane == 0) { out[3 * b] = polls; out[3 * b + 1] = (t1 - t0) / (polls ? polls : 1); out[3 * b + 2] = seen; }
}
int main() {
    unsigned char *bytes; long long *out;
    hipMalloc(&bytes, 4096); hipMalloc(&out, 64 * sizeof(long long));
    const int nprod = 200, steps = 100;
    for (int sf = 0; sf < 2; ++sf)
        for (int lf = 0; lf < 5; ++lf) {
            hipMemset(bytes, 0, 4096); hipMemset(out, 0, 64 * sizeof(long long));
            hipLaunchKernelGGL(probe, dim3(8 * (nprod + 1)), dim3(64), 0, 0, bytes, out, sf, lf, steps, nprod);
            hipDeviceSynchronize();
            std::vector<long long> h(64);
            hipMemcpy(h.data(), out, 64 * sizeof(long long), hipMemcpyDeviceToHost);
            printf("store %s load %d: producer clk/publish %lld | same-XCD consumer polls=%lld clk/poll=%lld seen=%lld | other-XCD polls=%lld clk/poll=%lld seen=%lld\n",
                   sf ? "sc1  " : "plain", lf, h[40], h[0], h[1], h[2], h[3], h[4], h[5]);
        }
    return 0;
}
