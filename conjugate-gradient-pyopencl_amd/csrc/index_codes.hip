// Column indices of a CSR matrix re-encoded for the single-RHS row-block SpMV (exact: the kernel rebuilds the very same column).
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"
#include "spmv_device.h"
#include "reduce_device.h"
#include "launch_util.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

namespace cgamd {

// ---- one-byte column codes (SpmvPlan::codes) ---------------------------------------------------------------------------
constexpr int kDictSlots = 1024;            // open-addressing table of the distinct offsets (<= 256 accepted)
constexpr int kDictEmpty = -2147483647 - 1;
CG_DEV unsigned dict_hash(int d) { return ((unsigned)d * 2654435761u) >> 22; }   // 10 bits
// one lane per row: every (column - row) offset goes into the table; count[0] = distinct offsets so far
__global__ __launch_bounds__(256) void index_offsets_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                            int *table, int *count) {
    for (long long row = blockIdx.x * 256LL + threadIdx.x; row < n; row += 256LL * gridDim.x) {
        if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 256) return;     // not codable: stop early
        int last = kDictEmpty;
        for (int j = ptr[row], e = ptr[row + 1]; j < e; ++j) {
            const int d = cols[j] - (int)row;
            if (d == last) continue;
            last = d;
            unsigned h = dict_hash(d);
            int probes = 0;
            for (; probes < kDictSlots; ++probes, h = (h + 1) & (kDictSlots - 1)) {
                int v = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == kDictEmpty) {
                    v = atomicCAS(table + h, kDictEmpty, d);
                    if (v == kDictEmpty) { atomicAdd(count, 1); break; }
                }
                if (v == d) break;
            }
            if (probes == kDictSlots) { atomicAdd(count, kDictSlots); return; }       // table full
        }
    }
}
// table slot -> code (position of the offset in the sorted dictionary); one lane per row writes its codes
__global__ __launch_bounds__(256) void index_encode_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                           const int *__restrict__ table, const unsigned char *__restrict__ slot_code,
                                                           unsigned char *__restrict__ codes) {
    __shared__ int stab[kDictSlots];
    __shared__ unsigned char scode[kDictSlots];
    for (int i = threadIdx.x; i < kDictSlots; i += 256) { stab[i] = table[i]; scode[i] = slot_code[i]; }
    __syncthreads();
    for (long long row = blockIdx.x * 256LL + threadIdx.x; row < n; row += 256LL * gridDim.x) {
        for (int j = ptr[row], e = ptr[row + 1]; j < e; ++j) {
            const int d = cols[j] - (int)row;
            unsigned h = dict_hash(d);
            while (stab[h] != d) h = (h + 1) & (kDictSlots - 1);      // every offset is in the table
            codes[j] = scode[h];
        }
    }
}

int build_index_codes(int n, long long nnz, const int *ptr_dev, const int *cols_dev, hipStream_t st, unsigned char **codes_out,
                      int **dict_out, int *distinct_out) {
    *codes_out = nullptr;
    *dict_out = nullptr;
    *distinct_out = 0;
    if (n <= 0 || nnz <= 0) return CGAMD_OK;
    int *work = nullptr;        // [1024 table | 1 count | 256 dict]
    CG_HIP(hipMalloc((void **)&work, (kDictSlots + 64 + 256) * sizeof(int) + kDictSlots));
    std::vector<int> table(kDictSlots + 1, kDictEmpty);
    table[kDictSlots] = 0;
    hipError_t e = hipMemcpyAsync(work, table.data(), (kDictSlots + 1) * sizeof(int), hipMemcpyHostToDevice, st);
    int g = (int)std::min<long long>((n + 255LL) / 256, 8192);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(index_offsets_kernel, dim3(g), dim3(256), 0, st, n, ptr_dev, cols_dev, work, work + kDictSlots);
        e = hipMemcpyAsync(table.data(), work, (kDictSlots + 1) * sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(work); return fail(CGAMD_ERR_HIP, std::string("build_index_codes: ") + hipGetErrorString(e)); }
    const int distinct = table[kDictSlots];
    if (distinct < 1 || distinct > 256) { (void)hipFree(work); return CGAMD_OK; }
    std::vector<int> dict;
    for (int i = 0; i < kDictSlots; ++i)
        if (table[i] != kDictEmpty) dict.push_back(table[i]);
    std::sort(dict.begin(), dict.end());
    std::vector<unsigned char> slot_code(kDictSlots, 0);
    for (int i = 0; i < kDictSlots; ++i)
        if (table[i] != kDictEmpty) slot_code[i] = (unsigned char)(std::lower_bound(dict.begin(), dict.end(), table[i]) - dict.begin());
    dict.resize(256, dict[0]);
    unsigned char *codes = nullptr;
    int *dict_dev = nullptr;
    unsigned char *slot_dev = reinterpret_cast<unsigned char *>(work + kDictSlots + 64 + 256);
    e = hipMalloc((void **)&codes, (size_t)nnz + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&dict_dev, 256 * sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(codes + nnz, 0, 64, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dict_dev, dict.data(), 256 * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(slot_dev, slot_code.data(), kDictSlots, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(index_encode_kernel, dim3(g), dim3(256), 0, st, n, ptr_dev, cols_dev, work, slot_dev, codes);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(work);
    if (e != hipSuccess) {
        if (codes) (void)hipFree(codes);
        if (dict_dev) (void)hipFree(dict_dev);
        return fail(CGAMD_ERR_HIP, std::string("build_index_codes: ") + hipGetErrorString(e));
    }
    *codes_out = codes;
    *dict_out = dict_dev;
    *distinct_out = distinct;
    return CGAMD_OK;
}

// ---- 16-bit block-relative columns ---------------------------------------------------------------------------------------
// For matrices with more than 256 distinct (column - row) offsets -- unstructured meshes, banded random patterns, everything a
// Matrix-Market file may hold (reference main.c:20-33) -- whose 256-row blocks still span fewer than 65 536 columns each (any
// matrix with a bandwidth below ~32k: a reasonable ordering of a mesh): codes16[j] = aCols[j] - base[rb(j)], base[rb] = the
// smallest column of row block rb.  2 instead of 4 index bytes per non-zero; exact like the one-byte codes.
__global__ __launch_bounds__(256) void rowblock_colrange_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols, int row_blocks,
                                                                int *__restrict__ base, int *flag) {
    __shared__ int smin, smax;
    const int rb = blockIdx.x;
    if (threadIdx.x == 0) { smin = 0x7fffffff; smax = -1; }
    __syncthreads();
    const int p0 = ptr[rb * 256], p1 = ptr[min(rb * 256 + 256, n)];
    int cmin = 0x7fffffff, cmax = -1;
    for (int j = p0 + (int)threadIdx.x; j < p1; j += 256) { cmin = min(cmin, cols[j]); cmax = max(cmax, cols[j]); }
    atomicMin(&smin, cmin);
    atomicMax(&smax, cmax);
    __syncthreads();
    if (threadIdx.x == 0) {
        base[rb] = smax >= 0 ? smin : 0;
        if (smax >= 0 && smax - smin > 65535) atomicOr(flag, 1);
    }
}
__global__ __launch_bounds__(256) void index_encode16_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ cols,
                                                             const int *__restrict__ base, unsigned short *__restrict__ codes) {
    const int rb = blockIdx.x;
    const int p0 = ptr[rb * 256], p1 = ptr[min(rb * 256 + 256, n)], b = base[rb];
    for (int j = p0 + (int)threadIdx.x; j < p1; j += 256) codes[j] = (unsigned short)(cols[j] - b);
}

// *codes_out (2 nnz + 64 bytes) and *base_out (row_blocks ints) are device allocations the caller frees; both null when some row
// block spans 65 536 columns or more.  Synchronises `st`.
int build_index_codes16(int n, long long nnz, const int *ptr_dev, const int *cols_dev, hipStream_t st, unsigned char **codes_out, int **base_out) {
    *codes_out = nullptr;
    *base_out = nullptr;
    if (n <= 0 || nnz <= 0) return CGAMD_OK;
    const int row_blocks = (n + 255) / 256;
    int *base = nullptr, *flag = nullptr;
    CG_HIP(hipMalloc((void **)&base, sizeof(int) * (size_t)row_blocks + 16));
    flag = base + row_blocks;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
    int bad = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rowblock_colrange_kernel, dim3(row_blocks), dim3(256), 0, st, n, ptr_dev, cols_dev, row_blocks, base, flag);
        e = hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess || bad) {
        (void)hipFree(base);
        return e == hipSuccess ? CGAMD_OK : fail(CGAMD_ERR_HIP, std::string("build_index_codes16: ") + hipGetErrorString(e));
    }
    unsigned char *codes = nullptr;
    e = hipMalloc((void **)&codes, 2 * (size_t)nnz + 64);
    if (e == hipSuccess) e = hipMemsetAsync(codes + 2 * (size_t)nnz, 0, 64, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(index_encode16_kernel, dim3(row_blocks), dim3(256), 0, st, n, ptr_dev, cols_dev, base, reinterpret_cast<unsigned short *>(codes));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        if (codes) (void)hipFree(codes);
        (void)hipFree(base);
        return fail(CGAMD_ERR_HIP, std::string("build_index_codes16: ") + hipGetErrorString(e));
    }
    *codes_out = codes;
    *base_out = base;
    return CGAMD_OK;
}

}  // namespace cgamd
