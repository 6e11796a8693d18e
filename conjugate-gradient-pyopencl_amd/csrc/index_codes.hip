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

// -------------------------------------------------------------------------------------------------
// VALUE codes: matrices with at most 256 distinct entries (constant-coefficient stencils, graph Laplacians, lattice operators: the
// headline system has 2) keep `vcodes[nnz]` (one byte each) and `vdict[256]` with aValues[j] == vdict[vcodes[j]] -- the same bits, so
// every product, sum and history is bit-identical to the kernels that read aValues.  With the column codes the SpMV then streams 2
// bytes per non-zero instead of sizeof(T) + 1 (fp64: 9).  Built with an open-addressing table of the value bits (4096 slots): a pass
// that inserts (it stops as soon as a 257th value shows up), a one-work-group pass that numbers the occupied slots, a pass that
// encodes.  Which code a value gets may differ from run to run (insertion races); the value it stands for does not.
// -------------------------------------------------------------------------------------------------
constexpr int kVSlots = 4096;
constexpr unsigned long long kVEmpty = ~0ULL;      // (a NaN pattern: a matrix that holds it is not coded)
template <typename T> CG_DEV unsigned long long value_key(const T &v) {
    if constexpr (sizeof(T) == 4) { unsigned b; __builtin_memcpy(&b, &v, 4); return b; }
    else { unsigned long long b; __builtin_memcpy(&b, &v, 8); return b; }
}
CG_DEV unsigned vhash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 29;
    return (unsigned)k & (kVSlots - 1);
}
// state: [0] number of distinct values, [1] failure flag
template <typename T>
__global__ __launch_bounds__(256) void value_insert_kernel(long long nnz, const T *__restrict__ vals, unsigned long long *table, int *state) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nnz; j += stride) {
        if (__hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const unsigned long long key = value_key(vals[j]);
        if (key == kVEmpty) { atomicExch(state + 1, 1); return; }
        unsigned s = vhash(key);
        for (int probe = 0; probe < kVSlots; ++probe, s = (s + 1) & (kVSlots - 1)) {
            unsigned long long cur = __hip_atomic_load(table + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == key) break;
            if (cur == kVEmpty) {
                cur = atomicCAS(table + s, kVEmpty, key);
                if (cur == kVEmpty) {                       // a new value
                    if (atomicAdd(state, 1) >= 256) atomicExch(state + 1, 1);
                    break;
                }
                if (cur == key) break;
            }
        }
    }
}
// one work-group: occupied slot -> code (slot order), dictionary entries
template <typename T>
__global__ __launch_bounds__(1024) void value_number_kernel(const unsigned long long *table, unsigned short *slot_code, T *vdict) {
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    int mine = 0;
    for (int k = 0; k < kVSlots / 1024; ++k) mine += table[t * (kVSlots / 1024) + k] != kVEmpty;
    int incl = mine;                                        // inclusive scan over the wave, then over the waves
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if ((t & 63) >= off) incl += o; }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
    int code = base + incl - mine;
    for (int k = 0; k < kVSlots / 1024; ++k) {
        const int s = t * (kVSlots / 1024) + k;
        const unsigned long long key = table[s];
        if (key != kVEmpty) {
            slot_code[s] = (unsigned short)code;
            if (code < 256) {
                T v;
                if constexpr (sizeof(T) == 4) { const unsigned b = (unsigned)key; __builtin_memcpy(&v, &b, 4); }
                else __builtin_memcpy(&v, &key, 8);
                vdict[code] = v;
            }
            ++code;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void value_encode_kernel(long long nnz, const T *__restrict__ vals, const unsigned long long *__restrict__ table,
                                                           const unsigned short *__restrict__ slot_code, unsigned char *__restrict__ vcodes) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nnz; j += stride) {
        const unsigned long long key = value_key(vals[j]);
        unsigned s = vhash(key);
        while (table[s] != key) s = (s + 1) & (kVSlots - 1);     // present: the insert pass put it there
        vcodes[j] = (unsigned char)slot_code[s];
    }
}

template <typename T>
static int build_value_codes_impl(long long nnz, const void *vals_dev, hipStream_t st, unsigned char **vcodes_out, void **vdict_out, int *n_values) {
    // scratch: table | slot codes | state
    char *scratch = nullptr;
    const size_t tb = sizeof(unsigned long long) * kVSlots, sb = sizeof(unsigned short) * kVSlots;
    CG_HIP(hipMalloc((void **)&scratch, tb + sb + 16));
    auto *table = reinterpret_cast<unsigned long long *>(scratch);
    auto *slot_code = reinterpret_cast<unsigned short *>(scratch + tb);
    int *state = reinterpret_cast<int *>(scratch + tb + sb);
    int h[2] = {0, 1};
    hipError_t e = hipMemsetAsync(table, 0xff, tb, st);
    if (e == hipSuccess) e = hipMemsetAsync(state, 0, 8, st);
    const int grid = (int)std::min<long long>((nnz + 255) / 256, 4096);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((value_insert_kernel<T>), dim3(grid), dim3(256), 0, st, nnz, static_cast<const T *>(vals_dev), table, state);
        e = hipMemcpyAsync(h, state, 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess || h[1] != 0 || h[0] < 1 || h[0] > 256) {
        (void)hipFree(scratch);
        return e == hipSuccess ? CGAMD_OK : fail(CGAMD_ERR_HIP, std::string("build_value_codes: ") + hipGetErrorString(e));
    }
    unsigned char *vcodes = nullptr;
    void *vdict = nullptr;
    e = hipMalloc((void **)&vcodes, (size_t)nnz + 64);
    if (e == hipSuccess) e = hipMalloc(&vdict, 256 * sizeof(T));
    if (e == hipSuccess) e = hipMemsetAsync(vcodes + (size_t)nnz, 0, 64, st);
    if (e == hipSuccess) e = hipMemsetAsync(vdict, 0, 256 * sizeof(T), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((value_number_kernel<T>), dim3(1), dim3(1024), 0, st, table, slot_code, static_cast<T *>(vdict));
        hipLaunchKernelGGL((value_encode_kernel<T>), dim3(grid), dim3(256), 0, st, nnz, static_cast<const T *>(vals_dev), table, slot_code, vcodes);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(scratch);
    if (e != hipSuccess) {
        if (vcodes) (void)hipFree(vcodes);
        if (vdict) (void)hipFree(vdict);
        return fail(CGAMD_ERR_HIP, std::string("build_value_codes: ") + hipGetErrorString(e));
    }
    *vcodes_out = vcodes; *vdict_out = vdict; *n_values = h[0];
    return CGAMD_OK;
}
// *vcodes_out (nnz + 64 bytes) and *vdict_out (256 values) are device allocations the caller frees; both null when the matrix has more
// than 256 distinct values (or is complex128).  Synchronises `st`.
int build_value_codes(int dtype, long long nnz, const void *vals_dev, hipStream_t st, unsigned char **vcodes_out, void **vdict_out, int *n_values) {
    *vcodes_out = nullptr; *vdict_out = nullptr; *n_values = 0;
    if (nnz <= 0 || dtype == CGAMD_C128) return CGAMD_OK;
    if (dtype == CGAMD_F32) return build_value_codes_impl<float>(nnz, vals_dev, st, vcodes_out, vdict_out, n_values);
    if (dtype == CGAMD_F64) return build_value_codes_impl<double>(nnz, vals_dev, st, vcodes_out, vdict_out, n_values);
    return build_value_codes_impl<float2>(nnz, vals_dev, st, vcodes_out, vdict_out, n_values);
}

// -------------------------------------------------------------------------------------------------
// JOINT codes: where a matrix has both one-byte column codes and one-byte value codes and at most 256 distinct (offset, value)
// PAIRS (a constant-coefficient stencil has as many pairs as offsets: 7), one byte per non-zero names the pair: jdict_off[code] /
// jdict_val[code].  The SpMV then streams 1 byte per non-zero and does one look-up per entry instead of two.  Built from the two
// code arrays: mark the pairs that occur (65 536 possible), number them in order, encode.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_mark_kernel(long long nnz, const unsigned char *__restrict__ codes, const unsigned char *__restrict__ vcodes,
                                                        unsigned char *present) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nnz; j += stride) {
        const unsigned k = ((unsigned)codes[j] << 8) | vcodes[j];
        if (!present[k]) present[k] = 1;            // (racing writers store the same byte)
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void pair_number_kernel(const unsigned char *present, const int *__restrict__ dict, const T *__restrict__ vdict,
                                                           unsigned short *pair_code, int *joff, T *jval, int *count) {
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    int mine = 0;
    for (int k = 0; k < 64; ++k) mine += present[t * 64 + k] != 0;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if ((t & 63) >= off) incl += o; }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
    int code = base + incl - mine;
    for (int k = 0; k < 64; ++k) {
        const int p = t * 64 + k;
        if (present[p]) {
            pair_code[p] = (unsigned short)code;
            if (code < 256) { joff[code] = dict[p >> 8]; jval[code] = vdict[p & 255]; }
            ++code;
        }
    }
    if (t == 1023) *count = code;
}
__global__ __launch_bounds__(256) void pair_encode_kernel(long long nnz, const unsigned char *__restrict__ codes, const unsigned char *__restrict__ vcodes,
                                                          const unsigned short *__restrict__ pair_code, unsigned char *__restrict__ jcodes) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nnz; j += stride)
        jcodes[j] = (unsigned char)pair_code[((unsigned)codes[j] << 8) | vcodes[j]];
}
template <typename T>
static int build_joint_codes_impl(long long nnz, const unsigned char *codes, const unsigned char *vcodes, const int *dict, const void *vdict,
                                  hipStream_t st, unsigned char **jcodes_out, int **joff_out, void **jval_out, int *n_pairs) {
    char *scratch = nullptr;
    CG_HIP(hipMalloc((void **)&scratch, 65536 + 65536 * 2 + 16));
    unsigned char *present = reinterpret_cast<unsigned char *>(scratch);
    unsigned short *pair_code = reinterpret_cast<unsigned short *>(scratch + 65536);
    int *count = reinterpret_cast<int *>(scratch + 65536 * 3);
    unsigned char *jcodes = nullptr;
    int *joff = nullptr;
    void *jval = nullptr;
    int h = 0;
    hipError_t e = hipMemsetAsync(scratch, 0, 65536 * 3 + 16, st);
    if (e == hipSuccess) e = hipMalloc((void **)&joff, 256 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&jval, 256 * sizeof(T));
    if (e == hipSuccess) e = hipMemsetAsync(joff, 0, 256 * sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(jval, 0, 256 * sizeof(T), st);
    const int grid = (int)std::min<long long>((nnz + 255) / 256, 4096);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pair_mark_kernel, dim3(grid), dim3(256), 0, st, nnz, codes, vcodes, present);
        hipLaunchKernelGGL((pair_number_kernel<T>), dim3(1), dim3(1024), 0, st, present, dict, static_cast<const T *>(vdict), pair_code, joff, static_cast<T *>(jval), count);
        e = hipMemcpyAsync(&h, count, sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && h >= 1 && h <= 256) {
        e = hipMalloc((void **)&jcodes, (size_t)nnz + 64);
        if (e == hipSuccess) e = hipMemsetAsync(jcodes + (size_t)nnz, 0, 64, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pair_encode_kernel, dim3(grid), dim3(256), 0, st, nnz, codes, vcodes, pair_code, jcodes);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(scratch);
    if (e != hipSuccess || !jcodes) {
        if (jcodes) (void)hipFree(jcodes);
        if (joff) (void)hipFree(joff);
        if (jval) (void)hipFree(jval);
        return e == hipSuccess ? CGAMD_OK : fail(CGAMD_ERR_HIP, std::string("build_joint_codes: ") + hipGetErrorString(e));
    }
    *jcodes_out = jcodes; *joff_out = joff; *jval_out = jval; *n_pairs = h;
    return CGAMD_OK;
}
// all three outputs null when the matrix has more than 256 distinct (offset, value) pairs.  Synchronises `st`.
int build_joint_codes(int dtype, long long nnz, const unsigned char *codes, const unsigned char *vcodes, const int *dict, const void *vdict,
                      hipStream_t st, unsigned char **jcodes_out, int **joff_out, void **jval_out, int *n_pairs) {
    *jcodes_out = nullptr; *joff_out = nullptr; *jval_out = nullptr; *n_pairs = 0;
    if (nnz <= 0 || !codes || !vcodes || dtype == CGAMD_C128) return CGAMD_OK;
    if (dtype == CGAMD_F32) return build_joint_codes_impl<float>(nnz, codes, vcodes, dict, vdict, st, jcodes_out, joff_out, jval_out, n_pairs);
    if (dtype == CGAMD_F64) return build_joint_codes_impl<double>(nnz, codes, vcodes, dict, vdict, st, jcodes_out, joff_out, jval_out, n_pairs);
    return build_joint_codes_impl<float2>(nnz, codes, vcodes, dict, vdict, st, jcodes_out, joff_out, jval_out, n_pairs);
}

}  // namespace cgamd
