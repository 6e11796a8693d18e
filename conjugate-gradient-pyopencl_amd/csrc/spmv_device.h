// Device-side pieces shared by the SpMV kernel files (spmv.hip, p2p.hip, cg1.hip): kernel arguments, the row-block -> XCD
// schedule, and the staging of a row block's matrix slice in LDS.
#pragma once
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"

namespace cgamd {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// =================================================================================================
// SpMV / SpMM, CSR-stream: a work-group owns BLOCK consecutive rows at a time.  Their non-zeros are one
// contiguous slice of aValues/aCols, streamed with 16 B coalesced non-temporal loads (4 nnz per lane
// per load group, slice start rounded down to a multiple of 4 so every load is aligned); each lane
// multiplies its non-zeros with the gathered x entries and parks the products in LDS; after one
// barrier, lane t sums the products of row t (left to right, CSR order) and writes y[t] coalesced.
// For nRHS > 1 the matrix quads stay in registers while the gather/LDS/sum phase repeats per RHS,
// so the matrix is read from HBM once per SpMM.
// =================================================================================================
template <typename T> struct SpmvArgs {
    int n;
    int nrhs;
    long long nnz;
    const T *vals;
    const int *ptr;
    const int *cols;
    const T *x;
    long long ldx;
    T *y;
    long long ldy;
    const T *dvec;                  // fused dot: sum dvec[row] * y[row]
    typename VT<T>::acc *partials;  // [nrhs][grid]
    int row_blocks;
    const int *rb_list;             // row-block kernel: optional explicit list of row blocks (multi-GPU interior / boundary split)
    int rb_count;
    int cap;   // row-block kernel: LDS slice capacity in entries (multiple of 4)
    int cycle; // row-block kernel: block-cyclic schedule over the XCDs, cycle length in row blocks (1 = one contiguous eighth per XCD)
    const unsigned char *codes;   // coded row-block kernel: one byte per non-zero, aCols[j] = row + dict[codes[j]] (build_index_codes)
    const int *dict;              // [256]
    const unsigned char *vcodes;  // value-coded row-block kernel: one byte per non-zero, aValues[j] == vdict[vcodes[j]] (build_value_codes)
    const T *vdict;               // [256]
};

// Row-block schedule shared by the row-block kernels: work-group b runs on XCD b%8 as that XCD's (b/8)-th block.
// cycle > 1: block-cyclic -- cycles of `cycle` row blocks, XCD j takes the j-th run of ceil(cycle/8) blocks of each;
// cycle <= 1: XCD j owns the j-th contiguous eighth.  Returns -1 for the padding work-groups of the grid.
__host__ __device__ __forceinline__ int rowblock_of(int b, int row_blocks, int cycle) {
    const int xcd = b & 7, i = b >> 3;
    if (cycle > 1) {
        const int chunk = (cycle + 7) >> 3;
        const int k = i / chunk, lo = xcd * chunk + (i - k * chunk);
        const int rb = k * cycle + lo;
        return (lo < cycle && rb < row_blocks) ? rb : -1;
    }
    const int xb = (int)((long long)xcd * row_blocks / 8), xe = (int)((long long)(xcd + 1) * row_blocks / 8);
    return i < xe - xb ? xb + i : -1;
}
inline int rowblock_grid(int row_blocks, int cycle) {
    if (cycle > 1) return 8 * ((cycle + 7) / 8) * ((row_blocks + cycle - 1) / cycle);
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) {
        const int m = (int)((long long)(x + 1) * row_blocks / 8) - (int)((long long)x * row_blocks / 8);
        per_xcd = m > per_xcd ? m : per_xcd;
    }
    return per_xcd * 8;
}

// Park the slice [cfirst, p1) of aValues/aCols in LDS with the VALUE stream interleaved across the lanes in 16-byte chunks:
// chunk k of a lane is chunk k*64 + lane of
// its wave's 256-entry span, so every load instruction of a wave covers one contiguous 1 KB (a quad-per-lane mapping gives
// a lane 4 consecutive entries = 16/32/64 contiguous bytes, and each of its 1/2/4 load instructions touches every cache
// line of the span partially).  With non-temporal loads the partially used lines of complex128 were fetched again by the
// later instructions: 435 -> 313 us for the N=10M SpMV.  Columns (4 B) keep the quad mapping: one 16-byte load per lane.
template <typename T> struct Chunk16;
template <> struct Chunk16<float> { using V = f32x4; };
template <> struct Chunk16<double> { using V = f64x2; };
template <> struct Chunk16<float2> { using V = f32x4; };
template <> struct Chunk16<double2> { using V = f64x2; };
// CODED: 0 = column indices (4 bytes), 1 = one-byte codes into the matrix's offset dictionary, 2 = 16-bit columns relative to the
// row block's first column (index_codes.hip); the code arrays are padded, so they need no tail handling
template <typename T, int BLOCK, bool NT, bool FULL, int CODED = 0>
CG_DEV void stage_slice_ilv(const T *__restrict__ vals, const int *__restrict__ cols, long long nnz, int cfirst, int p1,
                            T *sv, int *sc, const unsigned char *__restrict__ codes = nullptr) {
    using V = typename Chunk16<T>::V;
    constexpr int EPC = 16 / (int)sizeof(T);   // values per chunk
    constexpr int NV = 4 / EPC;                // chunks per lane and quad region
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    for (long long base = cfirst; base < p1; base += 8 * BLOCK) {
        V ch[2][NV];
        i32x4 cc[2];
        unsigned cw[2];
        u32x2 cw2[2];
        long long ev[2][NV], qc[2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const long long rbase = base + (long long)rg * 4 * BLOCK;
            qc[rg] = rbase + 4 * t;
            if constexpr (CODED == 1) {      // four one-byte codes per lane
                if (qc[rg] < p1) {
                    const unsigned *cp = reinterpret_cast<const unsigned *>(codes + qc[rg]);
                    cw[rg] = NT ? __builtin_nontemporal_load(cp) : *cp;
                }
            } else if constexpr (CODED == 2) {      // four 16-bit codes per lane
                if (qc[rg] < p1) {
                    const u32x2 *cp = reinterpret_cast<const u32x2 *>(codes + 2 * qc[rg]);
                    cw2[rg] = NT ? __builtin_nontemporal_load(cp) : *cp;
                }
            } else if (qc[rg] < p1 && (FULL || qc[rg] + 4 <= nnz)) cc[rg] = ld16<i32x4, NT>(cols + qc[rg]);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                ev[rg][k] = rbase + (long long)wave * 4 * kWave + (long long)(k * kWave + lane) * EPC;
                if (ev[rg][k] < p1 && (FULL || ev[rg][k] + EPC <= nnz)) ch[rg][k] = ld16<V, NT>(vals + ev[rg][k]);
            }
        }
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            if (qc[rg] < p1) {
                const int o = (int)(qc[rg] - cfirst);
                if constexpr (CODED == 1) {
                    *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(sc) + o) = cw[rg];
                } else if constexpr (CODED == 2) {
                    *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned char *>(sc) + 2 * o) = cw2[rg];
                } else if (FULL || qc[rg] + 4 <= nnz) {
                    *reinterpret_cast<i32x4 *>(sc + o) = cc[rg];
                } else {
                    for (int j = 0; j < 4; ++j) sc[o + j] = qc[rg] + j < nnz ? cols[qc[rg] + j] : 0;
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (ev[rg][k] < p1) {
                    const int o = (int)(ev[rg][k] - cfirst);
                    if (FULL || ev[rg][k] + EPC <= nnz) {
                        *reinterpret_cast<V *>(sv + o) = ch[rg][k];
                    } else {
                        for (int j = 0; j < EPC; ++j) sv[o + j] = ev[rg][k] + j < nnz ? vals[ev[rg][k] + j] : vzero<T>();
                    }
                }
            }
        }
    }
}

// POL: -2 = column indices, -3 = one-byte column codes, -4 = 16-bit block-relative columns.  Only the work-group that owns the very end
// of the matrix can meet a partial quad: block-uniform branch, so the common path carries no per-lane tail handling (whose control flow
// made hipcc serialise the loads).
template <typename T, int BLOCK, bool NT, int POL = -2>
CG_DEV void stage_slice(const T *__restrict__ vals, const int *__restrict__ cols, long long nnz, int cfirst, int p1, T *sv,
                        int *sc, const unsigned char *__restrict__ codes = nullptr) {
    constexpr int C = POL == -3 ? 1 : POL == -4 ? 2 : 0;
    if (((long long)(p1 + 3) & ~3LL) <= nnz) stage_slice_ilv<T, BLOCK, NT, true, C>(vals, cols, nnz, cfirst, p1, sv, sc, codes);
    else stage_slice_ilv<T, BLOCK, NT, false, C>(vals, cols, nnz, cfirst, p1, sv, sc, codes);
}

// Both code streams of the fully coded row-block kernel (column codes + value codes, one byte each per non-zero) for the slice
// [cfirst, p1): a dword of each per lane and round of 4 BLOCK entries, two rounds in flight.  The arrays are padded by 64 bytes:
// no tail handling.
template <int BLOCK, bool NT>
CG_DEV void stage_codes2(const unsigned char *__restrict__ codes, const unsigned char *__restrict__ vcodes, int cfirst, int p1,
                         unsigned char *sc, unsigned char *sv) {
    const int t = threadIdx.x;
    for (long long base = cfirst; base < p1; base += 8 * BLOCK) {
        unsigned cw[2], vw[2];
        long long q[2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            q[rg] = base + (long long)rg * 4 * BLOCK + 4 * t;
            if (q[rg] < p1) {
                const unsigned *cp = reinterpret_cast<const unsigned *>(codes + q[rg]), *vp = reinterpret_cast<const unsigned *>(vcodes + q[rg]);
                cw[rg] = NT ? __builtin_nontemporal_load(cp) : *cp;
                vw[rg] = NT ? __builtin_nontemporal_load(vp) : *vp;
            }
        }
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
            if (q[rg] < p1) {
                const int o = (int)(q[rg] - cfirst);
                *reinterpret_cast<unsigned *>(sc + o) = cw[rg];
                *reinterpret_cast<unsigned *>(sv + o) = vw[rg];
            }
    }
}

}  // namespace cgamd
