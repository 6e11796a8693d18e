// 16-byte packs and typed / non-temporal load-store helpers shared by the gfx950 kernel files.
#pragma once
#include "device_types.h"

namespace cgamd {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---- 16-byte packs -----------------------------------------------------------------------------
template <typename T> struct Pack {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
} __attribute__((aligned(16)));

template <typename T> CG_DEV Pack<T> ld_pack(const T *p) { return *reinterpret_cast<const Pack<T> *>(p); }
template <typename T> CG_DEV void st_pack(T *p, const Pack<T> &v) { *reinterpret_cast<Pack<T> *>(p) = v; }
// non-temporal forms for data that is touched once per iteration (x) or for the last time (q): keeps the
// vectors that are re-read (d, r) in L2 / Infinity Cache
template <typename T> CG_DEV Pack<T> ld_pack_nt(const T *p) {
    union { u32x4 raw; Pack<T> v; } u;
    u.raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return u.v;
}
template <typename T> CG_DEV void st_pack_nt(T *p, const Pack<T> &v) {
    union { u32x4 raw; Pack<T> v; } u;
    u.v = v;
    __builtin_nontemporal_store(u.raw, reinterpret_cast<u32x4 *>(p));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// 4 consecutive values by 16-byte loads, register to register (a union of ext-vectors and HIP vector structs
// sent the complex128 instance through scratch memory: 2x slower).  NT = non-temporal.
template <typename V, bool NT> CG_DEV V ld16(const void *p) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p)) : *reinterpret_cast<const V *>(p);
}
template <bool NT> CG_DEV void ld4(const float *p, float (&o)[4]) {
    const f32x4 w = ld16<f32x4, NT>(p);
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
}
template <bool NT> CG_DEV void ld4(const double *p, double (&o)[4]) {
    const f64x2 a = ld16<f64x2, NT>(p), b = ld16<f64x2, NT>(p + 2);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
template <bool NT> CG_DEV void ld4(const float2 *p, float2 (&o)[4]) {
    const f32x4 a = ld16<f32x4, NT>(p), b = ld16<f32x4, NT>(p + 2);
    o[0] = make_float2(a.x, a.y); o[1] = make_float2(a.z, a.w); o[2] = make_float2(b.x, b.y); o[3] = make_float2(b.z, b.w);
}
template <bool NT> CG_DEV void ld4(const double2 *p, double2 (&o)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f64x2 a = ld16<f64x2, NT>(p + i);
        o[i] = make_double2(a.x, a.y);
    }
}
template <typename T> CG_DEV void ld4_nt(const T *p, T (&out)[4]) { ld4<true>(p, out); }

}  // namespace cgamd
