// Host-side helpers shared by the kernel launchers.
#pragma once
#include <initializer_list>
#include <string>

#include "cgamd_internal.h"

namespace cgamd {

#define CG_DISPATCH(dtype, FN, ...)                                         \
    switch (dtype) {                                                        \
    case CGAMD_F32: return FN<float>(__VA_ARGS__);                          \
    case CGAMD_F64: return FN<double>(__VA_ARGS__);                         \
    case CGAMD_C64: return FN<float2>(__VA_ARGS__);                         \
    case CGAMD_C128: return FN<double2>(__VA_ARGS__);                       \
    default: return fail(CGAMD_ERR_INVALID, "bad dtype");                   \
    }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return CGAMD_OK;
}

// 16-byte vector path of the streaming kernels: every pointer aligned, and with several right-hand sides a leading dimension
// that keeps each of them aligned
inline bool vec_ok(int dtype, long long ld, int nrhs, std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    if (nrhs > 1 && ((ld * (long long)dtype_size(dtype)) & 15)) return false;
    return true;
}

}  // namespace cgamd
