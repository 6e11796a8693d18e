// `connect()` -- the reference drivers call libcg.connect() right after CDLL("./build/liboclcg.so")
// (reference p_h-PY_C-CL.py:38-39) although reference clcg.c exports no such symbol.  Kept in its own
// translation unit and linked ONLY into liboclcg.so (the ctypes drop-in): a link-time C user must use
// libcgamd.so, because a global `connect` would shadow the sockets connect(2) for the whole program.
#include <cstdio>

#include "../../include/cgamd.h"
#include "../../include/clcg.h"

extern "C" void connect(void) {
    const int n = cgamd_device_count();
    if (n <= 0) {
        fprintf(stderr, "error -- connect: no HIP device (%s)\n", cgamd_last_error());
        return;
    }
    cgamd_ctx *ctx = nullptr;
    if (cgamd_ctx_create(0, &ctx) != CGAMD_OK) {
        fprintf(stderr, "error -- connect: %s\n", cgamd_last_error());
        return;
    }
    cgamd_ctx_destroy(ctx);
}
