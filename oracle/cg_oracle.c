/* CPU oracle for the CG hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm (spmv, vdot, axpy, aypx, sub
 * and the cg() recurrence) for f32 / f64 / c64 / c128.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (conjugate-gradient-pyopencl_amd/) never does and fails loudly
 * when its HIP library is missing.
 *
 * Parity pin: tests/test_oracle_golden.py checks this library against
 * tests/golden/cg_iterates.npz, produced by oracle/make_golden.py from the
 * UNMODIFIED reference helmFE_var.CG (helmFE_var.py:507-544) in the build
 * container.  The reference's own C/OpenCL path (clcg.c + kernel/ *.cl) is
 * unbuildable/unrunnable here (needs BeBOP headers that are not vendored and an
 * OpenCL device; the container has none), so there is no oracle/_ref.
 *
 * Entry points take a dtype code: 0=f32 1=f64 2=c64 3=c128 (interleaved re,im).
 * mode bits: 0 = reference summation order everywhere; bit 0 = sequential row sums
 * in spmv; bit 1 = sequential dot.  3 = everything sequential, 1 = sequential rows + reference-order dot
 * (deterministic and OpenMP-parallel: the CPU baseline).
 */
#include <complex.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_WAVE_SIZE 32   /* clcg.c:42, cl.py:6 */
#define ORACLE_WG_SIZE 256    /* clcg.c:37, cl.py:7 */

/* per-type accessors */
static inline float re_f32(float a) { return a; }
static inline float im_f32(float a) { (void)a; return 0.f; }
static inline float mk_f32(float a, float b) { (void)b; return a; }
static inline double re_f64(double a) { return a; }
static inline double im_f64(double a) { (void)a; return 0.; }
static inline double mk_f64(double a, double b) { (void)b; return a; }
static inline float re_c64(float complex a) { return crealf(a); }
static inline float im_c64(float complex a) { return cimagf(a); }
static inline float complex mk_c64(float a, float b) { return CMPLXF(a, b); }
static inline double re_c128(double complex a) { return creal(a); }
static inline double im_c128(double complex a) { return cimag(a); }
static inline double complex mk_c128(double a, double b) { return CMPLX(a, b); }

#define T float
#define R float
#define SFX f32
#define ISCPLX 0
#include "cg_oracle_impl.h"
#undef T
#undef R
#undef SFX
#undef ISCPLX

#define T double
#define R double
#define SFX f64
#define ISCPLX 0
#include "cg_oracle_impl.h"
#undef T
#undef R
#undef SFX
#undef ISCPLX

#define T float complex
#define R float
#define SFX c64
#define ISCPLX 1
#include "cg_oracle_impl.h"
#undef T
#undef R
#undef SFX
#undef ISCPLX

#define T double complex
#define R double
#define SFX c128
#define ISCPLX 1
#include "cg_oracle_impl.h"
#undef T
#undef R
#undef SFX
#undef ISCPLX

#define DISPATCH(call_f32, call_f64, call_c64, call_c128) \
    switch (dtype) {                                      \
    case 0: call_f32; return 0;                           \
    case 1: call_f64; return 0;                           \
    case 2: call_c64; return 0;                           \
    case 3: call_c128; return 0;                          \
    default: return -1;                                   \
    }

int cgo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

int cgo_spmv(int dtype, int size, const void *aValues, const int *aPointers, const int *aCols,
             const void *x, void *y, int nRHS, int mode) {
    DISPATCH(spmv_f32(size, aValues, aPointers, aCols, x, y, nRHS, mode),
             spmv_f64(size, aValues, aPointers, aCols, x, y, nRHS, mode),
             spmv_c64(size, aValues, aPointers, aCols, x, y, nRHS, mode),
             spmv_c128(size, aValues, aPointers, aCols, x, y, nRHS, mode))
}

int cgo_vdot(int dtype, int size, const void *a, const void *b, void *out, int nRHS, int mode) {
    DISPATCH(vdot_f32(size, a, b, out, nRHS, mode), vdot_f64(size, a, b, out, nRHS, mode),
             vdot_c64(size, a, b, out, nRHS, mode), vdot_c128(size, a, b, out, nRHS, mode))
}

int cgo_axpy(int dtype, int size, const void *x, void *y, const void *a, int aSign, int nRHS) {
    DISPATCH(axpy_f32(size, x, y, a, aSign, nRHS), axpy_f64(size, x, y, a, aSign, nRHS),
             axpy_c64(size, x, y, a, aSign, nRHS), axpy_c128(size, x, y, a, aSign, nRHS))
}

int cgo_aypx(int dtype, int size, const void *x, void *y, const void *a, int nRHS) {
    DISPATCH(aypx_f32(size, x, y, a, nRHS), aypx_f64(size, x, y, a, nRHS),
             aypx_c64(size, x, y, a, nRHS), aypx_c128(size, x, y, a, nRHS))
}

int cgo_sub(int dtype, int size, const void *a, const void *b, void *result, int nRHS) {
    DISPATCH(vsub_f32(size, a, b, result, nRHS), vsub_f64(size, a, b, result, nRHS),
             vsub_c64(size, a, b, result, nRHS), vsub_c128(size, a, b, result, nRHS))
}

/* 3-D 7-point Laplacian, x fastest, Dirichlet, diag 6 / off-diag -1 (SURVEY 8d "M"; no reference generator exists for
 * it): the full-size input of bench.py's cpu_baseline leg, built in parallel so that the N=10M system costs a fraction
 * of a second.  Same matrix as oracle/cg_numpy.py::laplace3d (tests/test_oracle_golden.py compares them).
 * aPointers: nx*ny*nz+1 ints; aValues/aCols sized from the closed form 7n - 2(nx ny + ny nz + nx nz).  Real types only. */
int cgo_laplace3d(int dtype, int nx, int ny, int nz, void *aValues, int *aPointers, int *aCols) {
    if ((dtype != 0 && dtype != 1) || nx < 1 || ny < 1 || nz < 1) return -1;
    const long long pl = (long long)nx * ny, n = pl * nz;
    aPointers[0] = 0;
    /* entries before row i, closed form: 7 i - (missing neighbours of rows < i) */
#pragma omp parallel for schedule(static)
    for (long long i = 0; i <= n; i++) {
        const long long x0 = (i + nx - 1) / nx, x1 = i / nx;
        const long long full = i / pl, rem = i % pl;
        const long long y0 = full * nx + (rem < nx ? rem : nx);
        const long long y1 = full * nx + (rem > pl - nx ? rem - (pl - nx) : 0);
        const long long z0 = i < pl ? i : pl, z1 = i > n - pl ? i - (n - pl) : 0;
        aPointers[i] = (int)(7 * i - (x0 + x1 + y0 + y1 + z0 + z1));
    }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; i++) {
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / pl);
        long long p = aPointers[i];
#define PUT(c, v) do { aCols[p] = (int)(c); if (dtype == 0) ((float *)aValues)[p] = (float)(v); else ((double *)aValues)[p] = (v); ++p; } while (0)
        if (iz > 0) PUT(i - pl, -1.0);
        if (iy > 0) PUT(i - nx, -1.0);
        if (ix > 0) PUT(i - 1, -1.0);
        PUT(i, 6.0);
        if (ix < nx - 1) PUT(i + 1, -1.0);
        if (iy < ny - 1) PUT(i + nx, -1.0);
        if (iz < nz - 1) PUT(i + pl, -1.0);
#undef PUT
    }
    return 0;
}

/* Same argument order as the reference cg() (clcg.h:3-5: values, b, pointers,
 * cols, x) plus dtype in front and history/mode behind. */
int cgo_cg(int dtype, int size, int nonZeros, const void *aValues, const void *b,
           const int *aPointers, const int *aCols, void *x, int nRHS, int nIterations,
           void *history, int mode) {
    DISPATCH(cg_f32(size, nonZeros, aValues, b, aPointers, aCols, x, nRHS, nIterations, history, mode),
             cg_f64(size, nonZeros, aValues, b, aPointers, aCols, x, nRHS, nIterations, history, mode),
             cg_c64(size, nonZeros, aValues, b, aPointers, aCols, x, nRHS, nIterations, history, mode),
             cg_c128(size, nonZeros, aValues, b, aPointers, aCols, x, nRHS, nIterations, history, mode))
}
