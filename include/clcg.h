/* clcg.h -- legacy drop-in C ABI of the MI355X-native CG solver.
 *
 * Binary-compatible replacement for the single entry point of the reference
 * (reference clcg.h:3-5, implemented in reference clcg.c:111-466) so that the
 * existing callers keep working unchanged:
 *   - ctypes: CDLL("./build/liboclcg.so"); libcg.connect(); libcg.cg(...)
 *             (reference p_h-PY_C-CL.py:38-39,1948-1950,1979-1982)
 *   - C:      main.c:56
 *
 * Argument ORDER is values, b, pointers, cols, x (as in the reference header;
 * the reference README lists them in a different order).
 *
 *   size         number of rows/cols N of the square CSR matrix
 *   nonZeros     number of stored entries (aPointers[size])
 *   aValues      nonZeros values; float, or interleaved (re,im) float pairs when
 *                isComplex != 0 (reference clcg.c:154-158)
 *   b            nRHS right-hand sides, RHS-major: element i of RHS r at
 *                b[i + r*size] (reference kernel/real/spmv.cl:25,48)
 *   aPointers    size+1 row pointers, 0-based, int32
 *   aCols        nonZeros column indices, 0-based, int32, any order within a row
 *   x            in: initial guess, out: solution; same layout as b
 *   nRHS         number of right-hand sides solved as independent CG runs with
 *                their own alpha/beta (reference clcg.c:317-333,376-392)
 *   nIterations  exactly this many iterations are run; there is no convergence
 *                test (reference clcg.c:297)
 *   isComplex    0: real CG;  !=0: unconjugated complex-symmetric CG (COCG)
 *                (reference kernel/complex/vdot.cl:15)
 *
 * Returns x.  All arrays are caller-owned host memory, alive for the call only;
 * the call is synchronous and stateless (device state is created and released
 * inside, like the reference).  Errors: the reference has no error channel
 * (checkClSuccess prints and continues, clcg.c:52-56); this implementation
 * prints "error -- ..." to stderr, leaves x untouched and still returns x.
 * It never falls back to a CPU path: without a HIP device it reports the error.
 */
#ifndef CGAMD_CLCG_H
#define CGAMD_CLCG_H

#ifdef __cplusplus
extern "C" {
#endif

float *cg(int size, int nonZeros, const float *aValues, const float *b, const int *aPointers,
          const int *aCols, float *x, int nRHS, int nIterations, int isComplex);

/* The reference drivers call libcg.connect() right after loading the library
 * (p_h-PY_C-CL.py:39) although reference clcg.c exports no such symbol.  Here it
 * selects device 0 and warms the HIP runtime; failure is reported on stderr. */
void connect(void);

#ifdef __cplusplus
}
#endif
#endif /* CGAMD_CLCG_H */
