/* Type-generic body of the CPU oracle; included four times by cg_oracle.c with
 *   T      value type (float, double, float complex, double complex)
 *   R      its real type
 *   SFX    function suffix (f32, f64, c64, c128)
 *   ISCPLX 0/1
 * TEST INFRASTRUCTURE ONLY -- see cg_oracle.c header.
 *
 * Arithmetic follows the reference kernels as text:
 *   real:    kernel/real/{spmv,vdot,axpy,aypx,sub}.cl
 *   complex: kernel/complex/{spmv,vdot,axpy,aypx,sub}.cl with cmplx.h:6-25
 *            (componentwise cadd/csub, cmul = (ax*bx - ay*by, ax*by + ay*bx),
 *             unconjugated dot: complex/vdot.cl:15).
 * Compiled with -ffp-contract=off: one rounding per mul and per add.
 */
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#if ISCPLX
static inline T FN(mul)(T a, T b) {           /* cmplx.h:20-25 */
    R ax = FN(re)(a), ay = FN(im)(a), bx = FN(re)(b), by = FN(im)(b);
    return FN(mk)(ax * bx - ay * by, ax * by + ay * bx);
}
static inline T FN(add)(T a, T b) { return FN(mk)(FN(re)(a) + FN(re)(b), FN(im)(a) + FN(im)(b)); } /* cmplx.h:6-11 */
static inline T FN(sub)(T a, T b) { return FN(mk)(FN(re)(a) - FN(re)(b), FN(im)(a) - FN(im)(b)); } /* cmplx.h:13-18 */
#else
static inline T FN(mul)(T a, T b) { return a * b; }
static inline T FN(add)(T a, T b) { return a + b; }
static inline T FN(sub)(T a, T b) { return a - b; }
#endif

/* ---- spmv: kernel/real/spmv.cl:5-50, kernel/complex/spmv.cl:7-53 ---------
 * mode 0 (reference order): lane L of WAVE_SIZE=32 sums elements p+L, p+L+32,...
 * left to right (:23-26), then the stride-1,2,4,8,16 adjacent-pair tree (:32-43).
 * mode bit 0 set (sequential): plain left-to-right row sum (scipy csr_matvec order,
 * which is what helmFE_var.py:520 `A.dot(d)` executes). */
static void FN(spmv)(int size, const T *aValues, const int *aPointers, const int *aCols,
                     const T *x, T *y, int nRHS, int mode) {
    for (int r = 0; r < nRHS; r++) {
        const T *xr = x + (size_t)r * size;
        T *yr = y + (size_t)r * size;
#pragma omp parallel for schedule(static)
        for (int row = 0; row < size; row++) {
            const int row_start = aPointers[row], row_end = aPointers[row + 1];
            if (mode & 1) {
                T s = FN(mk)(0, 0);
                for (int j = row_start; j < row_end; j++) s = FN(add)(s, FN(mul)(aValues[j], xr[aCols[j]]));
                yr[row] = s;
            } else {
                T lane[ORACLE_WAVE_SIZE];
                for (int L = 0; L < ORACLE_WAVE_SIZE; L++) {
                    T s = FN(mk)(0, 0);
                    for (int j = row_start + L; j < row_end; j += ORACLE_WAVE_SIZE)
                        s = FN(add)(s, FN(mul)(aValues[j], xr[aCols[j]]));
                    lane[L] = s;
                }
                for (int offset = 1; offset < ORACLE_WAVE_SIZE; offset <<= 1)
                    for (int L = 0; L < ORACLE_WAVE_SIZE; L += 2 * offset)
                        lane[L] = FN(add)(lane[L], lane[L + offset]);
                yr[row] = lane[0];
            }
        }
    }
}

/* ---- vdot: kernel/real/vdot.cl:2-38 + host sum clcg.c:274-279,317-324 ----
 * mode 0: per work-group of WG_SIZE=256 an adjacent-pair tree (:20-29), then the
 * host adds the partials sequentially in work-group order.
 * mode bit 1 set: numpy.dot order is implementation defined; use plain sequential sum
 * (tests compare with tolerance). */
static void FN(vdot)(int size, const T *a, const T *b, T *out, int nRHS, int mode) {
    const int workGroups = 1 + (size - 1) / ORACLE_WG_SIZE;          /* clcg.c:124 */
    for (int r = 0; r < nRHS; r++) {
        const T *ar = a + (size_t)r * size, *br = b + (size_t)r * size;
        if (mode & 2) {
            T s = FN(mk)(0, 0);
            for (int i = 0; i < size; i++) s = FN(add)(s, FN(mul)(ar[i], br[i]));
            out[r] = s;
            continue;
        }
        T *partials = (T *)malloc(sizeof(T) * (size_t)workGroups);
#pragma omp parallel for schedule(static)
        for (int wg = 0; wg < workGroups; wg++) {
            T loc[ORACLE_WG_SIZE];
            for (int k = 0; k < ORACLE_WG_SIZE; k++) {
                int i = wg * ORACLE_WG_SIZE + k;
                loc[k] = (i < size) ? FN(mul)(ar[i], br[i]) : FN(mk)(0, 0);   /* :11-18 */
            }
            for (int offset = 1; offset < ORACLE_WG_SIZE; offset <<= 1)
                for (int k = 0; k < ORACLE_WG_SIZE; k += 2 * offset)
                    loc[k] = FN(add)(loc[k], loc[k + offset]);
            partials[wg] = loc[0];
        }
        T s = FN(mk)(0, 0);
        for (int wg = 0; wg < workGroups; wg++) s = FN(add)(s, partials[wg]);  /* clcg.c:276-279 */
        out[r] = s;
        free(partials);
    }
}

/* ---- axpy: kernel/real/axpy.cl:2-17, complex/axpy.cl:4-22 ---------------- */
static void FN(axpy)(int size, const T *x, T *y, const T *a, int aSign, int nRHS) {
    for (int r = 0; r < nRHS; r++) {
        const T *xr = x + (size_t)r * size;
        T *yr = y + (size_t)r * size;
        const T ar = a[r];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < size; i++)
            yr[i] = aSign ? FN(add)(yr[i], FN(mul)(ar, xr[i])) : FN(sub)(yr[i], FN(mul)(ar, xr[i]));
    }
}

/* ---- aypx: kernel/real/aypx.cl:2-10 (y*a + x), complex/aypx.cl:4-12 (cmul(a,y) + x) */
static void FN(aypx)(int size, const T *x, T *y, const T *a, int nRHS) {
    for (int r = 0; r < nRHS; r++) {
        const T *xr = x + (size_t)r * size;
        T *yr = y + (size_t)r * size;
        const T ar = a[r];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < size; i++)
#if ISCPLX
            yr[i] = FN(add)(FN(mul)(ar, yr[i]), xr[i]);
#else
            yr[i] = yr[i] * ar + xr[i];
#endif
    }
}

/* ---- sub: kernel/real/sub.cl:2-12, complex/sub.cl:4-15 ------------------- */
static void FN(vsub)(int size, const T *a, const T *b, T *result, int nRHS) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)size * nRHS; i++) result[i] = FN(sub)(a[i], b[i]);
}

/* ---- scalar division alpha = deltaNew/dq, beta = deltaNew/deltaOld --------
 * clcg.c:326-327,389-391: C99 `/` on float / float complex (numpy csingle
 * division in cl.py:145,177). */
static inline T FN(divs)(T a, T b) { return a / b; }

/* ---- cg driver: clcg.c:250-430 == cl.py:96-200 == helmFE_var.py:507-544 ---
 * exactly nIterations iterations, no convergence test (clcg.c:297).
 * history (optional): (nIterations+1) x nRHS, history[0] = r0.r0 (clcg.c:274-292),
 * history[k] = deltaNew after iteration k (clcg.c:384-387). */
static void FN(cg)(int size, int nonZeros, const T *aValues, const T *b, const int *aPointers,
                   const int *aCols, T *x, int nRHS, int nIterations, T *history, int mode) {
    (void)nonZeros;
    const size_t nv = (size_t)size * nRHS;
    T *r = (T *)malloc(sizeof(T) * nv), *d = (T *)malloc(sizeof(T) * nv), *q = (T *)malloc(sizeof(T) * nv);
    T *deltaNew = (T *)calloc(nRHS, sizeof(T)), *deltaOld = (T *)calloc(nRHS, sizeof(T));
    T *dq = (T *)calloc(nRHS, sizeof(T)), *alpha = (T *)calloc(nRHS, sizeof(T)), *beta = (T *)calloc(nRHS, sizeof(T));

    FN(spmv)(size, aValues, aPointers, aCols, x, q, nRHS, mode);      /* clcg.c:255 */
    FN(vsub)(size, b, q, r, nRHS);                                    /* :260 */
    memcpy(d, r, sizeof(T) * nv);                                     /* :264 */
    FN(vdot)(size, r, r, deltaNew, nRHS, mode);                       /* :268-279 */
    for (int k = 0; k < nRHS; k++) { deltaOld[k] = deltaNew[k]; if (history) history[k] = deltaNew[k]; }

    for (int it = 0; it < nIterations; it++) {                        /* :297 */
        FN(spmv)(size, aValues, aPointers, aCols, d, q, nRHS, mode);  /* :299-305 */
        FN(vdot)(size, d, q, dq, nRHS, mode);                         /* :309-324 */
        for (int k = 0; k < nRHS; k++) alpha[k] = FN(divs)(deltaNew[k], dq[k]);   /* :326-327 */
        FN(axpy)(size, d, x, alpha, 1, nRHS);                         /* :338-342 x += alpha d */
        FN(axpy)(size, q, r, alpha, 0, nRHS);                         /* :345-349 r -= alpha q */
        for (int k = 0; k < nRHS; k++) deltaOld[k] = deltaNew[k];     /* :352-353 */
        FN(vdot)(size, r, r, deltaNew, nRHS, mode);                   /* :369-387 */
        for (int k = 0; k < nRHS; k++) beta[k] = FN(divs)(deltaNew[k], deltaOld[k]);  /* :389-391 */
        FN(aypx)(size, r, d, beta, nRHS);                             /* :415 d = beta d + r */
        if (history) for (int k = 0; k < nRHS; k++) history[(size_t)(it + 1) * nRHS + k] = deltaNew[k];
    }
    free(r); free(d); free(q); free(deltaNew); free(deltaOld); free(dq); free(alpha); free(beta);
}

#undef FN
#undef CAT
#undef CAT_
