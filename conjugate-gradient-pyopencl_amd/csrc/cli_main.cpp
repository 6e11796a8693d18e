// oclcgex -- command-line front end with the reference's argv (reference main.c:13-61):
//     oclcgex <matrix.mtx> <nRHS> <isComplex> <nIterations> [--double] [--device N] [--quiet] [--history <file>]
// Matrix-Market file -> symmetric storage expanded -> CSR; b[r*n+i] = (r+1)*5, x0 = 0 (main.c:41-46).
// The reference prints nothing and discards x; this front end adds the residual history and timing.
// Values are cast to single precision as the reference does (main.c:49-53) unless --double is given.
// --history <file>: the whole residual history delta_k[r] = r_k.r_k (k = 0..nIterations, r < nRHS; what the reference computes and
// drops, clcg.c:274-292,384-387) as raw little-endian fp64, (nIterations + 1) x nRHS values -- pairs (re, im) for complex solves.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cgamd.h"

int main(int argc, char *argv[]) {
    if (argc < 5) {
        fprintf(stderr, "Usage: %s <input matrix file> <number of RHS> <is complex> <number of iterations> [--double] [--device N] [--quiet] [--history <file>]\n", argv[0]);
        return 1;
    }
    const char *path = argv[1];
    const int nRHS = atoi(argv[2]), isComplex = atoi(argv[3]), nIterations = atoi(argv[4]);
    bool dbl = false, quiet = false;
    int device = 0;
    const char *history_path = nullptr;
    for (int i = 5; i < argc; ++i) {
        if (!strcmp(argv[i], "--double")) dbl = true;
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--history") && i + 1 < argc) history_path = argv[++i];
    }
    if (nRHS < 1 || nIterations < 0) { fprintf(stderr, "bad nRHS / nIterations\n"); return 1; }

    int n = 0, fileComplex = 0;
    long long nnz = 0;
    double *vals = nullptr;
    int *ptr = nullptr, *cols = nullptr;
    if (cgamd_mm_read(path, &n, &nnz, &fileComplex, &vals, &ptr, &cols) != CGAMD_OK) {
        printf("Could not read matrix\n");   // reference main.c:22
        fprintf(stderr, "%s\n", cgamd_last_error());
        return 1;
    }
    // value conversion: file (double / double complex) -> solver dtype
    const int dtype = isComplex ? (dbl ? CGAMD_C128 : CGAMD_C64) : (dbl ? CGAMD_F64 : CGAMD_F32);
    const size_t vs = cgamd_dtype_size(dtype);
    std::vector<char> a(vs * (size_t)nnz), b(vs * (size_t)n * nRHS), x(vs * (size_t)n * nRHS, 0), hist(vs * (size_t)(nIterations + 1) * nRHS);
    auto put = [&](char *base, size_t idx, double re, double im) {
        switch (dtype) {
        case CGAMD_F32: ((float *)base)[idx] = (float)re; break;
        case CGAMD_F64: ((double *)base)[idx] = re; break;
        case CGAMD_C64: ((float *)base)[2 * idx] = (float)re; ((float *)base)[2 * idx + 1] = (float)im; break;
        default: ((double *)base)[2 * idx] = re; ((double *)base)[2 * idx + 1] = im; break;
        }
    };
    for (long long k = 0; k < nnz; ++k) put(a.data(), (size_t)k, fileComplex ? vals[2 * k] : vals[k], fileComplex ? vals[2 * k + 1] : 0.0);
    for (int r = 0; r < nRHS; ++r)
        for (int i = 0; i < n; ++i) put(b.data(), (size_t)r * n + i, (r + 1) * 5.0, 0.0);   // main.c:44

    const auto t0 = std::chrono::steady_clock::now();
    const int rc = cgamd_cg(dtype, n, nnz, a.data(), b.data(), ptr, cols, x.data(), nRHS, nIterations, hist.data(), device);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    cgamd_mm_free(vals); cgamd_mm_free(ptr); cgamd_mm_free(cols);
    if (rc != CGAMD_OK) { fprintf(stderr, "error -- cg failed (%d): %s\n", rc, cgamd_last_error()); return 2; }
    if (history_path) {
        const bool cplx = dtype == CGAMD_C64 || dtype == CGAMD_C128;
        const size_t count = (size_t)(nIterations + 1) * nRHS * (cplx ? 2 : 1);
        std::vector<double> out(count);
        for (size_t i = 0; i < count; ++i)
            out[i] = (dtype == CGAMD_F32 || dtype == CGAMD_C64) ? (double)((const float *)hist.data())[i] : ((const double *)hist.data())[i];
        FILE *f = fopen(history_path, "wb");
        if (!f || fwrite(out.data(), sizeof(double), count, f) != count || fclose(f) != 0) {
            fprintf(stderr, "error -- could not write the residual history to %s\n", history_path);
            return 3;
        }
    }
    if (!quiet) {
        auto mag = [&](size_t idx) {
            switch (dtype) {
            case CGAMD_F32: return std::fabs((double)((float *)hist.data())[idx]);
            case CGAMD_F64: return std::fabs(((double *)hist.data())[idx]);
            case CGAMD_C64: return std::hypot((double)((float *)hist.data())[2 * idx], (double)((float *)hist.data())[2 * idx + 1]);
            default: return std::hypot(((double *)hist.data())[2 * idx], ((double *)hist.data())[2 * idx + 1]);
            }
        };
        printf("matrix %s: n=%d nnz=%lld %s, nRHS=%d, %d iterations, dtype=%s\n", path, n, nnz, fileComplex ? "complex" : "real",
               nRHS, nIterations, dtype == CGAMD_F32 ? "f32" : dtype == CGAMD_F64 ? "f64" : dtype == CGAMD_C64 ? "c64" : "c128");
        const int step = nIterations > 20 ? nIterations / 10 : 1;
        for (int k = 0; k <= nIterations; k += step) {
            printf("iteration %6d  |delta| {", k);
            for (int r = 0; r < nRHS && r < 4; ++r) printf(" %.6e", mag((size_t)k * nRHS + r));
            printf("%s }\n", nRHS > 4 ? " ..." : "");
        }
        printf("wall %.3f s (upload + JIT-free solve + download), %.1f iterations/s\n", sec, nIterations / sec);
    }
    return 0;
}
