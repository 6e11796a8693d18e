// Matrix-Market coordinate reader -> expanded, canonical CSR.
//
// Replaces what the reference CLI gets from the un-vendored BeBOP converter (reference main.c:20-33:
// load_sparse_matrix(MATRIX_MARKET) -> sparse_matrix_expand_symmetric_storage -> convert to CSR).
// Supports field real|double|complex|integer|pattern and symmetry general|symmetric|hermitian|
// skew-symmetric; indices 1-based in the file, 0-based out; duplicate entries are summed; columns
// are sorted inside each row.  Pattern entries get the value 1.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "cgamd_internal.h"

using namespace cgamd;

namespace {
std::string lower(std::string s) {
    for (auto &ch : s) ch = (char)tolower((unsigned char)ch);
    return s;
}
struct Entry { int r, c; double re, im; };
}  // namespace

extern "C" {

void cgamd_mm_free(void *p) { free(p); }

int cgamd_mm_read(const char *path, int *size, long long *nnz_out, int *is_complex, double **values, int **pointers,
                  int **cols_out) {
    if (!path || !size || !nnz_out || !is_complex || !values || !pointers || !cols_out)
        return fail(CGAMD_ERR_INVALID, "mm_read: null argument");
    *values = nullptr; *pointers = nullptr; *cols_out = nullptr;
    FILE *f = fopen(path, "r");
    if (!f) return fail(CGAMD_ERR_IO, std::string("mm_read: cannot open ") + path);
    std::vector<char> line(1 << 16);
    auto bail = [&](const std::string &m) { fclose(f); return fail(CGAMD_ERR_IO, "mm_read: " + m); };

    if (!fgets(line.data(), (int)line.size(), f)) return bail("empty file");
    char banner[64], object[64], format[64], field[64], symmetry[64];
    if (sscanf(line.data(), "%63s %63s %63s %63s %63s", banner, object, format, field, symmetry) != 5)
        return bail("malformed banner line");
    if (lower(banner) != "%%matrixmarket" || lower(object) != "matrix") return bail("not a MatrixMarket matrix file");
    if (lower(format) != "coordinate") return bail("only 'coordinate' format is supported");
    const std::string fld = lower(field), sym = lower(symmetry);
    const bool cplx = fld == "complex", pattern = fld == "pattern";
    if (!cplx && !pattern && fld != "real" && fld != "double" && fld != "integer") return bail("unknown field '" + fld + "'");
    const bool general = sym == "general", symmetric = sym == "symmetric", herm = sym == "hermitian",
               skew = sym == "skew-symmetric";
    if (!general && !symmetric && !herm && !skew) return bail("unknown symmetry '" + sym + "'");

    long long M = 0, N = 0, L = 0;
    for (;;) {
        if (!fgets(line.data(), (int)line.size(), f)) return bail("missing size line");
        const char *p = line.data();
        while (*p && isspace((unsigned char)*p)) ++p;
        if (*p == '%' || *p == 0) continue;
        if (sscanf(p, "%lld %lld %lld", &M, &N, &L) != 3) return bail("malformed size line");
        break;
    }
    if (M != N) return bail("matrix is not square");
    if (M < 1 || M > 2147483647LL || L < 0) return bail("bad dimensions");

    std::vector<Entry> ent;
    ent.reserve((size_t)(general ? L : 2 * L));
    for (long long k = 0; k < L;) {
        if (!fgets(line.data(), (int)line.size(), f)) return bail("unexpected end of file after " + std::to_string(k) + " entries");
        char *p = line.data();
        while (*p && isspace((unsigned char)*p)) ++p;
        if (*p == '%' || *p == 0) continue;
        char *end;
        const long long i = strtoll(p, &end, 10);
        if (end == p) return bail("malformed entry line");
        p = end;
        const long long j = strtoll(p, &end, 10);
        if (end == p) return bail("malformed entry line");
        p = end;
        double re = 1.0, im = 0.0;
        if (!pattern) {
            re = strtod(p, &end);
            if (end == p) return bail("missing value");
            p = end;
            if (cplx) {
                im = strtod(p, &end);
                if (end == p) return bail("missing imaginary part");
            }
        }
        if (i < 1 || i > M || j < 1 || j > N) return bail("index out of range in entry " + std::to_string(k + 1));
        ent.push_back({(int)(i - 1), (int)(j - 1), re, im});
        if (!general && i != j) {
            if (symmetric) ent.push_back({(int)(j - 1), (int)(i - 1), re, im});
            else if (herm) ent.push_back({(int)(j - 1), (int)(i - 1), re, -im});
            else ent.push_back({(int)(j - 1), (int)(i - 1), -re, -im});
        }
        ++k;
    }
    fclose(f);

    const int n = (int)M;
    // counting sort by row, then sort each row by column, then merge duplicates
    std::vector<long long> start((size_t)n + 1, 0);
    for (const auto &e : ent) ++start[(size_t)e.r + 1];
    std::partial_sum(start.begin(), start.end(), start.begin());
    std::vector<Entry> byrow(ent.size());
    {
        std::vector<long long> pos(start.begin(), start.end() - 1);
        for (const auto &e : ent) byrow[(size_t)pos[e.r]++] = e;
    }
    ent.clear(); ent.shrink_to_fit();
    int *ptr = (int *)malloc(sizeof(int) * ((size_t)n + 1));
    if (!ptr) return fail(CGAMD_ERR_ALLOC, "mm_read: out of memory");
    long long w = 0;
    ptr[0] = 0;
    for (int r = 0; r < n; ++r) {
        auto b = byrow.begin() + start[r], e = byrow.begin() + start[(size_t)r + 1];
        std::stable_sort(b, e, [](const Entry &x, const Entry &y) { return x.c < y.c; });
        for (auto it = b; it != e; ++it) {
            if (w > ptr[r] && byrow[(size_t)w - 1].c == it->c) {
                byrow[(size_t)w - 1].re += it->re;
                byrow[(size_t)w - 1].im += it->im;
            } else {
                byrow[(size_t)w++] = *it;
            }
        }
        if (w > 2147483647LL) { free(ptr); return fail(CGAMD_ERR_IO, "mm_read: more than 2^31-1 entries after expansion"); }
        ptr[r + 1] = (int)w;
    }
    int *cols = (int *)malloc(sizeof(int) * (size_t)std::max<long long>(w, 1));
    double *vals = (double *)malloc(sizeof(double) * (size_t)std::max<long long>(w, 1) * (cplx ? 2 : 1));
    if (!cols || !vals) { free(ptr); free(cols); free(vals); return fail(CGAMD_ERR_ALLOC, "mm_read: out of memory"); }
    for (long long k = 0; k < w; ++k) {
        cols[k] = byrow[(size_t)k].c;
        if (cplx) { vals[2 * k] = byrow[(size_t)k].re; vals[2 * k + 1] = byrow[(size_t)k].im; }
        else vals[k] = byrow[(size_t)k].re;
    }
    *size = n; *nnz_out = w; *is_complex = cplx ? 1 : 0;
    *values = vals; *pointers = ptr; *cols_out = cols;
    return CGAMD_OK;
}

}  // extern "C"
