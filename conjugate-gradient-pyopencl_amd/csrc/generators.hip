// P1 finite-element Helmholtz matrices on a structured triangulation of a rectangle, generated straight into device CSR:
//   * helmFE_var(N, omega, C, rho, Nhoriz, Nvert)   -- reference helmFE_var.py:9-331 (BASELINE config 3: N = 500, omega = 12, C = 1,
//     rho = 0.15): variable wave speed, k = omega / C on every mesh square, absorption cr = 1 + i rho, impedance term i k;
//   * local_rect(N, k, eps, eta, L, Nhoriz, Nvert)  -- reference p_h-PY_C-CL.py:1439-1639 (the sub-domain matrices as_prec solves
//     with CG): constant k, shift k^2 + i eps, impedance parameter eta.
// Both have the same pattern: node (j, m) = column j, row m, index m Nhoriz + j couples to +-1, +-Nhoriz and +-(Nhoriz + 1)
// (7 entries inside, 5 on an edge, 4 / 3 in the corners: nnz = 2 (5 Nhoriz - 3) + (Nvert - 2)(7 Nhoriz - 4)), so row
// pointers are closed-form and one thread writes one row, columns ascending = scipy's canonical CSR of the reference's
// coordinate lists.  Every value is evaluated in the reference's own operation order in complex double (the library is built
// with -ffp-contract=off), then rounded to the requested type: tests/test_gpu_generators.py holds the result to the golden
// matrices produced by the unmodified reference (pattern exact, values to 1e-15) and to the CPU restatement at N = 500.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cgamd_internal.h"
#include "device_types.h"
#include "launch_util.h"

namespace cgamd {
namespace {

struct Cx {
    double re, im;
};
CG_DEV Cx cx(double re, double im) { return Cx{re, im}; }
CG_DEV Cx operator-(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
CG_DEV Cx operator*(Cx a, double s) { return Cx{a.re * s, a.im * s}; }      // complex x (s + 0i): the cross terms are exact zeros
CG_DEV Cx operator/(Cx a, double s) { return Cx{a.re / s, a.im / s}; }      // Smith's division by (s + 0i): ratio 0, denominator s
CG_DEV Cx neg(Cx a) { return Cx{-a.re, -a.im}; }
CG_DEV Cx real_minus(double r, Cx a) { return Cx{r - a.re, 0.0 - a.im}; }

struct FeArgs {
    int Nh, Nv, variable;
    double h, h2;
    double omega, rho;            // helmFE_var
    double k2, eps, eta;          // local_rect
    const double *C;              // [Nv - 1][Nh - 1] wave speed per square (helmFE_var)
};

// the zeroth-order coefficient on a square and the impedance coefficient on its boundary edge
struct Sq {
    double k;                     // helmFE_var: omega / C; local_rect: unused
};
CG_DEV double sq_k(const FeArgs &a, int m, int j) {
    m = min(max(m, 0), a.Nv - 2);
    j = min(max(j, 0), a.Nh - 2);
    return a.variable ? a.omega / a.C[(long long)m * (a.Nh - 1) + j] : 0.0;
}
// cr * s (helmFE_var: s a real combination of k^2) / (k2 + i eps) * w (local_rect: w the combination's weight)
CG_DEV Cx zterm(const FeArgs &a, double s_var, double w_const) {
    if (a.variable) return cx(s_var, a.rho * s_var);                        // (1 + i rho) * (s + 0i)
    return cx(a.k2 * w_const, a.eps * w_const);
}

CG_DEV long long fe_row_start(int m, int j, int Nh, int Nv) {
    const long long edge_row = 5LL * Nh - 3, mid_row = 7LL * Nh - 4;
    long long p = m == 0 ? 0 : edge_row + (long long)(m - 1) * mid_row;
    if (j > 0) {
        if (m == 0) p += 4 + 5LL * (j - 1);
        else if (m == Nv - 1) p += 3 + 5LL * (j - 1);
        else p += 5 + 7LL * (j - 1);
    }
    return p;
}

template <typename T> CG_DEV T to_val(Cx v);
template <> CG_DEV float2 to_val<float2>(Cx v) { return make_float2((float)v.re, (float)v.im); }
template <> CG_DEV double2 to_val<double2>(Cx v) { return make_double2(v.re, v.im); }

template <typename T>
__global__ __launch_bounds__(256) void gen_helm_fe_kernel(FeArgs a, T *__restrict__ vals, int *__restrict__ ptr, int *__restrict__ cols) {
    const long long nn = (long long)a.Nh * a.Nv;
    const double h = a.h, h2 = a.h2;
    for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row <= nn; row += (long long)gridDim.x * blockDim.x) {
        if (row == nn) { ptr[nn] = (int)(2 * (5LL * a.Nh - 3) + (long long)(a.Nv - 2) * (7LL * a.Nh - 4)); break; }
        const int m = (int)(row / a.Nh), j = (int)(row % a.Nh);
        long long p = fe_row_start(m, j, a.Nh, a.Nv);
        ptr[row] = (int)p;
        const bool bot = m == 0, top = m == a.Nv - 1, left = j == 0, right = j == a.Nh - 1;
        const double knw = sq_k(a, m, j - 1), ksw = sq_k(a, m - 1, j - 1), kne = sq_k(a, m, j), kse = sq_k(a, m - 1, j);
        // helmFE_var: k of the square(s) an entry belongs to; local_rect: the same formulas with k2 + i eps and eta
        auto edge = [&](double k) {          // -0.5 - z h2 / 24 - i kappa h / 6      (helmFE_var.py:147-293, p_h:1534-1600)
            const Cx z = zterm(a, k * k, 1.0);
            const double kap = a.variable ? k : a.eta;
            return cx(-0.5, 0.0) - z * h2 / 24. - cx(0.0, kap * h / 6.);
        };
        auto diagnb = [&](double k) {        // -z h2 / 12
            return neg(zterm(a, k * k, 1.0)) * h2 / 12.;
        };
        auto cross = [&](double ka, double kb) {   // -1 - cr (ka^2 + kb^2) h2 / 24   |   -1 - (k2 + i eps) h2 / 12
            if (a.variable) return cx(-1.0, 0.0) - zterm(a, ka * ka + kb * kb, 0.0) * h2 / 24.;
            return cx(-1.0, 0.0) - zterm(a, 0.0, 1.0) * h2 / 12.;
        };
        Cx dg;
        if (a.variable) {
            if (bot && left) dg = real_minus(1., zterm(a, kne * kne, 0.) * h2 / 6.) - cx(0.0, kne * 2 * h / 3.);
            else if (bot && right) dg = real_minus(1., zterm(a, knw * knw, 0.) * h2 / 12.) - cx(0.0, knw * 2. * h / 3.);
            else if (top && left) dg = real_minus(1., zterm(a, kse * kse, 0.) * h2 / 12.) - cx(0.0, kse * 2. * h / 3.);
            else if (top && right) dg = real_minus(1., zterm(a, ksw * ksw, 0.) * (h2 / 6.)) - cx(0.0, ksw * 2. * h / 3.);
            else if (bot) dg = real_minus(2., zterm(a, knw * knw + 2. * (kne * kne), 0.) * h2 / 12.) - cx(0.0, (knw + kne) * h / 3.);
            else if (top) dg = real_minus(2., zterm(a, 2. * (ksw * ksw) + kse * kse, 0.) * h2 / 12.) - cx(0.0, (ksw + kse) * h / 3.);
            else if (left) dg = real_minus(2., zterm(a, 2. * (kne * kne) + kse * kse, 0.) * h2 / 12.) - cx(0.0, (kne + kse) * h / 3.);
            else if (right) dg = real_minus(2., zterm(a, knw * knw + 2. * (ksw * ksw), 0.) * h2 / 12.) - cx(0.0, (knw + ksw) * h / 3.);
            else dg = real_minus(4., zterm(a, knw * knw + 2. * (ksw * ksw) + 2. * (kne * kne) + kse * kse, 0.) * h2 / 12.);
        } else {
            const Cx z = zterm(a, 0.0, 1.0);
            if ((bot && left) || (top && right)) dg = real_minus(1., z * h2 / 6.) - cx(0.0, a.eta * 2 * h / 3.);
            else if ((bot && right) || (top && left)) dg = real_minus(1., z * h2 / 12.) - cx(0.0, a.eta * 2 * h / 3.);
            else if (bot || top || left || right) dg = real_minus(2., z * h2 / 4.) - cx(0.0, 2. * a.eta * h / 3.);
            else dg = real_minus(4., z * h2 / 2.);
        }
        auto put = [&](long long col, Cx v) { cols[p] = (int)col; vals[p] = to_val<T>(v); ++p; };
        const int Nh = a.Nh;
        // columns ascending: row - Nh - 1, row - Nh, row - 1, row, row + 1, row + Nh, row + Nh + 1
        if (!bot && !left) put(row - Nh - 1, diagnb(ksw));                                     // all rows with a south-west square
        if (!bot) {
            if (left) put(row - Nh, edge(kse));
            else if (right) put(row - Nh, edge(ksw));
            else put(row - Nh, top ? cross(ksw, kse) : cross(ksw, kse));
        }
        if (!left) {
            if (bot) put(row - 1, edge(knw));
            else if (top) put(row - 1, edge(ksw));
            else put(row - 1, right ? cross(ksw, knw) : cross(knw, ksw));
        }
        put(row, dg);
        if (!right) {
            if (bot) put(row + 1, edge(kne));
            else if (top) put(row + 1, edge(kse));
            else put(row + 1, left ? cross(kse, kne) : cross(kne, kse));
        }
        if (!top) {
            if (left) put(row + Nh, edge(kne));
            else if (right) put(row + Nh, edge(knw));
            else put(row + Nh, cross(knw, kne));
        }
        if (!top && !right) put(row + Nh + 1, diagnb(kne));
    }
}

}  // namespace

long long helm_fe_nnz(int Nh, int Nv) { return 2 * (5LL * Nh - 3) + (long long)(Nv - 2) * (7LL * Nh - 4); }

int launch_gen_helm_fe(int dtype, int variable, int N, double p0, double p1, double p2, double L, const double *C_dev, int Nh, int Nv, void *vals,
                       int *ptr, int *cols, hipStream_t st) {
    FeArgs a;
    a.Nh = Nh; a.Nv = Nv; a.variable = variable;
    a.h = (variable ? 1.0 : L * 1.0) / (N - 1.0);        // helmFE_var.py:47 / p_h-PY_C-CL.py:1474
    a.h2 = a.h * a.h;
    a.omega = p0; a.rho = p1;
    a.k2 = p0 * p0; a.eps = p1; a.eta = p2;
    a.C = C_dev;
    const long long nn = (long long)Nh * Nv;
    const int grid = (int)std::min<long long>((nn + 256) / 256, 65536);
    if (dtype == 2) hipLaunchKernelGGL((gen_helm_fe_kernel<float2>), dim3(grid), dim3(256), 0, st, a, static_cast<float2 *>(vals), ptr, cols);
    else if (dtype == 3) hipLaunchKernelGGL((gen_helm_fe_kernel<double2>), dim3(grid), dim3(256), 0, st, a, static_cast<double2 *>(vals), ptr, cols);
    else return fail(CGAMD_ERR_INVALID, "gen_helm_fe: the matrix is complex (dtype complex64 or complex128)");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("gen_helm_fe launch: ") + hipGetErrorString(e));
    return CGAMD_OK;
}

}  // namespace cgamd

// =================================================================================================
// Stencil generators (device side, so multi-GB systems never cross PCIe): SURVEY 8(d) "M" / "C5" and reference Poisson()
// =================================================================================================
namespace cgamd {

__host__ __device__ inline long long lap3d_ptr(long long i, long long nx, long long ny, long long nz) {
    // entries stored before row i = 7 i - (missing neighbours of rows < i), closed form
    const long long pl = nx * ny, n = pl * nz;
    const long long x0 = (i + nx - 1) / nx;                       // rows j<i with ix == 0
    const long long x1 = i / nx;                                  // ix == nx-1
    const long long full = i / pl, rem = i % pl;
    const long long y0 = full * nx + (rem < nx ? rem : nx);       // iy == 0
    const long long y1 = full * nx + (rem > pl - nx ? rem - (pl - nx) : 0);  // iy == ny-1
    const long long z0 = i < pl ? i : pl;                         // iz == 0
    const long long z1 = i > n - pl ? i - (n - pl) : 0;           // iz == nz-1
    return 7 * i - (x0 + x1 + y0 + y1 + z0 + z1);
}
long long laplace3d_ptr(long long i, int nx, int ny, int nz) { return lap3d_ptr(i, nx, ny, nz); }

template <typename T> CG_DEV T real_val(double v);
template <> CG_DEV float real_val<float>(double v) { return (float)v; }
template <> CG_DEV double real_val<double>(double v) { return v; }
template <> CG_DEV float2 real_val<float2>(double v) { return make_float2((float)v, 0.f); }
template <> CG_DEV double2 real_val<double2>(double v) { return make_double2(v, 0.); }

template <typename T>
__global__ void gen_laplace3d_kernel(int nx, int ny, int nz, long long row_begin, long long row_end, T *vals, int *ptr,
                                     int *cols) {
    const long long nloc = row_end - row_begin;
    const long long base = lap3d_ptr(row_begin, nx, ny, nz);
    for (long long li = (long long)blockIdx.x * blockDim.x + threadIdx.x; li <= nloc;
         li += (long long)gridDim.x * blockDim.x) {
        const long long i = row_begin + li;
        long long p = lap3d_ptr(i, nx, ny, nz) - base;
        ptr[li] = (int)p;
        if (li == nloc) break;
        const long long pl = (long long)nx * ny;
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / pl);
        if (iz > 0) { cols[p] = (int)(i - pl); vals[p++] = real_val<T>(-1.0); }
        if (iy > 0) { cols[p] = (int)(i - nx); vals[p++] = real_val<T>(-1.0); }
        if (ix > 0) { cols[p] = (int)(i - 1); vals[p++] = real_val<T>(-1.0); }
        cols[p] = (int)i; vals[p++] = real_val<T>(6.0);
        if (ix < nx - 1) { cols[p] = (int)(i + 1); vals[p++] = real_val<T>(-1.0); }
        if (iy < ny - 1) { cols[p] = (int)(i + nx); vals[p++] = real_val<T>(-1.0); }
        if (iz < nz - 1) { cols[p] = (int)(i + pl); vals[p++] = real_val<T>(-1.0); }
    }
}

__host__ __device__ inline long long poi2d_ptr(long long i, long long N) {
    const long long n = N * N;
    const long long x0 = (i + N - 1) / N, x1 = i / N;
    const long long y0 = i < N ? i : N, y1 = i > n - N ? i - (n - N) : 0;
    return 5 * i - (x0 + x1 + y0 + y1);
}
long long poisson2d_ptr(long long i, int N) { return poi2d_ptr(i, N); }

template <typename T> __global__ void gen_poisson2d_kernel(int N, T *vals, int *ptr, int *cols) {
    const long long n = (long long)N * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x) {
        long long p = poi2d_ptr(i, N);
        ptr[i] = (int)p;
        if (i == n) break;
        const int jx = (int)(i % N), iy = (int)(i / N);
        if (iy > 0) { cols[p] = (int)(i - N); vals[p++] = real_val<T>(-1.0); }
        if (jx > 0) { cols[p] = (int)(i - 1); vals[p++] = real_val<T>(-1.0); }
        cols[p] = (int)i; vals[p++] = real_val<T>(4.0);
        if (jx < N - 1) { cols[p] = (int)(i + 1); vals[p++] = real_val<T>(-1.0); }
        if (iy < N - 1) { cols[p] = (int)(i + N); vals[p++] = real_val<T>(-1.0); }
    }
}

template <typename T>
static int gen3d_impl(int nx, int ny, int nz, long long rb, long long re, void *vals, int *ptr, int *cols, hipStream_t st) {
    const long long nloc = re - rb + 1;
    int g = (int)((nloc + 255) / 256 < 8192 ? (nloc + 255) / 256 : 8192);
    hipLaunchKernelGGL((gen_laplace3d_kernel<T>), dim3(g), dim3(256), 0, st, nx, ny, nz, rb, re, (T *)vals, ptr, cols);
    return check_launch("gen_laplace3d");
}
int launch_gen_laplace3d(int dtype, int nx, int ny, int nz, long long row_begin, long long row_end, void *vals, int *ptr,
                         int *cols, hipStream_t st) {
    CG_DISPATCH(dtype, gen3d_impl, nx, ny, nz, row_begin, row_end, vals, ptr, cols, st);
}
template <typename T> static int gen2d_impl(int N, void *vals, int *ptr, int *cols, hipStream_t st) {
    const long long n = (long long)N * N + 1;
    int g = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL((gen_poisson2d_kernel<T>), dim3(g), dim3(256), 0, st, N, (T *)vals, ptr, cols);
    return check_launch("gen_poisson2d");
}
int launch_gen_poisson2d(int dtype, int N, void *vals, int *ptr, int *cols, hipStream_t st) {
    CG_DISPATCH(dtype, gen2d_impl, N, vals, ptr, cols, st);
}

}  // namespace cgamd

// =================================================================================================
// Right-hand sides of the reference's Helmholtz drivers on an N x N node grid (reference helmFE_var.py:333-389), b[row][col] at
// row N + col:   kind 0  rhs(N, k)   plane-wave impedance data on the boundary nodes (:333-368), evaluated node by node in the
//                                    reference's operation order -- including its quirk that the RIGHT boundary is integrated over the
//                                    TOP boundary's points (:355-356) -- in complex double, then rounded to the requested type;
//                kind 1  rhsL(N, k)  k^2 on the left boundary, corners excluded (:370-377);
//                kind 2  rhsA(N, k)  k^2 on the four boundary lines (:379-389) -- BASELINE config 3's right-hand side rhsA(500, 12).
// =================================================================================================
namespace cgamd {
namespace {

struct Cd { double re, im; };
CG_DEV Cd cmul(Cd a, Cd b) { return Cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
CG_DEV Cd cadd(Cd a, Cd b) { return Cd{a.re + b.re, a.im + b.im}; }
CG_DEV Cd cscale(double s, Cd a) { return Cd{s * a.re - 0.0 * a.im, s * a.im + 0.0 * a.re}; }      // (s + 0i) a as numpy forms it
// exp(1. * 1.j * k * (p . a)) for a = (1, 1) / sqrt(2): the argument is purely imaginary
CG_DEV Cd wave(double k, double a0, double a1, double px, double py) {
    const double s = px * a0 + py * a1;
    const double th = k * s;
    return Cd{cos(th), sin(th)};
}
template <typename T> CG_DEV void put_c(T *b, long long i, Cd v);
template <> CG_DEV void put_c<float2>(float2 *b, long long i, Cd v) { b[i] = make_float2((float)v.re, (float)v.im); }
template <> CG_DEV void put_c<double2>(double2 *b, long long i, Cd v) { b[i] = make_double2(v.re, v.im); }
template <> CG_DEV void put_c<float>(float *b, long long i, Cd v) { b[i] = (float)v.re; }
template <> CG_DEV void put_c<double>(double *b, long long i, Cd v) { b[i] = v.re; }

template <typename T> __global__ void gen_rhs_kernel(int kind, int N, double k, T *b) {
    const long long n = (long long)N * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / N), col = (int)(i % N);
        Cd v{0.0, 0.0};
        if (kind == 2) {
            if (row == 0 || row == N - 1 || col == 0 || col == N - 1) v.re = k * k;
        } else if (kind == 1) {
            if (col == 0 && row >= 1 && row <= N - 2) v.re = k * k;
        } else if (row == 0 || row == N - 1 || col == 0 || col == N - 1) {
            const double a0 = 1. / sqrt(2.), a1 = 1. / sqrt(2.);
            const double h = 1. / (N - 1.), step = 1.0 / (N - 1);
            auto X = [&](int j) { return 0.0 + j * step; };                 // numpy.arange(0.0, 1.00001, 1.0 / (N - 1))
            auto Y = [&](int j) { return (X(j + 1) + X(j)) / 2.0; };         // mid points
            // multipliers i k (a.n - 1) on each side: 1.j * k * (real) = (0 + k i)(r + 0 i)
            auto mult = [&](double r) { return Cd{0.0 * r - k * 0.0, 0.0 * 0.0 + k * r}; };
            const Cd multbot = mult(-a1 - 1.), multtop = mult(a1 - 1.), multleft = mult(-a0 - 1.), multright = mult(a0 - 1.);
            auto side = [&](Cd m, double p0x, double p0y, double p1x, double p1y, double p2x, double p2y) {
                const Cd sum = cadd(cadd(wave(k, a0, a1, p0x, p0y), wave(k, a0, a1, p1x, p1y)), wave(k, a0, a1, p2x, p2y));
                return cmul(cscale(h / 3., m), sum);
            };
            auto corner = [&](Cd m, double p0x, double p0y, double p1x, double p1y) {      // (h/6) m (2 e(p0) + e(p1))
                const Cd two{2.0, 0.0};
                const Cd e0 = wave(k, a0, a1, p0x, p0y), e1 = wave(k, a0, a1, p1x, p1y);
                return cmul(cscale(h / 6., m), cadd(Cd{2.0 * e0.re - 0.0 * e0.im, 2.0 * e0.im + 0.0 * e0.re}, e1));
                (void)two;
            };
            const bool cr = (row == 0 || row == N - 1) && (col == 0 || col == N - 1);
            if (!cr) {
                // the reference assigns bottom, top, left, right in this order inside one loop over j: for a non-corner node exactly one applies
                if (row == 0) { const int j = col; v = side(multbot, Y(j - 1), 0., X(j), 0., Y(j), 0.); }
                else if (row == N - 1) { const int j = col; v = side(multtop, Y(j - 1), 1., X(j), 1., Y(j), 1.); }
                else if (col == 0) { const int j = row; v = side(multleft, 0., Y(j - 1), 0., X(j), 0., Y(j)); }
                else { const int j = row; v = side(multright, Y(j - 1), 1., X(j), 1., Y(j), 1.); }     // (sic: the top boundary's points)
            } else if (row == 0 && col == 0) {
                v = cadd(corner(multleft, 0., Y(0), 0., 0.), corner(multbot, Y(0), 0., 0., 0.));
            } else if (row == 0 && col == N - 1) {
                v = cadd(corner(multbot, Y(N - 2), 0., 1., 0.), corner(multright, 1., Y(0), 1., 0.));
            } else if (row == N - 1 && col == 0) {
                v = cadd(corner(multleft, 0., Y(N - 2), 0., 1.), corner(multtop, Y(0), 1., 0., 1.));
            } else {
                v = cadd(corner(multtop, Y(N - 2), 1., 1., 1.), corner(multright, 1., Y(N - 2), 1., 1.));
            }
        }
        put_c<T>(b, i, v);
    }
}

template <typename T> int gen_rhs_impl(int kind, int N, double k, void *b, hipStream_t st) {
    const long long n = (long long)N * N;
    const int g = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL((gen_rhs_kernel<T>), dim3(g), dim3(256), 0, st, kind, N, k, static_cast<T *>(b));
    return check_launch("gen_rhs");
}

}  // namespace

int launch_gen_rhs(int dtype, int kind, int N, double k, void *b, hipStream_t st) {
    CG_DISPATCH(dtype, gen_rhs_impl, kind, N, k, b, st);
}

}  // namespace cgamd
