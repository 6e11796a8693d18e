// Streaming vector kernels and scalar steps of the CG hot path, hand-written for gfx950.
//
//   dot_partials_kernel           reference kernel/{real,complex}/vdot.cl (partials stay on the device)
//   ewise_kernel                  reference kernel/{real,complex}/{axpy,aypx,sub}.cl
//   axpy_dot / aypx_beta_x        the fused loop: r -= alpha q + r.r partials; beta, x += alpha d, d = beta d + r
//                                 (reference clcg.c:338-416); axpy2_dot / aypx_beta: the form with x updated in the r launch
//   pcg_*                         diagonally preconditioned recurrence (reference helmFE_var.py:546-586)
//   cg_alpha/beta/delta0          the scalar work the reference does on the host (clcg.c:274-292,317-334,376-411)
//
// grid-stride work-groups of 256 threads (<= 2048; streaming single-RHS systems: 512 = two per CU), 16 B per lane.  Reductions
// are wave64 shuffles, then LDS across the 4 waves, then a fixed-order pass over the per-work-group partials: bitwise
// reproducible run to run (atomics only hand out tickets).
#include "cgamd_internal.h"
#include "device_types.h"
#include "device_mem.h"
#include "spmv_device.h"
#include "reduce_device.h"
#include "launch_util.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

namespace cgamd {

// =================================================================================================
// Streaming vector kernels.  grid = (G, nRHS); RHS r lives at base + r*ld.
// =================================================================================================
// x += alpha d ; r -= alpha q ; partial(r.r)
template <typename T, int BLOCK, bool VEC, int VNT>
CG_DEV void axpy2_dot_body(int n, const T *__restrict__ d, T *__restrict__ x, const T *__restrict__ q, T *__restrict__ rv,
                           long long ld, T al, typename VT<T>::acc *__restrict__ partials, typename VT<T>::acc *red) {
    using A = typename VT<T>::acc;
    const int r = blockIdx.y;
    const long long off = (long long)r * ld;
    d += off; x += off; q += off; rv += off;
    A acc = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pd = ld_pack(d + i * E), pq = (VNT & 2) ? ld_pack_nt(q + i * E) : ld_pack(q + i * E);
            Pack<T> px = (VNT & 1) ? ld_pack_nt(x + i * E) : ld_pack(x + i * E), pr = ld_pack(rv + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                px.v[k] = vadd(px.v[k], vmul(al, pd.v[k]));
                pr.v[k] = vsub(pr.v[k], vmul(al, pq.v[k]));
                acc = vadd(acc, to_acc(vmul(pr.v[k], pr.v[k])));
            }
            if (VNT & 1) st_pack_nt(x + i * E, px); else st_pack(x + i * E, px);
            st_pack(rv + i * E, pr);
        }
        i0 += npack * E;  // scalar tail
    }
    for (long long i = i0; i < n; i += stride) {
        x[i] = vadd(x[i], vmul(al, d[i]));
        const T rn = vsub(rv[i], vmul(al, q[i]));
        rv[i] = rn;
        acc = vadd(acc, to_acc(vmul(rn, rn)));
    }
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) partials[(long long)r * gridDim.x + blockIdx.x] = tot;
}

template <typename T, int BLOCK, bool VEC, int VNT = 0>
__global__ __launch_bounds__(BLOCK) void axpy2_dot_kernel(int n, const T *__restrict__ d, T *__restrict__ x,
                                                          const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                          const T *__restrict__ alpha,
                                                          typename VT<T>::acc *__restrict__ partials) {
    __shared__ typename VT<T>::acc red[BLOCK / kWave];
    axpy2_dot_body<T, BLOCK, VEC, VNT>(n, d, x, q, rv, ld, alpha[blockIdx.y], partials, red);
}

// Three-launch iteration for small systems (at most kFoldAlphaMax d.q partials per RHS): alpha is computed in the
// prologue of this launch -- every work-group adds the SpMV's d.q partials in the same fixed order, so all hold the
// bit-identical alpha = delta / d.q (clcg.c:317-327); work-group 0 records alpha and advances the iteration counter
// (nothing else in this launch reads either).  Saves the cg_alpha launch: 18.8 -> ~14 us per iteration at 250k rows.
constexpr int kFoldAlphaMax = 2048;     // N <= 524k rows; beyond, the separate cg_alpha launch is cheaper than every work-group summing
template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void axpy2_dot_alpha_kernel(int n, const T *__restrict__ d, T *__restrict__ x,
                                                                const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                                const typename VT<T>::acc *__restrict__ part_dq, int P, int K,
                                                                const T *__restrict__ delta, T *alpha, int *iter,
                                                                typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T alpha_s;
    const int r = blockIdx.y;
    {
        const A acc = thread_partials<BLOCK>(part_dq + (long long)r * P, P, K);
        const A dq = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const T dqT = from_acc<T>(dq);      // the reference rounds d.q to the value type before dividing (clcg.c:318-327)
            const T al = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
            alpha_s = al;
            if (blockIdx.x == 0) {
                alpha[r] = al;
                if (r == 0) *iter = *iter + 1;
            }
        }
        __syncthreads();
    }
    axpy2_dot_body<T, BLOCK, VEC, 0>(n, d, x, q, rv, ld, alpha_s, partials, red);
}

// ---- ten-vector-pass iteration -------------------------------------------------------------------------------
// x is read by nothing inside the loop, so x += alpha d may ride in the aypx launch, which reads d anyway:
//   axpy_dot_kernel      r -= alpha q, partials of r.r                      reads q, r    writes r      3 NV
//   aypx_beta_x_kernel   beta in the prologue; x += alpha d; d = beta d + r reads r, d, x writes d, x   5 NV
// instead of 6 NV + 3 NV: d is read once per iteration, not twice (fused minimum 10 NV + SpMV).  Every element sees the
// same operations in the same order as before, so x is bit-identical.
template <typename T, int BLOCK, bool VEC, int VNT>
CG_DEV void axpy_dot_body(int n, const T *__restrict__ q, T *__restrict__ rv, long long ld, T al,
                          typename VT<T>::acc *__restrict__ partials, typename VT<T>::acc *red) {
    using A = typename VT<T>::acc;
    const int r = blockIdx.y;
    q += (long long)r * ld; rv += (long long)r * ld;
    A acc = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pq = (VNT & 2) ? ld_pack_nt(q + i * E) : ld_pack(q + i * E);
            Pack<T> pr = ld_pack(rv + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                pr.v[k] = vsub(pr.v[k], vmul(al, pq.v[k]));
                acc = vadd(acc, to_acc(vmul(pr.v[k], pr.v[k])));
            }
            st_pack(rv + i * E, pr);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T rn = vsub(rv[i], vmul(al, q[i]));
        rv[i] = rn;
        acc = vadd(acc, to_acc(vmul(rn, rn)));
    }
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) partials[(long long)r * gridDim.x + blockIdx.x] = tot;
}
template <typename T, int BLOCK, bool VEC, int VNT = 0>
__global__ __launch_bounds__(BLOCK) void axpy_dot_kernel(int n, const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                         const T *__restrict__ alpha, typename VT<T>::acc *__restrict__ partials) {
    __shared__ typename VT<T>::acc red[BLOCK / kWave];
    axpy_dot_body<T, BLOCK, VEC, VNT>(n, q, rv, ld, alpha[blockIdx.y], partials, red);
}
// small systems: alpha in the prologue (see axpy2_dot_alpha_kernel)
template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void axpy_dot_alpha_kernel(int n, const T *__restrict__ q, T *__restrict__ rv, long long ld,
                                                               const typename VT<T>::acc *__restrict__ part_dq, int P, int K,
                                                               const T *__restrict__ delta, T *alpha, int *iter,
                                                               typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T alpha_s;
    const int r = blockIdx.y;
    {
        const A acc = thread_partials<BLOCK>(part_dq + (long long)r * P, P, K);
        const A dq = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const T dqT = from_acc<T>(dq);
            const T al = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
            alpha_s = al;
            if (blockIdx.x == 0) {
                alpha[r] = al;
                if (r == 0) *iter = *iter + 1;
            }
        }
        __syncthreads();
    }
    axpy_dot_body<T, BLOCK, VEC, 0>(n, q, rv, ld, alpha_s, partials, red);
}

template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void dot_partials_kernel(int n, const T *__restrict__ a, const T *__restrict__ b,
                                                             long long ld, typename VT<T>::acc *__restrict__ partials) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    const int r = blockIdx.y;
    a += (long long)r * ld; b += (long long)r * ld;
    A acc = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pa = ld_pack(a + i * E), pb = ld_pack(b + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) acc = vadd(acc, to_acc(vmul(pa.v[k], pb.v[k])));
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) acc = vadd(acc, to_acc(vmul(a[i], b[i])));
    const A tot = block_sum<BLOCK>(acc, red);
    if (threadIdx.x == 0) partials[(long long)r * gridDim.x + blockIdx.x] = tot;
}

// OP 0: y += a x   1: y -= a x   2: y = a y + x   3: res(y) = x - b
template <typename T, int BLOCK, bool VEC, int OP>
__global__ __launch_bounds__(BLOCK) void ewise_kernel(int n, const T *__restrict__ x, T *__restrict__ y,
                                                      const T *__restrict__ b2, long long ld,
                                                      const T *__restrict__ alpha) {
    const int r = blockIdx.y;
    x += (long long)r * ld; y += (long long)r * ld;
    if (OP == 3) b2 += (long long)r * ld;
    const T al = (OP == 3) ? vzero<T>() : alpha[r];
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    auto f = [&](T xv, T yv, T bv) -> T {
        if (OP == 0) return vadd(yv, vmul(al, xv));
        if (OP == 1) return vsub(yv, vmul(al, xv));
        if (OP == 2) return vaypx(al, yv, xv);
        return vsub(xv, bv);
    };
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py, pb;
            if (OP != 3) py = ld_pack(y + i * E);
            if (OP == 3) pb = ld_pack(b2 + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) py.v[k] = f(px.v[k], OP != 3 ? py.v[k] : vzero<T>(), OP == 3 ? pb.v[k] : vzero<T>());
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride)
        y[i] = f(x[i], OP != 3 ? y[i] : vzero<T>(), OP == 3 ? b2[i] : vzero<T>());
}

// d = beta d + r with beta computed in the prologue (replaces the cg_beta launch of the 5-launch loop):
// every work-group adds the P partials of r.r in the same fixed order (thread-strided, wave tree, 4 wave
// sums), so all of them hold the bit-identical delta_new and beta = delta_new / delta_old
// (clcg.c:376-391); delta_old is history[iter-1] -- nothing in this launch writes that entry, work-group 0
// alone writes delta/beta/history[iter].  The iteration counter was advanced by cg_alpha.
template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void aypx_beta_kernel(int n, const T *__restrict__ x, T *__restrict__ y, long long ld,
                                                          const typename VT<T>::acc *__restrict__ partials, int P, int K,
                                                          int nrhs, T *delta, T *beta, T *history, int history_cap, const int *iter) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int r = blockIdx.y;
    {
        const A acc = thread_partials<BLOCK>(partials + (long long)r * P, P, K);
        const A tot = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const int it = *iter;
            const T dnT = from_acc<T>(tot);
            const T dold = history[(long long)(it - 1) * nrhs + r];
            const T b = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
            beta_s = b;
            if (blockIdx.x == 0) {
                beta[r] = b;
                delta[r] = dnT;
                if (it < history_cap) history[(long long)it * nrhs + r] = dnT;
            }
        }
        __syncthreads();
    }
    const T al = beta_s;
    x += (long long)r * ld; y += (long long)r * ld;
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py = ld_pack(y + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) py.v[k] = vaypx(al, py.v[k], px.v[k]);
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) y[i] = vaypx(al, y[i], x[i]);
}

// the same with the deferred x += alpha d (ten-vector-pass iteration): xs = solution vector, alpha of THIS iteration
template <typename T, int BLOCK, bool VEC, int VNT>
__global__ __launch_bounds__(BLOCK) void aypx_beta_x_kernel(int n, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ xs,
                                                            long long ld, const typename VT<T>::acc *__restrict__ partials, int P, int K,
                                                            int nrhs, const T *__restrict__ alpha, T *delta, T *beta, T *history,
                                                            int history_cap, const int *iter) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int r = blockIdx.y;
    {
        const A acc = thread_partials<BLOCK>(partials + (long long)r * P, P, K);
        const A tot = block_sum<BLOCK>(acc, red);
        if (threadIdx.x == 0) {
            const int it = *iter;
            const T dnT = from_acc<T>(tot);
            const T dold = history[(long long)(it - 1) * nrhs + r];
            const T b = from_acc<T>(acc_div(to_acc(dnT), to_acc(dold)));
            beta_s = b;
            if (blockIdx.x == 0) {
                beta[r] = b;
                delta[r] = dnT;
                if (it < history_cap) history[(long long)it * nrhs + r] = dnT;
            }
        }
        __syncthreads();
    }
    const T bt = beta_s, al = alpha[r];
    x += (long long)r * ld; y += (long long)r * ld; xs += (long long)r * ld;
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> px = ld_pack(x + i * E);
            Pack<T> py = ld_pack(y + i * E);
            Pack<T> ps = (VNT & 1) ? ld_pack_nt(xs + i * E) : ld_pack(xs + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                ps.v[k] = vadd(ps.v[k], vmul(al, py.v[k]));
                py.v[k] = vaypx(bt, py.v[k], px.v[k]);
            }
            if (VNT & 1) st_pack_nt(xs + i * E, ps); else st_pack(xs + i * E, ps);
            st_pack(y + i * E, py);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T dv = y[i];
        xs[i] = vadd(xs[i], vmul(al, dv));
        y[i] = vaypx(bt, dv, x[i]);
    }
}

// =================================================================================================
// Diagonally (Jacobi) preconditioned CG -- the reference's PCG with a diagonal CSR `M`, z = M.dot(r)
// (helmFE_var.py:546-586): rho = r.z, p = z + (rho/rho_old) p, q = A p, alpha = rho / p.q, x += alpha p, r -= alpha q,
// stop on sqrt|r.r|.  Same four launches as the plain loop: the SpMV (+p.q) and cg_alpha are shared (delta holds rho);
//   pcg_axpy2_dot2_kernel : r -= alpha q, partials of r.(m r) and of r.r                    (4NV bytes)
//   pcg_aypx_beta_kernel  : beta in the prologue, x += alpha p, p = m r + beta p            (6NV bytes)
// (x += alpha p rides in the second launch, which reads p anyway: see the ten-vector-pass iteration above)
// m[i] is what multiplies r[i] (the inverse diagonal for Jacobi), shared by all right-hand sides.  rho of the previous
// iteration is read from a two-entry parity buffer so that work-group 0 may publish the new one in the same launch.
// =================================================================================================
template <typename T, int BLOCK, bool VEC, bool INIT>
__global__ __launch_bounds__(BLOCK) void pcg_axpy2_dot2_kernel(int n, const T *__restrict__ d, T *__restrict__ x,
                                                               const T *__restrict__ q, T *__restrict__ rv,
                                                               const T *__restrict__ m, long long ld,
                                                               const T *__restrict__ alpha,
                                                               typename VT<T>::acc *__restrict__ part_rz,
                                                               typename VT<T>::acc *__restrict__ part_rr) {
    // INIT: no update, d = m r instead (set_rhs: p0 = z0), same two dot products
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    const int r = blockIdx.y;
    T *dw = const_cast<T *>(d) + (long long)r * ld;
    d += (long long)r * ld; x += (long long)r * ld; q += (long long)r * ld; rv += (long long)r * ld;
    const T al = INIT ? vzero<T>() : alpha[r];
    A arz = vzero<A>(), arr = vzero<A>();
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            Pack<T> pr = ld_pack(rv + i * E);
            const Pack<T> pm = ld_pack(m + i * E);
            if (!INIT) {
                const Pack<T> pq = ld_pack(q + i * E);
#pragma unroll
                for (int k = 0; k < E; ++k) pr.v[k] = vsub(pr.v[k], vmul(al, pq.v[k]));
                st_pack(rv + i * E, pr);
            }
            Pack<T> pz;
#pragma unroll
            for (int k = 0; k < E; ++k) {
                pz.v[k] = vmul(pm.v[k], pr.v[k]);
                arz = vadd(arz, to_acc(vmul(pr.v[k], pz.v[k])));
                arr = vadd(arr, to_acc(vmul(pr.v[k], pr.v[k])));
            }
            if (INIT) st_pack(dw + i * E, pz);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        T rn = rv[i];
        if (!INIT) {
            rn = vsub(rn, vmul(al, q[i]));
            rv[i] = rn;
        }
        const T z = vmul(m[i], rn);
        if (INIT) dw[i] = z;
        arz = vadd(arz, to_acc(vmul(rn, z)));
        arr = vadd(arr, to_acc(vmul(rn, rn)));
    }
    const A trz = block_sum<BLOCK>(arz, red);
    if (threadIdx.x == 0) part_rz[(long long)r * gridDim.x + blockIdx.x] = trz;
    const A trr = block_sum<BLOCK>(arr, red);
    if (threadIdx.x == 0) part_rr[(long long)r * gridDim.x + blockIdx.x] = trr;
}

template <typename T, int BLOCK, bool VEC>
__global__ __launch_bounds__(BLOCK) void pcg_aypx_beta_kernel(int n, const T *__restrict__ rv, T *__restrict__ pv,
                                                              const T *__restrict__ m, long long ld,
                                                              const typename VT<T>::acc *__restrict__ part_rz,
                                                              const typename VT<T>::acc *__restrict__ part_rr, int P, int K, int nrhs,
                                                              T *delta, T *beta, T *history, int history_cap, T *rho2, const int *iter,
                                                              T *__restrict__ xs, const T *__restrict__ alpha) {
    using A = typename VT<T>::acc;
    __shared__ A red[BLOCK / kWave];
    __shared__ T beta_s;
    const int r = blockIdx.y;
    {
        const A acc = thread_partials<BLOCK>(part_rz + (long long)r * P, P, K);
        const A rho = block_sum<BLOCK>(acc, red);
        A acc2 = vzero<A>();
        if (blockIdx.x == 0) {
            acc2 = thread_partials<BLOCK>(part_rr + (long long)r * P, P, K);
            acc2 = block_sum<BLOCK>(acc2, red);
        }
        if (threadIdx.x == 0) {
            const int it = *iter;
            const T rhoT = from_acc<T>(rho);
            const T rold = rho2[(long long)((it - 1) & 1) * nrhs + r];
            const T b = from_acc<T>(acc_div(to_acc(rhoT), to_acc(rold)));
            beta_s = b;
            if (blockIdx.x == 0) {
                beta[r] = b;
                delta[r] = rhoT;                                   // cg_alpha divides this by p.q
                rho2[(long long)(it & 1) * nrhs + r] = rhoT;
                if (it < history_cap) history[(long long)it * nrhs + r] = from_acc<T>(acc2);   // r.r: what the stopping test looks at
            }
        }
        __syncthreads();
    }
    const T bt = beta_s, al = alpha[r];
    rv += (long long)r * ld; pv += (long long)r * ld; xs += (long long)r * ld;
    constexpr int E = Pack<T>::N;
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (VEC) {
        const long long npack = n / E;
        for (long long i = i0; i < npack; i += stride) {
            const Pack<T> pr = ld_pack(rv + i * E), pm = ld_pack(m + i * E);
            Pack<T> pp = ld_pack(pv + i * E), px = ld_pack(xs + i * E);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                px.v[k] = vadd(px.v[k], vmul(al, pp.v[k]));
                pp.v[k] = vadd(vmul(bt, pp.v[k]), vmul(pm.v[k], pr.v[k]));
            }
            st_pack(xs + i * E, px);
            st_pack(pv + i * E, pp);
        }
        i0 += npack * E;
    }
    for (long long i = i0; i < n; i += stride) {
        const T pv0 = pv[i];
        xs[i] = vadd(xs[i], vmul(al, pv0));
        pv[i] = vadd(vmul(bt, pv0), vmul(m[i], rv[i]));
    }
}

// set_rhs: delta = rho0 = sum r.z partials, rho2[0] = rho0, history[0] = r.r, iter = 0
template <typename T>
__global__ __launch_bounds__(1024) void pcg_delta0_kernel(const typename VT<T>::acc *part_rz, const typename VT<T>::acc *part_rr, int P,
                                                          int nrhs, T *delta, T *history, T *rho2, int *iter);



template <typename T>
__global__ __launch_bounds__(kScalarBlock) void reduce_to_value_kernel(const typename VT<T>::acc *partials, int grid, int nrhs,
                                                              T *result) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto s = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) result[r] = from_acc<T>(s);
}

template <typename T>
__global__ __launch_bounds__(1024) void pcg_delta0_kernel(const typename VT<T>::acc *part_rz, const typename VT<T>::acc *part_rr, int P,
                                                          int nrhs, T *delta, T *history, T *rho2, int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto rho = sum_partials_block(part_rz + (long long)r * P, P, smem);
    __syncthreads();
    const auto rr = sum_partials_block(part_rr + (long long)r * P, P, smem);
    if (threadIdx.x == 0) {
        delta[r] = from_acc<T>(rho);
        rho2[r] = from_acc<T>(rho);
        history[r] = from_acc<T>(rr);
        if (r == 0) *iter = 0;
    }
}

template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_delta0_kernel(const typename VT<T>::acc *partials, int grid, int nrhs, T *delta,
                                                        T *history, int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto s = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) {
        delta[r] = from_acc<T>(s);
        history[r] = from_acc<T>(s);
        if (r == 0) *iter = 0;
    }
}

// alpha[r] = delta[r] / (d.q)[r].  Block 0 also advances the iteration counter: the counter is only READ by
// the cg_beta kernel of the same iteration (a later launch), never inside this launch.
template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_alpha_kernel(const typename VT<T>::acc *partials, int grid, int K, int nrhs,
                                                       const T *delta, T *alpha, int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto dq = sum_partials_block(partials + (long long)r * grid, grid, smem, K);
    if (threadIdx.x == 0) {
        // the reference rounds dq to the value type before dividing (clcg.c:318-327)
        const T dqT = from_acc<T>(dq);
        alpha[r] = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
        if (r == 0) *iter = *iter + 1;
    }
}

// Two-level form for long partial arrays (one per row block: 39 063 at N=10M): kAlphaParts work-groups each sum one
// contiguous part (fixed order inside), the last one to finish (device ticket) adds the part sums in part order and does
// the scalar step -- the result does not depend on which work-group came last.  One work-group needed 5 rounds of 8 loads
// per thread (8 us); this needs one (4.5 us).
constexpr int kAlphaParts = 32;
template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_alpha2_kernel(const typename VT<T>::acc *partials, int grid, int nrhs,
                                                        const T *delta, T *alpha, int *iter,
                                                        typename VT<T>::acc *stage, unsigned *ticket) {
    using A = typename VT<T>::acc;
    __shared__ A smem[kScalarBlock / kWave];
    __shared__ bool last;
    const int r = blockIdx.y, part = blockIdx.x;
    const int per = (grid + kAlphaParts - 1) / kAlphaParts;
    const int lo = min(part * per, grid), hi = min(lo + per, grid);
    const A sum = sum_partials_block(partials + (long long)r * grid + lo, hi - lo, smem);
    if (threadIdx.x == 0) {
        __hip_atomic_store(reinterpret_cast<double *>(stage + (long long)r * kAlphaParts + part), to_acc2(sum).x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (VT<T>::cplx)
            __hip_atomic_store(reinterpret_cast<double *>(stage + (long long)r * kAlphaParts + part) + 1, to_acc2(sum).y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(ticket + r, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = prev + 1 == (unsigned)kAlphaParts;
        if (last) {
            __hip_atomic_store(ticket + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double2 tot = make_double2(0., 0.);
            for (int k = 0; k < kAlphaParts; ++k) {
                const double *p = reinterpret_cast<const double *>(stage + (long long)r * kAlphaParts + k);
                tot.x += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (VT<T>::cplx) tot.y += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const T dqT = from_acc<T>(from_acc2<A>(tot));
            alpha[r] = from_acc<T>(acc_div(to_acc(delta[r]), to_acc(dqT)));
            if (r == 0) *iter = *iter + 1;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kScalarBlock) void cg_beta_kernel(const typename VT<T>::acc *partials, int grid, int nrhs, T *delta,
                                                      T *beta, T *history, int history_cap, const int *iter) {
    __shared__ typename VT<T>::acc smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const auto dn = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) {
        const int it = *iter;   // already advanced by cg_alpha of this iteration
        const T dnT = from_acc<T>(dn);
        beta[r] = from_acc<T>(acc_div(to_acc(dnT), to_acc(delta[r])));   // clcg.c:389-391
        delta[r] = dnT;
        if (it < history_cap) history[(long long)it * nrhs + r] = dnT;
    }
}

// partials -> one accumulator value per RHS (input of the RCCL all-reduce in the multi-GPU loop)
template <typename A>
__global__ __launch_bounds__(kScalarBlock) void reduce_to_acc_kernel(const A *partials, int grid, int nrhs, A *out) {
    __shared__ A smem[kScalarBlock / kWave];
    const int r = blockIdx.x;
    const A s = sum_partials_block(partials + (long long)r * grid, grid, smem);
    if (threadIdx.x == 0) out[r] = s;
}

// halo pack: out[k] = v[index[k]]  (boundary entries of d that neighbouring ranks gather in their SpMV)
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(int count, const int *__restrict__ index, const T *__restrict__ v,
                                                   T *__restrict__ out) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < count; k += gridDim.x * 256) out[k] = v[index[k]];
}

int vec_grid(long long n, int dtype, int nrhs) {
    // 16-byte packs per thread the grid is sized for: 4 for streaming sizes; small systems want every CU busy instead
    // (profiles/r2_experiments/vec_ppt.log: 16k rows 10.4 -> 9.2 us per iteration with 1, 250k rows 16.2 -> 15.8 with 2,
    // N = 1M 32.2 -> 34.0 with 1); "vec_ppt" overrides
    const long long total = n * (long long)(nrhs > 0 ? nrhs : 1);
    // (up to 65536 rows always 1, whatever the number of right-hand sides: the partial-sum structure the resident loop reproduces)
    const int ppt = tune().vec_ppt > 0 ? tune().vec_ppt : ((total <= 262144 || n <= 65536) ? 1 : total <= 524288 ? 2 : 4);
    const long long per_block = (long long)kBlock * (16 / (long long)dtype_size(dtype)) * ppt;
    long long g = (n + per_block - 1) / per_block;
    const long long cap = tune().vec_grid > 0 ? tune().vec_grid : kMaxGrid;
    if (g > cap) g = cap;
    // single right-hand side, streaming sizes: exactly two work-groups per CU.  611 (1.25M rows) or 1221 (2.5M) leave the CUs
    // unevenly loaded, and every work-group of the beta launch adds all the r.r partials in its prologue: 1.25M rows 37.5 -> 36.7 us
    // per iteration, 2.5M 64.9 -> 62.5, 5M 124.8 -> 123.4, 10M 237.5 -> 235.9 (profiles/r2_experiments/vec_grid_ab.log)
    if (nrhs <= 1 && tune().vec_grid == 0 && g > 512) g = 512;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename T>
static int dot_impl(int n, const void *a, const void *b, long long ld, int nrhs, void *partials, int grid, bool vec,
                    hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    auto *pp = static_cast<typename VT<T>::acc *>(partials);
    if (vec) hipLaunchKernelGGL((dot_partials_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)a, (const T *)b, ld, pp);
    else hipLaunchKernelGGL((dot_partials_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)a, (const T *)b, ld, pp);
    return check_launch("vdot");
}
int launch_dot_partials(int dtype, int n, const void *a, const void *b, long long ld, int nrhs, void *partials, int grid,
                        hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {a, b});
    CG_DISPATCH(dtype, dot_impl, n, a, b, ld, nrhs, partials, grid, vec, st);
}

template <typename T> static int reduce_impl(const void *partials, int grid, int nrhs, void *result, hipStream_t st) {
    hipLaunchKernelGGL((reduce_to_value_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st,
                       static_cast<const typename VT<T>::acc *>(partials), grid, nrhs, static_cast<T *>(result));
    return check_launch("reduce");
}
int launch_reduce_to_value(int dtype, const void *partials, int grid, int nrhs, void *result, hipStream_t st) {
    CG_DISPATCH(dtype, reduce_impl, partials, grid, nrhs, result, st);
}

template <typename T, int OP>
static int ewise_impl(int n, const void *x, void *y, const void *b2, long long ld, const void *alpha, int nrhs, bool vec,
                      hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    if (vec) hipLaunchKernelGGL((ewise_kernel<T, kBlock, true, OP>), g, blk, 0, st, n, (const T *)x, (T *)y, (const T *)b2, ld, (const T *)alpha);
    else hipLaunchKernelGGL((ewise_kernel<T, kBlock, false, OP>), g, blk, 0, st, n, (const T *)x, (T *)y, (const T *)b2, ld, (const T *)alpha);
    return check_launch("ewise");
}
template <typename T> static int axpy_p(int n, const void *x, void *y, long long ld, const void *a, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 0>(n, x, y, nullptr, ld, a, nrhs, v, st); }
template <typename T> static int axpy_m(int n, const void *x, void *y, long long ld, const void *a, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 1>(n, x, y, nullptr, ld, a, nrhs, v, st); }
template <typename T> static int aypx_i(int n, const void *x, void *y, long long ld, const void *a, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 2>(n, x, y, nullptr, ld, a, nrhs, v, st); }
template <typename T> static int sub_i(int n, const void *a, const void *b, void *res, long long ld, int nrhs, bool v, hipStream_t st) { return ewise_impl<T, 3>(n, a, res, b, ld, nullptr, nrhs, v, st); }

int launch_axpy(int dtype, int n, const void *x, void *y, long long ld, const void *a, int sign, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y});
    if (sign) { CG_DISPATCH(dtype, axpy_p, n, x, y, ld, a, nrhs, v, st); }
    CG_DISPATCH(dtype, axpy_m, n, x, y, ld, a, nrhs, v, st);
}
int launch_aypx(int dtype, int n, const void *x, void *y, long long ld, const void *a, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y});
    CG_DISPATCH(dtype, aypx_i, n, x, y, ld, a, nrhs, v, st);
}
int launch_sub(int dtype, int n, const void *a, const void *b, void *res, long long ld, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {a, b, res});
    CG_DISPATCH(dtype, sub_i, n, a, b, res, ld, nrhs, v, st);
}

template <typename T>
static int axpy2_impl(int n, const void *d, void *x, const void *q, void *r, long long ld, const void *alpha, int nrhs,
                      void *partials, int grid, bool vec, int vnt, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    auto *pp = static_cast<typename VT<T>::acc *>(partials);
    if (vec && vnt == 1) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true, 1>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec && vnt == 2) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true, 2>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec && vnt == 3) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true, 3>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec) hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else hipLaunchKernelGGL((axpy2_dot_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    return check_launch("axpy2_dot");
}
template <typename T>
static int axpy2_alpha_impl(int n, const void *d, void *x, const void *q, void *r, long long ld, const void *part_dq, int P,
                            const CgScalars &sc, int nrhs, void *partials, int grid, bool vec, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
    if (vec) hipLaunchKernelGGL((axpy2_dot_alpha_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const A *)part_dq, P, sc.kdq, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    else hipLaunchKernelGGL((axpy2_dot_alpha_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, (T *)r, ld, (const A *)part_dq, P, sc.kdq, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    return check_launch("axpy2_dot_alpha");
}
template <typename T>
static int axpy_dot_impl(int n, const void *q, void *r, long long ld, const void *alpha, int nrhs, void *partials, int grid, bool vec,
                         int vnt, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    auto *pp = static_cast<typename VT<T>::acc *>(partials);
    if (vec && (vnt & 2)) hipLaunchKernelGGL((axpy_dot_kernel<T, kBlock, true, 2>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else if (vec) hipLaunchKernelGGL((axpy_dot_kernel<T, kBlock, true, 0>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    else hipLaunchKernelGGL((axpy_dot_kernel<T, kBlock, false, 0>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const T *)alpha, pp);
    return check_launch("axpy_dot");
}
int launch_axpy_dot(int dtype, int n, const void *q, void *r, long long ld, const void *alpha, int nrhs, void *partials, int grid,
                    hipStream_t st, int vec_nt) {
    const bool vec = vec_ok(dtype, ld, nrhs, {q, r});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, axpy_dot_impl, n, q, r, ld, alpha, nrhs, partials, grid, vec, vnt, st);
}
template <typename T>
static int axpy_dot_alpha_impl(int n, const void *q, void *r, long long ld, const void *part_dq, int P, const CgScalars &sc, int nrhs,
                               void *partials, int grid, bool vec, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
    if (vec) hipLaunchKernelGGL((axpy_dot_alpha_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const A *)part_dq, P, sc.kdq, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    else hipLaunchKernelGGL((axpy_dot_alpha_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)q, (T *)r, ld, (const A *)part_dq, P, sc.kdq, (const T *)sc.delta, (T *)sc.alpha, sc.iter, (A *)partials);
    return check_launch("axpy_dot_alpha");
}
int launch_axpy_dot_alpha(int dtype, int n, const void *q, void *r, long long ld, const void *part_dq, int P, const CgScalars &sc,
                          int nrhs, void *partials, int grid, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {q, r});
    CG_DISPATCH(dtype, axpy_dot_alpha_impl, n, q, r, ld, part_dq, P, sc, nrhs, partials, grid, vec, st);
}
template <typename T>
static int aypx_beta_x_impl(int n, const void *x, void *y, void *xs, long long ld, const void *partials, int P, int nrhs,
                            const CgScalars &sc, bool vec, int vnt, hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    auto *pp = static_cast<const typename VT<T>::acc *>(partials);
#define CG_AX(V, N) hipLaunchKernelGGL((aypx_beta_x_kernel<T, kBlock, V, N>), g, blk, 0, st, n, (const T *)x, (T *)y, (T *)xs, ld, pp, P, sc.krr, nrhs, \
                                       (const T *)sc.alpha, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, (const int *)sc.iter)
    if (vec && (vnt & 1)) CG_AX(true, 1); else if (vec) CG_AX(true, 0); else CG_AX(false, 0);
#undef CG_AX
    return check_launch("aypx_beta_x");
}
int launch_aypx_beta_x(int dtype, int n, const void *x, void *y, void *xs, long long ld, const void *partials, int P, int nrhs,
                       const CgScalars &sc, hipStream_t st, int vec_nt) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y, xs});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, aypx_beta_x_impl, n, x, y, xs, ld, partials, P, nrhs, sc, v, vnt, st);
}
bool fold_alpha_ok(int n_partials, int fold_max) { return tune().dev_no_fold_alpha == 0 && n_partials <= (fold_max > 0 ? fold_max : kFoldAlphaMax); }
int launch_axpy2_dot_alpha(int dtype, int n, const void *d, void *x, const void *q, void *r, long long ld, const void *part_dq,
                           int P, const CgScalars &sc, int nrhs, void *partials, int grid, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {d, x, q, r});
    CG_DISPATCH(dtype, axpy2_alpha_impl, n, d, x, q, r, ld, part_dq, P, sc, nrhs, partials, grid, vec, st);
}
int launch_axpy2_dot(int dtype, int n, const void *d, void *x, const void *q, void *r, long long ld, const void *alpha,
                     int nrhs, void *partials, int grid, hipStream_t st, int vec_nt) {
    const bool vec = vec_ok(dtype, ld, nrhs, {d, x, q, r});
    const int vnt = tune().vec_nt >= 0 ? tune().vec_nt : vec_nt;
    CG_DISPATCH(dtype, axpy2_impl, n, d, x, q, r, ld, alpha, nrhs, partials, grid, vec, vnt, st);
}

template <typename T> static int delta0_impl(const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    hipLaunchKernelGGL((cg_delta0_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, static_cast<const typename VT<T>::acc *>(partials),
                       grid, nrhs, (T *)s.delta, (T *)s.history, s.iter);
    return check_launch("cg_delta0");
}
int launch_cg_delta0(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    CG_DISPATCH(dtype, delta0_impl, partials, grid, nrhs, s, st);
}
template <typename T> static int alpha_impl(const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    using A = typename VT<T>::acc;
    if (s.stage && s.ticket && grid >= 16384)
        hipLaunchKernelGGL((cg_alpha2_kernel<T>), dim3(kAlphaParts, nrhs), dim3(kScalarBlock), 0, st, static_cast<const A *>(partials),
                           grid, nrhs, (const T *)s.delta, (T *)s.alpha, s.iter, (A *)s.stage, s.ticket);
    else
        hipLaunchKernelGGL((cg_alpha_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, static_cast<const A *>(partials),
                           grid, s.kdq, nrhs, (const T *)s.delta, (T *)s.alpha, s.iter);
    return check_launch("cg_alpha");
}
int launch_cg_alpha(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    CG_DISPATCH(dtype, alpha_impl, partials, grid, nrhs, s, st);
}
template <typename T> static int beta_impl(const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    hipLaunchKernelGGL((cg_beta_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, static_cast<const typename VT<T>::acc *>(partials),
                       grid, nrhs, (T *)s.delta, (T *)s.beta, (T *)s.history, s.history_cap, s.iter);
    return check_launch("cg_beta");
}
int launch_cg_beta(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st) {
    CG_DISPATCH(dtype, beta_impl, partials, grid, nrhs, s, st);
}

int launch_reduce_to_acc(int dtype, const void *partials, int grid, int nrhs, void *out, hipStream_t st) {
    if (dtype == CGAMD_F32 || dtype == CGAMD_F64)
        hipLaunchKernelGGL((reduce_to_acc_kernel<double>), dim3(nrhs), dim3(kScalarBlock), 0, st, (const double *)partials, grid, nrhs, (double *)out);
    else
        hipLaunchKernelGGL((reduce_to_acc_kernel<double2>), dim3(nrhs), dim3(kScalarBlock), 0, st, (const double2 *)partials, grid, nrhs, (double2 *)out);
    return check_launch("reduce_to_acc");
}

template <typename T> static int pack_impl(int count, const int *index, const void *v, void *out, hipStream_t st) {
    int g = (count + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL((pack_kernel<T>), dim3(g), dim3(256), 0, st, count, index, (const T *)v, (T *)out);
    return check_launch("pack");
}
int launch_pack(int dtype, int count, const int *index, const void *v, void *out, hipStream_t st) {
    if (count <= 0) return CGAMD_OK;
    CG_DISPATCH(dtype, pack_impl, count, index, v, out, st);
}

template <typename T>
static int aypx_beta_impl(int n, const void *x, void *y, long long ld, const void *partials, int P, int nrhs,
                          const CgScalars &sc, bool vec, hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    auto *pp = static_cast<const typename VT<T>::acc *>(partials);
    if (vec) hipLaunchKernelGGL((aypx_beta_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)x, (T *)y, ld, pp, P, sc.krr, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, sc.iter);
    else hipLaunchKernelGGL((aypx_beta_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)x, (T *)y, ld, pp, P, sc.krr, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, sc.iter);
    return check_launch("aypx_beta");
}
int launch_aypx_beta(int dtype, int n, const void *x, void *y, long long ld, const void *partials, int P, int nrhs,
                     const CgScalars &sc, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    const bool v = vec_ok(dtype, ld, nrhs, {x, y});
    CG_DISPATCH(dtype, aypx_beta_impl, n, x, y, ld, partials, P, nrhs, sc, v, st);
}

// ---- diagonally preconditioned CG -----------------------------------------------------------------
template <typename T>
static int pcg_axpy2_impl(bool init, int n, const void *d, void *x, const void *q, void *r, const void *m, long long ld,
                          const void *alpha, int nrhs, void *part_rz, void *part_rr, int grid, bool vec, hipStream_t st) {
    dim3 g(grid, nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
#define CG_PCG(V, I) hipLaunchKernelGGL((pcg_axpy2_dot2_kernel<T, kBlock, V, I>), g, blk, 0, st, n, (const T *)d, (T *)x, (const T *)q, \
                                        (T *)r, (const T *)m, ld, (const T *)alpha, (A *)part_rz, (A *)part_rr)
    if (init) { if (vec) CG_PCG(true, true); else CG_PCG(false, true); }
    else { if (vec) CG_PCG(true, false); else CG_PCG(false, false); }
#undef CG_PCG
    return check_launch("pcg_axpy2_dot2");
}
int launch_pcg_axpy2_dot2(int dtype, bool init, int n, const void *d, void *x, const void *q, void *r, const void *m,
                          long long ld, const void *alpha, int nrhs, void *part_rz, void *part_rr, int grid, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {d, x, q, r, m});
    CG_DISPATCH(dtype, pcg_axpy2_impl, init, n, d, x, q, r, m, ld, alpha, nrhs, part_rz, part_rr, grid, vec, st);
}
template <typename T>
static int pcg_aypx_impl(int n, const void *r, void *p, const void *m, long long ld, const void *part_rz, const void *part_rr,
                         int P, int nrhs, const CgScalars &sc, void *rho2, void *xs, bool vec, hipStream_t st) {
    dim3 g(vec_grid(n, VT<T>::dtype, nrhs), nrhs), blk(kBlock);
    using A = typename VT<T>::acc;
    if (vec) hipLaunchKernelGGL((pcg_aypx_beta_kernel<T, kBlock, true>), g, blk, 0, st, n, (const T *)r, (T *)p, (const T *)m, ld, (const A *)part_rz, (const A *)part_rr, P, sc.krr, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, (T *)rho2, (const int *)sc.iter, (T *)xs, (const T *)sc.alpha);
    else hipLaunchKernelGGL((pcg_aypx_beta_kernel<T, kBlock, false>), g, blk, 0, st, n, (const T *)r, (T *)p, (const T *)m, ld, (const A *)part_rz, (const A *)part_rr, P, sc.krr, nrhs, (T *)sc.delta, (T *)sc.beta, (T *)sc.history, sc.history_cap, (T *)rho2, (const int *)sc.iter, (T *)xs, (const T *)sc.alpha);
    return check_launch("pcg_aypx_beta");
}
int launch_pcg_aypx_beta(int dtype, int n, const void *r, void *p, const void *m, long long ld, const void *part_rz,
                         const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, void *xs, hipStream_t st) {
    const bool vec = vec_ok(dtype, ld, nrhs, {r, p, m, xs});
    CG_DISPATCH(dtype, pcg_aypx_impl, n, r, p, m, ld, part_rz, part_rr, P, nrhs, sc, rho2, xs, vec, st);
}
template <typename T>
static int pcg_delta0_impl(const void *part_rz, const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, hipStream_t st) {
    using A = typename VT<T>::acc;
    hipLaunchKernelGGL((pcg_delta0_kernel<T>), dim3(nrhs), dim3(kScalarBlock), 0, st, (const A *)part_rz, (const A *)part_rr, P, nrhs,
                       (T *)sc.delta, (T *)sc.history, (T *)rho2, sc.iter);
    return check_launch("pcg_delta0");
}
// p = beta p + m r on its own: what the chip-wide resident PCG loop leaves to the host at the end of a call (it returns the direction
// of the last iteration; the launched loop keeps the NEXT one) -- the very expression of pcg_aypx_beta_kernel
template <typename T>
__global__ __launch_bounds__(256) void pcg_p_update_kernel(long long n, const T *__restrict__ rv, T *__restrict__ pv, const T *__restrict__ m,
                                                           long long ld, const T *__restrict__ beta) {
    const int r = blockIdx.y;
    const T bt = beta[r];
    rv += (long long)r * ld; pv += (long long)r * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        pv[i] = vadd(vmul(bt, pv[i]), vmul(m[i], rv[i]));
}
template <typename T> static int pcg_p_update_impl(int n, const void *r, void *p, const void *m, long long ld, const void *beta, int nrhs, hipStream_t st) {
    const int g = (int)std::min<long long>(((long long)n + 255) / 256, 2048);
    hipLaunchKernelGGL((pcg_p_update_kernel<T>), dim3(g, nrhs), dim3(256), 0, st, (long long)n, (const T *)r, (T *)p, (const T *)m, ld, (const T *)beta);
    return check_launch("pcg_p_update");
}
int launch_pcg_p_update(int dtype, int n, const void *r, void *p, const void *m, long long ld, const void *beta, int nrhs, hipStream_t st) {
    if (n <= 0) return CGAMD_OK;
    CG_DISPATCH(dtype, pcg_p_update_impl, n, r, p, m, ld, beta, nrhs, st);
}
int launch_pcg_delta0(int dtype, const void *part_rz, const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, hipStream_t st) {
    CG_DISPATCH(dtype, pcg_delta0_impl, part_rz, part_rr, P, nrhs, sc, rho2, st);
}

}  // namespace cgamd
