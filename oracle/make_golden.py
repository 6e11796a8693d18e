#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference, in the build container.

TEST INFRASTRUCTURE ONLY.  Run once here (`python oracle/make_golden.py`); the
GPU box never sees /root/reference, so the vectors are committed as data.

What is imported / executed from the reference (read-only, /root/reference):
  * helmFE_var.py as a module: helmFE_var (:9-331), rhsA/rhsL/rhs (:333-389),
    CG (:507-544), PCG (:546-586).  Needs only numpy/scipy.
  * three pure functions of p_h-PY_C-CL.py extracted by `ast` and exec'd with
    numpy/scipy names (the file itself cannot be imported: it needs mpi4py and
    pyopencl and runs a whole solve at import): Poisson (:1642-1682),
    CG with tolerance (:1338-1369), local_rect (:1439-1639).
No reference source text is written into the fixtures: only inputs (parameters,
seeds) and numeric outputs.
"""
import ast
import os
import sys

import numpy as np
import scipy.sparse
import scipy.sparse.linalg  # noqa: F401

REF = os.environ.get("CG_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_helm():
    sys.path.insert(0, REF)
    import helmFE_var as hv  # noqa
    sys.path.pop(0)
    return hv


def extract_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {}
    import math
    import numpy
    ns.update({k: getattr(numpy, k) for k in
               ("zeros", "ones", "dot", "sqrt", "array", "arange", "concatenate", "exp", "real", "imag")})
    ns.update(dict(scipy=scipy, math=math, np=numpy, numpy=numpy))
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    return ns


def csr_parts(A):
    A = scipy.sparse.csr_matrix(A)
    A.sort_indices()
    return dict(indptr=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32), data=A.data)


def main():
    os.makedirs(OUT, exist_ok=True)
    hv = load_helm()

    # ---- (1) helmFE_var matrices, constant and variable wave speed --------
    gen = {}
    for N in (4, 8, 16):
        H = hv.helmFE_var(N=N, omega=12.0, C=np.ones((N - 1, N - 1)), rho=0.15, Nhoriz=N, Nvert=N)
        for k, v in csr_parts(H).items():
            gen[f"helm_const_N{N}_{k}"] = v
    rng = np.random.RandomState(20250216)
    Cvar = 0.5 + rng.random_sample((11, 11))
    H = hv.helmFE_var(N=12, omega=9.5, C=Cvar, rho=0.3, Nhoriz=12, Nvert=12)
    gen["helm_var_C"] = Cvar
    for k, v in csr_parts(H).items():
        gen[f"helm_var_N12_{k}"] = v
    # rectangular sub-domain shape as as_prec uses it (p_h-PY_C-CL.py:1872-1876)
    Crect = 0.75 + rng.random_sample((6, 9))
    H = hv.helmFE_var(N=20, omega=7.0, C=Crect, rho=0.2, Nhoriz=10, Nvert=7)
    gen["helm_rect_C"] = Crect
    for k, v in csr_parts(H).items():
        gen[f"helm_rect_{k}"] = v
    # ---- (2) right-hand sides ------------------------------------------------
    for N in (4, 8, 16):
        gen[f"rhsA_N{N}"] = hv.rhsA(N, 12.0)
        gen[f"rhsL_N{N}"] = hv.rhsL(N, 12.0)
    gen["rhs_N8"] = hv.rhs(8, 12.0)
    np.savez_compressed(os.path.join(OUT, "generators.npz"), **gen)

    # ---- (3) CG iterates from the unmodified reference CG -------------------
    cgv = {}
    N = 16
    H = hv.helmFE_var(N=N, omega=12.0, C=np.ones((N - 1, N - 1)), rho=0.15, Nhoriz=N, Nvert=N)
    b = hv.rhsA(N, 12.0).flatten()
    K = 60
    X = np.stack([hv.CG(H, b, maxit=k) for k in range(0, K + 1)])   # x_0 .. x_K
    for k, v in csr_parts(H).items():
        cgv[f"helm16_{k}"] = v
    cgv["helm16_b"] = b
    cgv["helm16_X"] = X
    # warm start: x passed in (README.md:25 'can be initialized with a close')
    x0 = X[7].copy()
    cgv["helm16_warm_x0"] = x0
    cgv["helm16_warm_X5"] = hv.CG(H, b, x=x0.copy(), maxit=5)

    N = 32
    H = hv.helmFE_var(N=N, omega=12.0, C=np.ones((N - 1, N - 1)), rho=0.15, Nhoriz=N, Nvert=N)
    b = hv.rhsA(N, 12.0).flatten()
    ks = np.array([1, 2, 3, 5, 10, 25, 50, 100])
    for k, v in csr_parts(H).items():
        cgv[f"helm32_{k}"] = v
    cgv["helm32_b"] = b
    cgv["helm32_ks"] = ks
    cgv["helm32_X"] = np.stack([hv.CG(H, b, maxit=int(k)) for k in ks])

    # real SPD system through the same (complex-typed) reference CG
    fns = extract_functions(os.path.join(REF, "p_h-PY_C-CL.py"), {"Poisson", "CG", "local_rect"})
    P8 = fns["Poisson"](8)
    for k, v in csr_parts(P8).items():
        cgv[f"poisson8_{k}"] = v
    bP = np.linspace(1.0, 2.0, 64)
    cgv["poisson8_b"] = bP
    ksP = np.array([1, 2, 4, 8, 12, 16, 20])
    cgv["poisson8_ks"] = ksP
    cgv["poisson8_X"] = np.stack([hv.CG(P8, bP, maxit=int(k)) for k in ksP])
    # multiple right-hand sides, main.c:41-46 convention b[r] = (r+1)*5
    B = np.repeat((np.arange(3) + 1) * 5.0, 64).reshape(3, 64)
    cgv["poisson8_multi_B"] = B
    cgv["poisson8_multi_X10"] = np.stack([hv.CG(P8, B[r], maxit=10) for r in range(3)])
    # tolerance-stopping driver variant (UseCG==5)
    xt = fns["CG"](P8, bP.astype(complex), tol=1e-8)
    cgv["poisson8_tol1e-8_x"] = xt
    np.savez_compressed(os.path.join(OUT, "cg_iterates.npz"), **cgv)

    # ---- (4) driver-side generators -----------------------------------------
    drv = {}
    for N in (3, 5, 8):
        for k, v in csr_parts(fns["Poisson"](N)).items():
            drv[f"poisson{N}_{k}"] = v
    try:
        L = fns["local_rect"](17, k=10.0, eps=10.0, eta=10.0, L=1.0, Nhoriz=9, Nvert=6)
        for k, v in csr_parts(L).items():
            drv[f"local_rect_{k}"] = v
        drv["local_rect_params"] = np.array([17, 10.0, 10.0, 10.0, 1.0, 9, 6])
    except Exception as e:  # pragma: no cover
        print("local_rect extraction failed:", e)
    np.savez_compressed(os.path.join(OUT, "driver_generators.npz"), **drv)

    # ---- (5) diagonally preconditioned CG: the unmodified reference PCG (helmFE_var.py:546-586) ----------
    pc = {}
    N = 16
    H = scipy.sparse.csr_matrix(hv.helmFE_var(N=N, omega=12.0, C=np.ones((N - 1, N - 1)), rho=0.15, Nhoriz=N, Nvert=N))
    b = hv.rhsA(N, 12.0).flatten()
    m = 1.0 / H.diagonal()
    M = scipy.sparse.csr_matrix(scipy.sparse.diags(m))
    for k, v in csr_parts(H).items():
        pc[f"helm16_{k}"] = v
    pc["helm16_b"], pc["helm16_m"] = b, m
    K = 40     # tol = 0 never triggers the stopping test: exactly k iterations
    pc["helm16_jacobi_X"] = np.stack([hv.PCG(H, b, M=M, tol=0.0, maxit=k)[0] for k in range(1, K + 1)])
    x, i = hv.PCG(H, b, M=M, tol=1e-6, maxit=1000)
    pc["helm16_jacobi_tol1e-6_x"], pc["helm16_jacobi_tol1e-6_i"] = x, np.array(i)
    x, i = hv.PCG(H, b, M=None, tol=1e-6, maxit=1000)
    pc["helm16_none_tol1e-6_x"], pc["helm16_none_tol1e-6_i"] = x, np.array(i)
    # real SPD system with a non-constant diagonal, so that Jacobi changes the iteration
    A2 = scipy.sparse.csr_matrix(fns["Poisson"](8) + scipy.sparse.diags(np.linspace(0.0, 3.0, 64)))
    m2 = 1.0 / A2.diagonal()
    M2 = scipy.sparse.csr_matrix(scipy.sparse.diags(m2))
    for k, v in csr_parts(A2).items():
        pc[f"shifted_poisson8_{k}"] = v
    pc["shifted_poisson8_b"], pc["shifted_poisson8_m"] = bP, m2
    pc["shifted_poisson8_jacobi_X"] = np.stack([hv.PCG(A2, bP, M=M2, tol=0.0, maxit=k)[0] for k in range(1, 21)])
    x, i = hv.PCG(A2, bP, M=M2, tol=1e-10, maxit=1000)
    pc["shifted_poisson8_jacobi_tol1e-10_x"], pc["shifted_poisson8_jacobi_tol1e-10_i"] = x, np.array(i)
    np.savez_compressed(os.path.join(OUT, "pcg_iterates.npz"), **pc)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
