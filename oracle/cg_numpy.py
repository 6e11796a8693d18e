"""NumPy restatement of the reference's CG recurrence and matrix generators.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

Each function cites the reference file:line it follows (paths relative to the
upstream repository root).  The restatement is pinned against outputs of the
unmodified reference (helmFE_var.py imported in the build container) by
oracle/make_golden.py -> tests/golden/*.npz and tests/test_oracle_golden.py.
"""
import numpy as np


# ----------------------------------------------------------------------------
# CSR helpers (a7: CSR triple, int32 indices, 0-based, RHS-major RHS block)
# ----------------------------------------------------------------------------
def csr_matvec(indptr, indices, data, x):
    """y = A x for one vector; per-row left-to-right sum in CSR order.

    Same operation as scipy's csr_matvec used by the reference oracle
    (helmFE_var.py:510,518 `A.dot(x)`), which also sums each row left to right.
    """
    prod = data * x[indices]
    n = len(indptr) - 1
    y = np.zeros(n, dtype=np.result_type(data.dtype, x.dtype))
    # np.add.reduceat sums left to right inside each segment; empty rows need care
    nz_rows = np.flatnonzero(np.diff(indptr) > 0)
    if len(nz_rows):
        # sequential accumulation to stay bit-identical with scipy's loop
        rows = np.repeat(np.arange(n), np.diff(indptr))
        np.add.at(y, rows, prod)
    return y


def cg_fixed(indptr, indices, data, b, x0=None, maxit=10, dtype=None, matvec=None):
    """Fixed-iteration (CO)CG; follows helmFE_var.py:507-544 line by line.

    No convergence test, unconjugated dot (numpy.dot, helmFE_var.py:516,522,535),
    exactly `maxit` iterations.  Returns (x, delta_history) where
    delta_history[0] = r0.r0 and delta_history[k] = deltaNew after iteration k
    (the reference computes these but never returns them: clcg.c:274-292,376-387).
    """
    if dtype is None:
        dtype = np.result_type(data.dtype, b.dtype)
    b = np.asarray(b, dtype=dtype)
    data = np.asarray(data, dtype=dtype)
    if matvec is None:
        try:
            import scipy.sparse as sp
            A = sp.csr_matrix((data, indices, indptr), shape=(len(indptr) - 1,) * 2)
            matvec = A.dot
        except Exception:  # pragma: no cover
            matvec = lambda v: csr_matvec(indptr, indices, data, v)
    x = np.zeros(b.size, dtype=dtype) if x0 is None else np.array(x0, dtype=dtype)
    r = b - matvec(x)                       # helmFE_var.py:510-512
    d = r                                   # :514
    delta_new = np.dot(r, r)                # :516
    hist = [delta_new]
    for _ in range(maxit):                  # :519
        q = matvec(d)                       # :520
        dq = np.dot(d, q)                   # :522
        alpha = delta_new / dq              # :524
        x = x + alpha * d                   # :527
        r = r - alpha * q                   # :530
        delta_old = delta_new               # :533
        delta_new = np.dot(r, r)            # :535
        beta = delta_new / delta_old        # :539
        d = r + beta * d                    # :542
        hist.append(delta_new)
    return x, np.array(hist, dtype=dtype)


def cg_fixed_multi(indptr, indices, data, B, X0, maxit, dtype=None):
    """Batched independent CG over nRHS right-hand sides (RHS-major block).

    clcg.c:317-333,376-392: each RHS r has its own alpha[r], beta[r]; layout
    element i of RHS r at i + r*size (kernel/real/spmv.cl:25,48).
    B, X0: arrays of shape (nrhs, size).  Returns (X, hist[maxit+1, nrhs]).
    """
    B = np.atleast_2d(B)
    X0 = np.atleast_2d(X0)
    xs, hs = [], []
    for r in range(B.shape[0]):
        x, h = cg_fixed(indptr, indices, data, B[r], X0[r], maxit, dtype=dtype)
        xs.append(x)
        hs.append(h)
    return np.stack(xs), np.stack(hs, axis=1)


def cg_tol(indptr, indices, data, b, x0=None, tol=1e-5, maxit=1000):
    """Tolerance-stopping variant: p_h-PY_C-CL.py:1338-1369 (UseCG==5 path).

    Stops when sqrt(|r.r|) < tol; at most 2*b.size iterations (the reference
    ignores `maxit` and loops `range(2*b.size)`, :1349).
    Returns (x, iterations_done).
    """
    import scipy.sparse as sp
    A = sp.csr_matrix((data, indices, indptr), shape=(len(indptr) - 1,) * 2)
    x = np.zeros(b.size, dtype=complex) if x0 is None else x0
    r = b - A.dot(x)
    it = 0
    for i in range(2 * b.size):
        z = r
        rho = np.dot(r, z)
        if i == 0:
            p = z
        else:
            beta = rho / rho_2
            p = z + beta * p
        q = A.dot(p)
        alpha = rho / np.dot(p, q)
        x = x + alpha * p
        r = r - alpha * q
        res2norm = np.sqrt(abs(np.dot(r, r)))
        it = i + 1
        if res2norm < tol:
            break
        rho_2 = rho
    return x, it


def pcg_diag(indptr, indices, data, b, m=None, x0=None, tol=1e-6, maxit=1000, history=False):
    """Preconditioned CG with a diagonal M: helmFE_var.py:546-586, branch `z = M.dot(r)` (M a CSR matrix with
    nnz <= n, i.e. diagonal) or `z = r` for M None.  m is that diagonal as a vector.  Unconjugated dots, stop on
    sqrt(|r.r|) < tol AFTER the update, returns (x, i) with i the index of the last iteration run -- exactly the
    reference's return value.  history=True additionally returns [r0.r0, r1.r1, ...] for the device parity tests."""
    import scipy.sparse as sp
    A = sp.csr_matrix((data, indices, indptr), shape=(len(indptr) - 1,) * 2)
    M = None if m is None else sp.csr_matrix(sp.diags(np.asarray(m)))   # the reference multiplies through a CSR matvec
    x = np.zeros(b.size, dtype=complex) if x0 is None else x0
    r = b - A.dot(x)
    hist = [np.dot(r, r)]
    i = -1
    for i in range(maxit):
        z = r if M is None else M.dot(r)
        rho = np.dot(r, z)
        if i == 0:
            p = z
        else:
            beta = rho / rho_2
            p = z + beta * p
        q = A.dot(p)
        alpha = rho / np.dot(p, q)
        x = x + alpha * p
        r = r - alpha * q
        hist.append(np.dot(r, r))
        if np.sqrt(abs(hist[-1])) < tol:
            break
        rho_2 = rho
    return (x, i, np.asarray(hist)) if history else (x, i)


# ----------------------------------------------------------------------------
# Generators
# ----------------------------------------------------------------------------
def poisson2d(N, dtype=np.float64):
    """2-D 5-point FD Laplacian, Dirichlet: diag 4, off -1.

    Restates Poisson(N) p_h-PY_C-CL.py:1642-1682 (config C2 at N=1000).
    Returns canonical CSR (sorted columns) as (indptr, indices, data).
    """
    n = N * N
    idx = np.arange(n, dtype=np.int64)
    i, j = idx // N, idx % N
    cols = [idx - N, idx - 1, idx, idx + 1, idx + N]
    mask = [i > 0, j > 0, np.ones(n, bool), j < N - 1, i < N - 1]
    vals = [-1.0, -1.0, 4.0, -1.0, -1.0]
    return _assemble_banded(n, cols, mask, vals, dtype)


def laplace3d(nx, ny, nz, dtype=np.float64):
    """3-D 7-point Laplacian, x fastest, Dirichlet: diag 6, off -1 (SURVEY §8d 'M').

    No reference generator exists for this (the reference has no 3-D case);
    it is the synthetic headline matrix of BASELINE.json.
    """
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64)
    ix = idx % nx
    iy = (idx // nx) % ny
    iz = idx // (nx * ny)
    cols = [idx - nx * ny, idx - nx, idx - 1, idx, idx + 1, idx + nx, idx + nx * ny]
    mask = [iz > 0, iy > 0, ix > 0, np.ones(n, bool), ix < nx - 1, iy < ny - 1, iz < nz - 1]
    vals = [-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0]
    return _assemble_banded(n, cols, mask, vals, dtype)


def _assemble_banded(n, cols, mask, vals, dtype):
    M = np.stack(mask, axis=1)                       # n x k
    C = np.stack(cols, axis=1)
    if np.isscalar(vals[0]):
        V = np.broadcast_to(np.asarray(vals, dtype=dtype), M.shape)
    else:
        V = np.stack(vals, axis=1).astype(dtype)
    counts = M.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = C[M].astype(np.int32)
    data = np.ascontiguousarray(V[M], dtype=dtype)
    assert indptr[-1] < 2 ** 31
    return indptr.astype(np.int32), indices, data


def helm_fe_var(N, omega, C, rho, Nhoriz, Nvert):
    """Vectorised restatement of helmFE_var(N,omega,C,rho,Nhoriz,Nvert),
    helmFE_var.py:9-331 (config C3 generator: N=500, omega=12, C=ones, rho=0.15).

    P1 finite elements on a structured triangulation of a rectangle with
    impedance boundary; node (j,m) = column j, row m, index m*Nhoriz+j.
    Returns canonical CSR (indptr int32, indices int32, data complex128), i.e.
    what scipy.sparse.csr_matrix((c,(a,b))) yields in the reference (:329).
    """
    C = np.asarray(C, dtype=float)
    h = 1.0 / (N - 1.0)                               # :47
    h2 = h ** 2                                       # :48
    nn = Nhoriz * Nvert
    cr = (1.0 + rho * 1j)
    K = omega / C                                     # k on each square [m, j]

    rows, cols, vals = [], [], []

    def add(mask, rowidx, colidx, val):
        rows.append(rowidx[mask])
        cols.append(colidx[mask])
        vals.append(np.broadcast_to(val, mask.shape)[mask] if np.ndim(val) else
                    np.full(int(mask.sum()), val, dtype=complex))

    jj, mm = np.meshgrid(np.arange(Nhoriz), np.arange(Nvert), indexing="xy")  # [m, j]
    row = mm * Nhoriz + jj

    def sq(m_idx, j_idx):
        """k on square (m_idx, j_idx) with clipping for masked-out entries."""
        return K[np.clip(m_idx, 0, Nvert - 2), np.clip(j_idx, 0, Nhoriz - 2)]

    knw = sq(mm, jj - 1)
    ksw = sq(mm - 1, jj - 1)
    kne = sq(mm, jj)
    kse = sq(mm - 1, jj)

    bot, top = mm == 0, mm == Nvert - 1
    left, right = jj == 0, jj == Nhoriz - 1
    inner_j = (~left) & (~right)
    inner_m = (~bot) & (~top)

    # ---- diagonal entries (:77-138)
    diag = np.zeros((Nvert, Nhoriz), dtype=complex)
    k = kne
    diag = np.where(bot & left, 1. - cr * k ** 2 * h2 / 6. - 1j * k * 2 * h / 3., diag)          # :80-85
    k = knw
    diag = np.where(bot & right, 1. - cr * k ** 2 * h2 / 12. - 1j * k * 2. * h / 3., diag)       # :86-91
    k = kse
    diag = np.where(top & left, 1. - cr * k ** 2 * h2 / 12. - 1j * k * 2. * h / 3., diag)        # :92-97
    k = ksw
    diag = np.where(top & right, 1. - cr * k ** 2 * (h2 / 6.) - 1j * k * 2. * h / 3., diag)      # :98-103
    kl, kr = knw, kne
    diag = np.where(bot & inner_j, 2. - cr * (kl ** 2 + 2. * kr ** 2) * h2 / 12. - 1j * (kl + kr) * h / 3., diag)  # :107-113
    kl, kr = ksw, kse
    diag = np.where(top & inner_j, 2. - cr * (2. * kl ** 2 + kr ** 2) * h2 / 12. - 1j * (kl + kr) * h / 3., diag)  # :114-120
    kt, kb = kne, kse
    diag = np.where(left & inner_m, 2. - cr * (2. * kt ** 2 + kb ** 2) * h2 / 12. - 1j * (kt + kb) * h / 3., diag)  # :121-127
    kt, kb = knw, ksw
    diag = np.where(right & inner_m, 2. - cr * (kt ** 2 + 2. * kb ** 2) * h2 / 12. - 1j * (kt + kb) * h / 3., diag)  # :128-134
    diag = np.where(inner_j & inner_m,
                    4. - cr * (knw ** 2 + 2. * ksw ** 2 + 2. * kne ** 2 + kse ** 2) * h2 / 12., diag)  # :135-143
    add(np.ones_like(bot), row, row, diag)

    def edge(k):
        return -0.5 - cr * k ** 2 * h2 / 24. - 1j * k * h / 6.

    def diagnb(k):
        return - cr * k ** 2 * h2 / 12.

    # ---- corners (:147-198)
    m_ = bot & left
    add(m_, row, row + 1, edge(kne)); add(m_, row, row + Nhoriz, edge(kne)); add(m_, row, row + Nhoriz + 1, diagnb(kne))
    m_ = top & right
    add(m_, row, row - 1, edge(ksw)); add(m_, row, row - Nhoriz, edge(ksw)); add(m_, row, row - Nhoriz - 1, diagnb(ksw))
    m_ = bot & right
    add(m_, row, row - 1, edge(knw)); add(m_, row, row + Nhoriz, edge(knw))
    m_ = top & left
    add(m_, row, row + 1, edge(kse)); add(m_, row, row - Nhoriz, edge(kse))

    # ---- bottom edge (:202-222)
    m_ = bot & inner_j
    kl, kr = knw, kne
    add(m_, row, row - 1, edge(kl)); add(m_, row, row + 1, edge(kr))
    add(m_, row, row + Nhoriz, -1. - cr * (kl ** 2 + kr ** 2) * h2 / 24.)
    add(m_, row, row + Nhoriz + 1, diagnb(kr))
    # ---- top edge (:226-245)
    m_ = top & inner_j
    kl, kr = ksw, kse
    add(m_, row, row - 1, edge(kl)); add(m_, row, row + 1, edge(kr))
    add(m_, row, row - Nhoriz, -1. - cr * (kl ** 2 + kr ** 2) * h2 / 24.)
    add(m_, row, row - Nhoriz - 1, diagnb(kl))
    # ---- right edge (:249-269)
    m_ = right & inner_m
    kb, kt = ksw, knw
    add(m_, row, row - Nhoriz, edge(kb)); add(m_, row, row + Nhoriz, edge(kt))
    add(m_, row, row - 1, -1. - cr * (kb ** 2 + kt ** 2) * h2 / 24.)
    add(m_, row, row - Nhoriz - 1, diagnb(kb))
    # ---- left edge (:273-293)
    m_ = left & inner_m
    kb, kt = kse, kne
    add(m_, row, row - Nhoriz, edge(kb)); add(m_, row, row + Nhoriz, edge(kt))
    add(m_, row, row + 1, -1. - cr * (kb ** 2 + kt ** 2) * h2 / 24.)
    add(m_, row, row + Nhoriz + 1, diagnb(kt))
    # ---- interior (:297-326)
    m_ = inner_j & inner_m
    add(m_, row, row + 1, -1. - cr * (kne ** 2 + kse ** 2) * h2 / 24.)
    add(m_, row, row - 1, -1. - cr * (knw ** 2 + ksw ** 2) * h2 / 24.)
    add(m_, row, row + Nhoriz, -1. - cr * (knw ** 2 + kne ** 2) * h2 / 24.)
    add(m_, row, row - Nhoriz, -1. - cr * (ksw ** 2 + kse ** 2) * h2 / 24.)
    add(m_, row, row - Nhoriz - 1, diagnb(ksw))
    add(m_, row, row + Nhoriz + 1, diagnb(kne))

    a = np.concatenate(rows)
    b = np.concatenate(cols)
    c = np.concatenate(vals)
    # canonical CSR: sort by (row, col); the pattern has no duplicates
    order = np.lexsort((b, a))
    a, b, c = a[order], b[order], c[order]
    indptr = np.zeros(nn + 1, dtype=np.int64)
    np.cumsum(np.bincount(a, minlength=nn), out=indptr[1:])
    return indptr.astype(np.int32), b.astype(np.int32), c.astype(complex)


def rhsA(N, k):
    """helmFE_var.py:379-389: k^2 on the four boundary lines of an N x N grid."""
    b = np.zeros((N, N), dtype=complex)
    b[:, 0] = k * k
    b[:, -1] = k * k
    b[0, :] = k * k
    b[-1, :] = k * k
    return b


def rhsL(N, k):
    """helmFE_var.py:370-377: k^2 on the left boundary, corners excluded."""
    b = np.zeros((N, N), dtype=complex)
    b[1:N - 1, 0] = k * k
    return b


def cli_rhs(n, nrhs, dtype):
    """main.c:41-46: b[r*n+i] = (r+1)*5, x0 = 0."""
    return np.repeat((np.arange(nrhs) + 1) * 5.0, n).astype(dtype).reshape(nrhs, n)


# ----------------------------------------------------------------------------
# Algorithmic byte model -- shared by bench.py and DESIGN.md
# ----------------------------------------------------------------------------
def spmv_bytes(n, nnz, vbytes, nrhs=1, ibytes=4):
    return nnz * (vbytes + ibytes) + (n + 1) * ibytes + 2 * n * vbytes * nrhs


def cg_iter_bytes(n, nnz, vbytes, nrhs=1, ibytes=4, fused=False):
    return nnz * (vbytes + ibytes) + (n + 1) * ibytes + (11 if fused else 14) * n * vbytes * nrhs
