"""CPU tests of the Matrix-Market ingest (csrc/mmio.cpp; replaces BeBOP in reference main.c:20-33).
Parity for MM ingest is unpinned by any reference test (BeBOP is not vendored): it is pinned here
against scipy.io.mmread round trips on self-generated files."""
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp


def _check(pkg, path):
    n, ip, ix, da = pkg.mmio.mmread(path)
    want = sp.csr_matrix(scipy.io.mmread(path))
    want.sum_duplicates()
    want.sort_indices()
    got = sp.csr_matrix((da, ix, ip), shape=(n, n))
    assert n == want.shape[0]
    assert np.all(np.diff(ip) >= 0) and ip[0] == 0 and ip[-1] == len(ix)
    for r in range(n):                       # canonical: sorted, no duplicates
        c = ix[ip[r]:ip[r + 1]]
        assert np.all(np.diff(c) > 0)
    assert abs(got - want).max() < 1e-14 * max(1.0, abs(want).max())
    return n, ip, ix, da


def test_real_symmetric_spd_100(pkg, tmp_path):
    """stand-in for nos4 (100x100 SPD 'coordinate real symmetric'; the real file is not available offline)"""
    rng = np.random.default_rng(0)
    n = 100
    M = sp.random(n, n, density=0.03, random_state=np.random.RandomState(1))
    A = sp.tril(M + M.T + sp.identity(n) * 8.0).tocoo()
    p = str(tmp_path / "spd100.mtx")
    pkg.mmio.mmwrite(p, n, A.row, A.col, A.data, "real", "symmetric", comment="synthetic nos4 stand-in")
    n_, ip, ix, da = _check(pkg, p)
    full = sp.csr_matrix((da, ix, ip), shape=(n, n))
    assert abs(full - full.T).max() == 0 and da.dtype == np.float64


@pytest.mark.parametrize("field,symmetry", [("real", "general"), ("complex", "general"), ("complex", "symmetric"),
                                            ("complex", "hermitian"), ("real", "skew-symmetric"), ("integer", "general"),
                                            ("pattern", "general"), ("pattern", "symmetric")])
def test_fields_and_symmetries(pkg, tmp_path, field, symmetry):
    rng = np.random.default_rng(5)
    n = 23
    M = sp.random(n, n, density=0.15, random_state=np.random.RandomState(2)).tocoo()
    rows, cols = M.row, M.col
    if symmetry != "general":
        keep = rows > cols if symmetry == "skew-symmetric" else rows >= cols
        rows, cols = rows[keep], cols[keep]
    vals = rng.standard_normal(len(rows))
    if field == "complex":
        vals = vals + 1j * rng.standard_normal(len(rows))
        if symmetry == "hermitian":
            vals = np.where(rows == cols, vals.real, vals)
    if field == "integer":
        vals = rng.integers(-9, 9, len(rows))
    p = str(tmp_path / f"m_{field}_{symmetry}.mtx")
    pkg.mmio.mmwrite(p, n, rows, cols, vals, field, symmetry)
    n_, ip, ix, da = _check(pkg, p)
    assert (da.dtype == np.complex128) == (field == "complex")


def test_duplicates_are_summed_and_comments_skipped(pkg, tmp_path):
    p = str(tmp_path / "dup.mtx")
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% a comment\n%another\n\n3 3 5\n1 1 1.5\n1 1 2.5\n3 2 -1\n2 3 4e0\n1 3 1\n")
    n, ip, ix, da = pkg.mmio.mmread(p)
    assert n == 3 and list(ip) == [0, 2, 3, 4] and list(ix) == [0, 2, 2, 1]
    assert np.allclose(da, [4.0, 1.0, 4.0, -1.0])


def test_reference_hinted_2x2_complex_case(pkg, tmp_path):
    """p_helmholtz.py:1813-1816 hints at a 2x2 complex test matrix with b = [3-4j, -1+0.5j] (file not in the tree)"""
    p = str(tmp_path / "c2.mtx")
    pkg.mmio.mmwrite(p, 2, [0, 1, 1], [0, 0, 1], [2 + 1j, 0.5 - 0.25j, 3 - 1j], "complex", "symmetric")
    n, ip, ix, da = pkg.mmio.mmread(p)
    A = sp.csr_matrix((da, ix, ip), shape=(2, 2)).toarray()
    assert np.allclose(A, [[2 + 1j, 0.5 - 0.25j], [0.5 - 0.25j, 3 - 1j]])


@pytest.mark.parametrize("text,msg", [
    ("", "empty"), ("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", "coordinate"),
    ("%%MatrixMarket matrix coordinate real general\n2 3 1\n1 1 1\n", "square"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1\n", "end of file"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1\n", "out of range"),
    ("%%MatrixMarket matrix coordinate quaternion general\n2 2 1\n1 1 1\n", "field"),
    ("hello\n", "banner")])
def test_malformed_files_are_reported(pkg, tmp_path, text, msg):
    p = str(tmp_path / "bad.mtx")
    open(p, "w").write(text)
    with pytest.raises(pkg.CgAmdError) as e:
        pkg.mmio.mmread(p)
    assert e.value.status == 5 and msg in str(e.value)
    with pytest.raises(pkg.CgAmdError):
        pkg.mmio.mmread(str(tmp_path / "missing.mtx"))


def test_cli_usage_and_read_error(pkg, tmp_path):
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "oclcgex")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr                       # main.c:15-18
    r = subprocess.run([exe, str(tmp_path / "nope.mtx"), "1", "0", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not read matrix" in r.stdout       # main.c:21-24
