"""CPU tests of the drop-in boundary: the C-ABI libraries load and export every symbol that include/*.h
declares, the legacy entry keeps the reference's exact signature, and the product path fails LOUDLY
(no CPU fallback) when there is no GPU.  No compute call is made here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

INCLUDE = os.path.join(ROOT, "include")


def _declared_functions(header):
    src = open(os.path.join(INCLUDE, header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"#[^\n]*", "", src)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def _exports(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_headers_declare_expected_surface():
    legacy = _declared_functions("clcg.h")
    assert legacy == ["cg", "connect"]
    ext = _declared_functions("cgamd.h")
    for must in ("cgamd_spmv", "cgamd_vdot", "cgamd_axpy", "cgamd_aypx", "cgamd_sub", "cgamd_solver_create",
                 "cgamd_solver_iterate", "cgamd_solver_history", "cgamd_cg", "cgamd_mm_read", "cgamd_dist_create",
                 "cgamd_gen_laplace3d", "cgamd_ctx_create"):
        assert must in ext


def test_libraries_export_every_declared_symbol(pkg):
    assert os.path.exists(pkg.LIB_PATH), "run __graft_entry__.build() first"
    ext = set(_declared_functions("cgamd.h"))
    lib_syms = _exports(pkg.LIB_PATH)
    missing = ext - lib_syms
    assert not missing, f"libcgamd.so lacks {sorted(missing)}"
    assert "cg" in lib_syms and "connect" not in lib_syms          # link-time library: no sockets-shadowing connect
    legacy_syms = _exports(pkg.LEGACY_LIB_PATH)
    assert {"cg", "connect"} <= legacy_syms and not (ext - legacy_syms)
    # ctypes table and header agree (getattr on every name happens inside load())
    lib = pkg._lib.load()
    for name in ext:
        assert hasattr(lib, name)


def test_code_objects_are_gfx950_only(pkg):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={pkg.LIB_PATH}"],
                         capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    else:   # fall back to a string scan of the fat binary
        blob = open(pkg.LIB_PATH, "rb").read()
        assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_legacy_cg_signature_matches_reference_call_site(pkg):
    """reference p_h-PY_C-CL.py:1948-1950: argtypes [c_int, c_int, csingle*, csingle*, intc*, intc*, csingle*, c_int x3]"""
    from numpy.ctypeslib import ndpointer
    libcg = ctypes.CDLL(pkg.LEGACY_LIB_PATH)
    libcg.cg.argtypes = [ctypes.c_int, ctypes.c_int, ndpointer(dtype=np.csingle, ndim=1, flags="C"),
                         ndpointer(dtype=np.csingle, ndim=1, flags="C"), ndpointer(dtype=np.intc, ndim=1, flags="C"),
                         ndpointer(dtype=np.intc, ndim=1, flags="C"), ndpointer(dtype=np.csingle, ndim=1, flags="C"),
                         ctypes.c_int, ctypes.c_int, ctypes.c_int]
    assert hasattr(libcg, "connect")
    # size 0 is a no-op on any machine and must hand x back (reference returns x: clcg.c:465)
    libcg.cg.restype = ctypes.c_void_p
    z = np.zeros(1, dtype=np.csingle)
    zi = np.zeros(1, dtype=np.intc)
    ret = libcg.cg(0, 0, z, z, zi, zi, z, 1, 0, 1)
    assert ret == z.ctypes.data


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    """in the build container there is no GPU: every compute entry must raise / report, never compute on the CPU"""
    lib = pkg._lib.load()
    n = lib.cgamd_device_count()
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.CgAmdError):
        pkg.get_gpu_devices()
    with pytest.raises(pkg.CgAmdError):
        pkg.initialize_cl_environment()
    x = np.full(4, 7.0, dtype=np.float64)
    ptr = np.array([0, 1, 2, 3, 4], dtype=np.int32)
    cols = np.arange(4, dtype=np.int32)
    rc = lib.cgamd_cg(1, 4, 4, pkg._lib.ptr(np.ones(4)), pkg._lib.ptr(np.ones(4)), pkg._lib.ptr(ptr), pkg._lib.ptr(cols),
                      pkg._lib.ptr(x), 1, 3, None, 0)
    assert rc == 2 and b"no CPU fallback" in lib.cgamd_last_error()      # CGAMD_ERR_NO_DEVICE
    assert np.all(x == 7.0)                                               # untouched: nothing was computed


PUBLIC_TUNE_KEYS = {"resident": 1, "resident_min": 8, "resident_wide": 1, "resident_wide_min": 16, "resident_claim_ms": 200,
                    "two_launch": 1, "spmm_rowmajor": 1, "index_codes": 1, "index_codes16": 1, "index_codes_min_mb": 32, "pad_rows": 1,
                    "spmv_nt": -1, "vec_nt": -1, "spmv_cycle": 64, "vec_grid": 0}


def test_tune_table_is_the_documented_one(pkg):
    """The public configuration table is the 15 keys of INTEGRATION.md section 6; the knobs of decided experiments are gone (they
    are an error now, not a silent no-op), test hooks live under the "dev." prefix."""
    import re
    lib = pkg._lib.load()
    assert len(PUBLIC_TUNE_KEYS) == 15
    for k, v in PUBLIC_TUNE_KEYS.items():
        assert lib.cgamd_tune(k.encode(), v) == 0, k
    for k in ("fold_alpha", "alpha_two_level", "defer_x", "spmv_ilv", "spmv_variant", "vec_skew", "spmm_tq", "spmm_nq",
              "resident_test_short_grid", "resident_lock", "vec_ppt"):
        assert lib.cgamd_tune(k.encode(), 1) == 1, k
    assert lib.cgamd_tune(b"dev.resident_test_short_grid", 0) == 0
    # the source has no other public key, and the documents name exactly these
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.dirname(pkg.__file__)
    api = open(os.path.join(pkg_dir, "csrc", "api.cpp")).read()
    keys = set(re.findall(r'k == "([^"]+)"', api))
    assert {k for k in keys if not k.startswith("dev.")} == set(PUBLIC_TUNE_KEYS)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for k in PUBLIC_TUNE_KEYS:
        assert "| `%s` |" % k in doc, k


def test_argument_validation_without_gpu(pkg):
    lib = pkg._lib.load()
    assert lib.cgamd_dtype_size(0) == 4 and lib.cgamd_dtype_size(1) == 8
    assert lib.cgamd_dtype_size(2) == 8 and lib.cgamd_dtype_size(3) == 16
    assert lib.cgamd_cg(1, -1, 0, None, None, None, None, None, 1, 1, None, 0) == 1     # CGAMD_ERR_INVALID
    assert lib.cgamd_cg(1, 4, 4, None, None, None, None, None, 1, 1, None, 0) == 1
    assert lib.cgamd_tune(b"no_such_key", 1) == 1
    assert lib.cgamd_tune(b"spmv_nt", -1) == 0
    out = ctypes.c_longlong()
    assert lib.cgamd_gen_laplace3d(None, 1, 250, 200, 200, 0, 10_000_000, None, None, None, ctypes.byref(out)) == 0
    assert out.value == 69_720_000
    assert lib.cgamd_gen_laplace3d(None, 1, 464, 464, 464, 0, 464 ** 3, None, None, None, ctypes.byref(out)) == 0
    assert out.value == 697_989_632
    # a slab of the 8-GPU partition of the 100M system
    n = 464 ** 3
    assert lib.cgamd_gen_laplace3d(None, 1, 464, 464, 464, n // 8, n // 4, None, None, None, ctypes.byref(out)) == 0
    assert 87_000_000 < out.value < 87_500_000
    assert lib.cgamd_gen_poisson2d(None, 1, 1000, None, None, None, ctypes.byref(out)) == 0 and out.value == 4_996_000


def test_python_module_mirrors_reference_names(pkg):
    """names and arity of the reference's cl.py (cl.py:16-44,203)"""
    import inspect
    cl = pkg.cl
    for name in ("initialize_cl_environment", "initialize_cl_environment_with_device", "get_gpu_devices",
                 "load_and_build_kernels", "CG", "conjugate_gradient_multi_gpu"):
        assert callable(getattr(cl, name))
    sig = list(inspect.signature(cl.CG).parameters)
    assert sig[:13] == ["ctx", "queue", "kernels", "size", "non_zeros", "a_values", "b_values", "a_pointers", "a_cols",
                        "x", "n_rhs", "n_iterations", "device"]
    sig = list(inspect.signature(cl.conjugate_gradient_multi_gpu).parameters)
    assert sig == ["ctx", "queue", "kernels", "size", "non_zeros", "a_values", "b_values", "a_pointers", "a_cols", "x",
                   "n_rhs", "n_iterations", "device"]
    assert cl.WAVE_SIZE == 64 and cl.LOCAL_SIZE == 256 and cl.IS_COMPLEX is True


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: no file of the product package may reference it"""
    pkgdir = os.path.join(ROOT, "conjugate-gradient-pyopencl_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "cg_oracle" not in text and "cg_numpy" not in text and "oracle/" not in text, f
    # ... nor the measurement scripts (their matrices come from the product's device generators); bench.py only in its CPU-baseline leg
    for f in os.listdir(os.path.join(ROOT, "scripts")):
        if f.endswith(".py"):
            text = open(os.path.join(ROOT, "scripts", f), errors="replace").read()
            assert "cg_oracle" not in text and "cg_numpy" not in text, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    lo, hi = bench.index("def cpu_baseline"), bench.index("def spawn_ranks")
    outside = bench[:lo] + bench[hi:]
    assert "import cg_oracle" in bench[lo:hi]
    assert "cg_oracle" not in outside.replace("oracle/cg_oracle", "") and "cg_numpy" not in outside and "import cg_" not in outside


def test_two_threads_on_the_handle_api_without_a_gpu(pkg):
    """CPU side of SURVEY 8b "Threading": error messages are per thread, and the tuning table may be edited from several
    threads at once (it is copied into every solver at creation; edits are serialised).  No compute call is made."""
    import threading
    lib = pkg._lib.load()
    seen, errors = {}, []

    def worker(k):
        try:
            for i in range(300):
                if k == 0:
                    rc = lib.cgamd_tune(b"no_such_knob_%d" % k, i)
                    msg = lib.cgamd_last_error().decode()
                    assert rc == 1 and "no_such_knob_0" in msg, msg
                else:
                    rc = lib.cgamd_ctx_create(-5 - k, ctypes.byref(ctypes.c_void_p()))
                    msg = lib.cgamd_last_error().decode()
                    # no GPU here: either "no HIP device" or, with one, "out of range" -- never the other thread's text
                    assert rc != 0 and "no_such_knob" not in msg, msg
                assert lib.cgamd_tune(b"vec_grid", 64 * (k + 1)) == 0
            seen[k] = True
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert lib.cgamd_tune(b"vec_grid", 0) == 0
    assert not errors, errors
    assert seen == {0: True, 1: True}


def test_generator_size_queries_without_gpu(pkg, golden):
    """non-zero counts of the device generators are closed-form and answered without a GPU (aPointers == NULL): checked against the
    reference's own matrices (golden) and SURVEY App. B's formula 7 N^2 - 8 N + 2"""
    import ctypes
    lib = pkg._lib.load()
    out = ctypes.c_longlong()
    g = golden["generators"]
    for N in (4, 8, 16, 500):
        pkg._lib.check(lib.cgamd_gen_helm_fe_var(None, pkg._lib.C64, N, 12.0, None, 0.15, N, N, None, None, None, ctypes.byref(out)))
        assert out.value == 7 * N * N - 8 * N + 2
        if N <= 16:
            assert out.value == len(g[f"helm_const_N{N}_indices"])
    pkg._lib.check(lib.cgamd_gen_helm_fe_var(None, pkg._lib.C64, 20, 7.0, None, 0.2, 10, 7, None, None, None, ctypes.byref(out)))
    assert out.value == len(g["helm_rect_indices"])
    d = golden["driver_generators"]
    N, k, eps, eta, L, Nh, Nv = d["local_rect_params"]
    pkg._lib.check(lib.cgamd_gen_local_rect(None, pkg._lib.C64, int(N), k, eps, eta, L, int(Nh), int(Nv), None, None, None, ctypes.byref(out)))
    assert out.value == len(d["local_rect_indices"])
    pkg._lib.check(lib.cgamd_gen_poisson2d(None, pkg._lib.F64, 8, None, None, None, ctypes.byref(out)))
    assert out.value == len(d["poisson8_indices"])
    assert lib.cgamd_gen_helm_fe_var(None, pkg._lib.C64, 1, 12.0, None, 0.15, 1, 1, None, None, None, ctypes.byref(out)) == pkg._lib.ERR_INVALID
