"""MI355X-native conjugate-gradient solver: drop-in for the CG hot path of
ziyamammadov/conjugate-gradient-pyopencl (cl.py / clcg.c), hand-written HIP for gfx950.

    from importlib import import_module
    pcl = import_module("conjugate-gradient-pyopencl_amd").cl      # same names as the reference's cl.py

Layout: csrc/ (HIP kernels + C ABI, built in-tree into libcgamd.so / liboclcg.so),
cl.py (host mirror of the reference's cl.py), mmio.py (Matrix-Market ingest),
generators.py (synthetic systems born in HBM), dist.py (row-partitioned multi-GPU CG over RCCL).
No CPU fallback: importing works anywhere, computing needs the built library and a GPU.
"""
from . import _lib, cl, mmio, generators  # noqa: F401
from ._lib import CgAmdError, LIB_PATH, LEGACY_LIB_PATH  # noqa: F401
from .cl import (CG, Context, CommandQueue, Device, DeviceBuffer, Solver, conjugate_gradient_multi_gpu,  # noqa: F401
                 solve_subdomains, solve_rhs_sharded, distribute_workloads_on_devices, distribute_computations_with_threads,
                 get_gpu_devices, initialize_cl_environment, initialize_cl_environment_with_device,
                 load_and_build_kernels)

__all__ = ["cl", "mmio", "generators", "CG", "Solver", "Context", "CommandQueue", "Device", "DeviceBuffer",
           "conjugate_gradient_multi_gpu", "solve_rhs_sharded", "distribute_workloads_on_devices",
           "distribute_computations_with_threads", "get_gpu_devices", "initialize_cl_environment",
           "initialize_cl_environment_with_device", "load_and_build_kernels", "CgAmdError"]
