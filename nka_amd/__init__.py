"""nka_amd -- MI355X-native drop-in for the accel_update path of nncarlson/nka.

The product is libnka_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/nka_hip.h).  This package holds only what that path needs:
  csrc/      HIP kernels + the C ABI
  fortran/   the Fortran host side (module nka_type, vector_class, ...) over iso_c_binding
  nka.py     a Python mirror of the reference's `type nka` used by tests and bench.py
  dist.py    slicing + RCCL bootstrap for the sharded (one rank per GPU) run
There is NO CPU fallback: without the HIP library every entry point raises.
"""
from ._lib import build, lib_path, load  # noqa: F401
from .nka import (FLAVOR_C, FLAVOR_DEFAULT, FLAVOR_F08, FLAVOR_F08_VECTOR, SUMS_AUTO, SUMS_BLOCKED, SUMS_BLOCKED_ROUNDED,  # noqa: F401
                  SUMS_REFERENCE_ORDER, NKAError, nka)
