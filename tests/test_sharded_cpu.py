"""N > 1 path on the CPU (gloo): slicing, id broadcast, and the sharded contract
(row (e) of SURVEY.md 8).  The GPU arithmetic itself is covered by -m gpu tests;
the 8-GPU RCCL run is the driver's."""
import os
import socket
import subprocess

from launch_util import run_ranks  # noqa: E402
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_slice_bounds_partition_and_alignment():
    from nka_amd.dist import slice_bounds
    for n in (0, 1, 7, 1000, 1001, 10**8, 10**8 + 3):
        for world in (1, 2, 3, 4, 8):
            prev_hi = 0
            for r in range(world):
                lo, hi = slice_bounds(n, world, r)
                assert lo == prev_hi and hi >= lo          # contiguous, ordered, complete
                if r > 0:
                    assert lo % 2 == 0                      # 16-byte aligned slice starts
                prev_hi = hi
            assert prev_hi == n
            sizes = [slice_bounds(n, world, r)[1] - slice_bounds(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 3
    with pytest.raises(ValueError):
        slice_bounds(10, 2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_contract_world_size_n_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_sharded_worker.py")]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count(" OK") == world


def test_c_abi_exports_every_declared_symbol():
    """include/nka_hip.h vs the built library: loadable on a CPU-only box and
    exporting every entry point the header declares (no compute calls here)."""
    import re
    import nka_amd
    from nka_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nka_hip.h")).read() + \
        open(os.path.join(ROOT, "include", "nka_example_dev.h")).read()
    declared = set(re.findall(r"\b(nka_(?:hip|ex)_[a-z0-9_]+)\s*\(", hdr)) - {"nka_hip_allreduce_fn", "nka_hip_host_allreduce_fn", "nka_hip_host_dot_fn"}
    L = nka_amd.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"libnka_hip.so lacks {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # the builder's lab is NOT in the product: include/nka_hip_diag.h <-> libnka_hip_diag.so only
    dh = open(os.path.join(ROOT, "include", "nka_hip_diag.h")).read()
    lab = set(re.findall(r"\b(nka_hip_[a-z0-9_]+)\s*\(", dh))
    assert lab == set(_lib.DIAG_SIGNATURES), lab ^ set(_lib.DIAG_SIGNATURES)
    D = _lib.load_diag()
    for name in sorted(lab):
        assert hasattr(D, name), f"libnka_hip_diag.so lacks {name}"
        assert not hasattr(L, name), f"libnka_hip.so exports the diagnostic entry {name}"
    for name in sorted(declared):
        assert hasattr(D, name), f"libnka_hip_diag.so lacks {name}"


def test_c_drop_in_library_exports_the_nine_reference_symbols():
    """libnka_c_compat.so: exactly the functions of src-C/nonlinear_krylov_accelerator.h:3-12 (plus the device-
    pointer variant), loadable without a GPU."""
    import ctypes
    import nka_amd
    nka_amd.load()
    path = os.path.join(os.path.dirname(nka_amd.lib_path()), "libnka_c_compat.so")
    assert os.path.exists(path), "build it with __graft_entry__.build()"
    L = ctypes.CDLL(path)
    for name in ("nka_init", "nka_delete", "nka_accel_update", "nka_restart", "nka_relax", "nka_num_vec", "nka_max_vec",
                 "nka_vec_len", "nka_vec_tol", "nka_accel_update_dev"):
        assert hasattr(L, name), name


def test_product_fails_loudly_without_gpu():
    import torch
    import nka_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nka_amd.NKAError):
        nka_amd.nka().init(10, 3)


def test_public_headers_compile_as_c_and_cxx(tmp_path):
    """include/*.h are the boundary a C (or Fortran-through-C) caller builds against: each must compile on its own as
    C99 and as C++17 (plain pointers and sizes, no HIP or torch types)."""
    inc = os.path.join(ROOT, "include")
    for hdr in sorted(f for f in os.listdir(inc) if f.endswith(".h")):
        for lang, comp, std in (("c", "gcc", "-std=c99"), ("c++", "g++", "-std=c++17")):
            src = tmp_path / f"use_{hdr.replace('.', '_')}.{'c' if lang == 'c' else 'cpp'}"
            src.write_text(f'#include "{hdr}"\nint main(void) {{ return 0; }}\n')
            p = subprocess.run([comp, std, "-Wall", "-Werror", "-Wno-unused-function", "-fsyntax-only", f"-I{inc}", str(src)],
                               capture_output=True, text=True)
            assert p.returncode == 0, (hdr, lang, p.stderr[-2000:])
