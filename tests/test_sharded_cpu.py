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
    """include/nka_hip.h (core) + nka_hip_ext.h + nka_hip_vec.h vs the built library: loadable on a CPU-only box and
    exporting every entry point the headers declare (no compute calls here)."""
    import re
    import nka_amd
    from nka_amd import _lib
    def names(*hdrs):
        text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in hdrs)
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)                    # declarations only, not the prose
        return set(re.findall(r"\b(nka_(?:hip|ex)_[a-z0-9_]+)\s*\(", text))
    # the CORE (what a caller of the reference needs: VERDICT r5 item 6) is small and holds no optional machinery
    core = names("nka_hip.h")
    assert len(core) <= 30, sorted(core)
    assert not any(n.startswith(("nka_hip_p2p_", "nka_hip_vec_", "nka_hip_get_", "nka_hip_set_timing")) or
                   n in ("nka_hip_accel_update_swap", "nka_hip_capture_safe", "nka_hip_list_bound")
                   for n in core - {"nka_hip_vec_len", "nka_hip_vec_tol"}), sorted(core)      # (the accessors vec_len / vec_tol of F08:238-246)
    for need in ("nka_hip_create", "nka_hip_destroy", "nka_hip_clone", "nka_hip_accel_update", "nka_hip_accel_update_host",
                 "nka_hip_restart", "nka_hip_relax", "nka_hip_set_vec_tol", "nka_hip_num_vec", "nka_hip_max_vec", "nka_hip_vec_len",
                 "nka_hip_vec_tol", "nka_hip_defined", "nka_hip_set_host_dot", "nka_hip_set_allreduce", "nka_hip_comm_init_rank",
                 "nka_hip_set_sum_order", "nka_hip_last_error"):
        assert need in core, need
    assert sum(1 for _ in open(os.path.join(ROOT, "include", "nka_hip.h"))) <= 250
    declared = names("nka_hip.h", "nka_hip_ext.h", "nka_hip_vec.h", "nka_example_dev.h")
    assert not core & names("nka_hip_ext.h") and not core & names("nka_hip_vec.h")          # each entry declared once
    L = nka_amd.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"libnka_hip.so lacks {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # the builder's lab is NOT in the product: include/nka_hip_diag.h <-> libnka_hip_diag.so only
    lab = names("nka_hip_diag.h")
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


def test_integration_md_c_binding_compiles_against_the_core_header_alone(tmp_path):
    """the binding a maintainer of the reference adds (the C front end) needs
    include/nka_hip.h and nothing else; the opt-in out-of-place loop shown after it needs include/nka_hip_ext.h.  Both
    snippets are taken from the document and compiled (syntax only); include/nka_c_compat.h -- the shipped form of the
    first -- must not pull the optional headers in either."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```c\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 2 and '#include "nka_hip.h"' in blocks[0] and "nka_hip_accel_update_swap" in blocks[1]
    inc = os.path.join(ROOT, "include")
    core = tmp_path / "core_only.c"
    core.write_text("#define NKA_HIP_EXT_H\n#define NKA_HIP_VEC_H\n"            # (the optional headers, had they been included, would be empty)
                    "static void nka_compat_check_(int rc, const char *what) { (void)rc; (void)what; }\n"
                    "#include <stdint.h>\nstatic double nka_compat_dp_trampoline_(void *c, int64_t n, const double *x, const double *y)"
                    " { (void)c; (void)n; (void)x; (void)y; return 0.0; }\n" + blocks[0] + "\nint main(void) { return 0; }\n")
    ext = tmp_path / "with_ext.c"
    ext.write_text('#include "nka_hip_ext.h"\nextern double *my_device_buffer; extern double *x; extern nka_hip_t a;\n'
                   "void compute_correction(double *, double *); void update_solution(double *, const double *);\n"
                   "void loop(void) {\n" + blocks[1] + "}\n")
    for src in (core, ext):
        p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-unused-function", "-fsyntax-only", f"-I{inc}", str(src)],
                           capture_output=True, text=True)
        assert p.returncode == 0, (src.name, p.stderr[-2000:])
    compat = open(os.path.join(inc, "nka_c_compat.h")).read()
    assert '#include "nka_hip.h"' in compat and "nka_hip_ext.h" not in compat and "nka_hip_vec.h" not in compat
