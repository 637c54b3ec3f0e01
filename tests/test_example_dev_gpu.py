"""the caller of the hot path -- the example's finite-volume
system (residual / update_system / pc_ssor / u = u - r of
/root/reference/src-F08/nka_example.F90:103-179, 248) -- as device kernels
(include/nka_example_dev.h), so that a whole solve stays in HBM and feeds
accel_update with device memory.

The kernels restate the reference expressions without FMA and run the SSOR
sweeps as anti-diagonal wavefronts, which read exactly what the lexicographic
loops read: an UNACCELERATED device solve must therefore reproduce the oracle's
solution BIT FOR BIT, and the accelerated one the reference_output tables."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import scenarios as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "nka_amd", "fortran", "build")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


class DeviceExample:
    """Python twin of nka_example_dev.F90: the solve loop over the C ABI."""

    def __init__(self, torch, nx, ny, a=0.02):
        import nka_amd
        self.torch, self.L = torch, nka_amd.load()
        self.nx, self.ny = nx, ny
        self.h = C.c_void_p()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert self.L.nka_ex_create(C.byref(self.h), nx, ny, a, 0, stream) == 0, self.L.nka_hip_last_error()
        self.u = torch.zeros((nx + 2) * (ny + 2), dtype=torch.float64, device="cuda")
        self.r = torch.zeros(nx * ny, dtype=torch.float64, device="cuda")

    def close(self):
        self.L.nka_ex_destroy(self.h)

    def solve(self, nsweep=2, omega=1.4, accel=None, maxitr=999, tol=1e-6):
        L, P = self.L, (lambda t: C.c_void_p(t.data_ptr()))
        assert L.nka_ex_residual(self.h, P(self.u), P(self.r)) == 0
        rn = [float(self.torch.linalg.vector_norm(self.r))]
        for _ in range(maxitr):
            assert L.nka_ex_pc_ssor(self.h, nsweep, omega, P(self.r)) == 0
            if accel is not None:
                accel.accel_update(self.r)
            assert L.nka_ex_update_solution(self.h, P(self.u), P(self.r)) == 0
            assert L.nka_ex_residual(self.h, P(self.u), P(self.r)) == 0
            rn.append(float(self.torch.linalg.vector_norm(self.r)))
            if rn[-1] < tol * rn[0]:
                break
        u = self.u.cpu().numpy().reshape(self.ny + 2, self.nx + 2)[1:-1, 1:-1].reshape(-1).copy()
        return np.array(rn), u


@pytest.mark.parametrize("nx,ny,nsweep,maxitr", [(50, 50, 2, 40), (37, 23, 1, 25), (3, 3, 2, 10), (129, 200, 2, 6),
                                                 (400, 400, 2, 4), (1100, 7, 3, 5)])
def test_unaccelerated_device_solve_is_bit_identical_to_the_oracle(torch_cuda, oracle, nx, ny, nsweep, maxitr):
    """No accelerator in the loop: every bit of the solution after `maxitr`
    iterations of residual / wavefront SSOR / update equals the oracle's
    lexicographic C restatement (square, non-square, tiny, and wider than one
    workgroup's 1024 threads)."""
    dev = DeviceExample(torch_cuda, nx, ny)
    rn_d, u_d = dev.solve(nsweep=nsweep, maxitr=maxitr, tol=0.0)
    dev.close()
    rn_o, u_o = oracle.example_solve(nx=nx, ny=ny, nsweep=nsweep, maxitr=maxitr, tol=0.0)
    assert len(rn_d) == len(rn_o) == maxitr + 1
    assert np.array_equal(u_d, u_o), np.abs(u_d - u_o).max()
    assert np.allclose(rn_d, rn_o, rtol=1e-12, atol=0)          # only the norm's own summation order differs


@pytest.mark.parametrize("mvec,nsweep,key", [(5, 2, "f08 --nka-vec 5"), (5, 4, "f08 --sweeps 4 --nka-vec 5")])
@pytest.mark.parametrize("flavor", [0, 2])
def test_accelerated_device_resident_solve_prints_the_reference_tables(torch_cuda, oracle, mvec, nsweep, key, flavor):
    """BASELINE config 1 with nothing but the norm leaving the GPU."""
    import nka_amd
    with open(os.path.join(S.GOLD, "example_tables.json")) as fh:
        tables = json.load(fh)
    dev = DeviceExample(torch_cuda, 50, 50)
    acc = nka_amd.nka().init(2500, mvec, flavor=flavor)
    rn, _ = dev.solve(nsweep=nsweep, accel=acc)
    dev.close()
    lines = [f"{0:3d}:{rn[0]:14.6E}"] + [oracle.format_example_line(i, rn[i], rn[0]) for i in range(1, len(rn))]
    assert lines == tables[key][1:]


@pytest.mark.parametrize("args,key", [([], "f08"), (["--nka-vec", "5"], "f08 --nka-vec 5"),
                                      (["--sweeps", "4", "--nka-vec", "5"], "f08 --sweeps 4 --nka-vec 5")])
def test_fortran_device_resident_example_driver(args, key):
    """nka_example_dev: Fortran host code, every array in HBM, accel_update_dev."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    with open(os.path.join(S.GOLD, "example_tables.json")) as fh:
        tables = json.load(fh)
    p = subprocess.run([os.path.join(BUILD, "nka_example_dev")] + args, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.splitlines() == tables[key]


def _to_grid_layout(a):
    """natural (ny+2, nx+2) array [k, j] -> device grid-vector layout (interior packed, then the ring:
    row k=0, row k=ny+1, column j=0, column j=nx+1; include/nka_example_dev.h)."""
    return np.concatenate([a[1:-1, 1:-1].ravel(), a[0, :], a[-1, :], a[1:-1, 0], a[1:-1, -1]])


@pytest.mark.parametrize("nx,ny", [(50, 50), (37, 23), (3, 3), (5, 1100)])
def test_grid_vector_layout_stencils_equal_the_natural_layout_ones(torch_cuda, nx, ny):
    """nka_ex_residual_grid / nka_ex_pc_ssor_grid read the packed-ring layout of the
    device grid vector: same bits as the natural-layout entry points on the same
    values (ghosts random: the boundary terms are exercised), the ring of r left
    alone by the residual and zeroed by the preconditioner."""
    import nka_amd
    torch, L = torch_cuda, nka_amd.load()
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rng = np.random.default_rng(nx * 1000 + ny)
    u = rng.uniform(0.0, 1.0, (ny + 2, nx + 2))
    ntot, n = (nx + 2) * (ny + 2), nx * ny
    h = C.c_void_p()
    assert L.nka_ex_create(C.byref(h), nx, ny, 0.02, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    u_nat = torch.from_numpy(u.ravel().copy()).cuda()
    u_grid = torch.from_numpy(_to_grid_layout(u)).cuda()
    r_nat = torch.zeros(n, dtype=torch.float64, device="cuda")
    r_grid = torch.full((ntot,), 7.0, dtype=torch.float64, device="cuda")
    assert L.nka_ex_residual(h, P(u_nat), P(r_nat)) == 0
    assert L.nka_ex_residual_grid(h, P(u_grid), P(r_grid)) == 0
    assert np.array_equal(r_grid[:n].cpu().numpy(), r_nat.cpu().numpy())
    assert np.all(r_grid[n:].cpu().numpy() == 7.0)                       # ring untouched
    assert L.nka_ex_pc_ssor(h, 2, 1.4, P(r_nat)) == 0
    assert L.nka_ex_pc_ssor_grid(h, 2, 1.4, P(r_grid)) == 0
    assert np.array_equal(r_grid[:n].cpu().numpy(), r_nat.cpu().numpy())
    assert np.all(r_grid[n:].cpu().numpy() == 0.0)                       # r(:,:) = z, z = 0 on the ring
    L.nka_ex_destroy(h)


@pytest.mark.parametrize("args,key", [([], "f08vec"), (["--nka-vec", "5"], "f08vec --nka-vec 5"),
                                      (["--sweeps", "4", "--nka-vec", "5"], "f08vec --sweeps 4 --nka-vec 5")])
def test_vector_flavour_device_resident_example_prints_the_reference_tables(args, key):
    """nka_example_vec_dev: the caller of the abstract-vector flavour
    (src-F08-vector/nka_example.F90) with u and r as device grid vectors (ghost
    ring, hip_grid_vector_type.F90), the system on the device and the accelerator
    reaching the GPU only through the hooks of class(vector): every printed digit
    of the compiled reference's tables."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    with open(os.path.join(S.GOLD, "example_tables.json")) as fh:
        tables = json.load(fh)
    p = subprocess.run([os.path.join(BUILD, "nka_example_vec_dev")] + args, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.splitlines() == tables[key]


def test_vector_flavour_device_example_compact_option_converges_alike():
    """The compact option (v - w kept in the v slots) changes rounding only: same
    iteration count and the same residual norms to 6 digits on config 1."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    with open(os.path.join(S.GOLD, "example_tables.json")) as fh:
        want = json.load(fh)["f08vec --nka-vec 5"]
    p = subprocess.run([os.path.join(BUILD, "nka_example_vec_dev"), "--nka-vec", "5", "--compact", "1"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    got = p.stdout.splitlines()
    assert len(got) == len(want)
    for g, w in zip(got[1:], want[1:]):
        assert abs(float(g.split()[1]) / float(w.split()[1]) - 1) < 1e-5, (g, w)


def test_larger_grid_accelerated_device_solve_tracks_the_oracle(torch_cuda, oracle):
    """400 x 400 (n = 160 000), mvec = 8: residual norms of the device-resident
    solve against the oracle driven by its own accelerator."""
    import nka_amd
    nx = ny = 400
    dev = DeviceExample(torch_cuda, nx, ny)
    acc = nka_amd.nka().init(nx * ny, 8)
    rn_d, _ = dev.solve(accel=acc, maxitr=40, tol=0.0)
    dev.close()
    rn_o, _ = oracle.example_solve(nx=nx, ny=ny, accel=oracle.OracleNKA(nx * ny, 8), maxitr=40, tol=0.0)
    assert acc.num_vec() == 8
    assert np.allclose(rn_d, rn_o, rtol=1e-8, atol=0), np.abs(rn_d / rn_o - 1).max()
    assert rn_d[-1] < rn_d[0]


def test_2000x2000_device_resident_solve_matches_the_oracle_digit_for_digit(torch_cuda, oracle):
    """n = 4e6, mvec = 10, ten iterations with u, r and 80 MB of subspace in HBM:
    every printed digit of the table equals the oracle's (which takes ~0.5 s per
    iteration on a host core; the device loop ~15 ms)."""
    import nka_amd
    nx = ny = 2000
    dev = DeviceExample(torch_cuda, nx, ny)
    acc = nka_amd.nka().init(nx * ny, 10)
    rn_d, _ = dev.solve(accel=acc, maxitr=10, tol=0.0)
    dev.close()
    rn_o, _ = oracle.example_solve(nx=nx, ny=ny, accel=oracle.OracleNKA(nx * ny, 10), maxitr=10, tol=0.0)
    got = [oracle.format_example_line(i, rn_d[i], rn_d[0]) for i in range(1, 11)]
    want = [oracle.format_example_line(i, rn_o[i], rn_o[0]) for i in range(1, 11)]
    assert got == want


def test_argument_checks(torch_cuda):
    import nka_amd
    L = nka_amd.load()
    h = C.c_void_p()
    assert L.nka_ex_create(C.byref(h), 2, 50, 0.02, 0, None) == -1          # nx >= 3   (nka_example.F90:91)
    assert L.nka_ex_create(C.byref(h), 50, 50, 0.0, 0, None) == -1          # a > 0     (:90)
    assert L.nka_ex_create(C.byref(h), 2000, 2000, 0.02, 0, None) == 0
    short = torch_cuda.zeros(10, dtype=torch_cuda.float64, device="cuda")   # lives in a 2 MiB block of torch's allocator
    assert L.nka_ex_residual(h, C.c_void_p(short.data_ptr()), C.c_void_p(short.data_ptr())) == -1   # no launch on a short buffer
    assert L.nka_ex_pc_ssor(h, 0, 1.4, C.c_void_p(short.data_ptr())) == -1  # nsweep >= 1 (:158)
    L.nka_ex_destroy(h)
