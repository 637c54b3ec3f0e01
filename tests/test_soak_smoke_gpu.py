"""A few seeds of the soak tool (tools/fuzz_gpu.py) as part of the suite: random call sequences -- updates with fresh,
dependent, repeated and zero inputs, relax, restart, set_vec_tol, deep copies -- at random sizes, capacities and
flavours against the oracle in lock step: through the C ABI, with a user dot product on both sides (bit for bit),
through the Fortran abstract-vector accelerator on a device block vector (nka_vector_driver script), and the latter
sharded over two processes that share the GPU.  The long runs are recorded in profiles/r03/fuzz_soak.txt."""
import os
import subprocess
import sys

import pytest

import parity_util as P
import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


@pytest.fixture(scope="module")
def fortran_build():
    import nka_amd
    if not os.path.exists(nka_amd.lib_path()):
        nka_amd.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    return os.path.join(ROOT, "nka_amd", "fortran", "build")


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("hostdot", [False, True])
def test_random_sequences_through_the_c_abi(torch_cuda, oracle, seed, hostdot):
    import fuzz_gpu
    import nka_amd
    fuzz_gpu.one_seed(seed, torch_cuda, oracle, P, S, nka_amd, steps=80, hostdot=hostdot)


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_sequences_through_the_abstract_vector_flavour(fortran_build, oracle, tmp_path, seed, world):
    import fuzz_gpu
    fuzz_gpu.one_seed_vector(seed, oracle, P, S, str(tmp_path), steps=60, world=world)
