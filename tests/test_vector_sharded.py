"""SURVEY.md 8(e), abstract-vector flavour: the reference distributes this flavour THROUGH the
vector class -- "the implementation of the vector base class reduction methods will necessarily
be parallel-aware" (src-F08-vector/README.md:16-22).  Each rank holds a contiguous slice; the
accelerator object of every rank keeps its own copy of the lists and of the Gram / Cholesky
matrix and must take the same decisions on the same (globally summed) inner products.

  * CPU (-m "not gpu"): the vector flavour of nka_type on a user-style CPU vector with
    parallel-aware dot_/norm2 (tests/fortran/host_slice_vector_type.F90; only the eleven
    deferred hooks, so the accelerator runs the reference's own hook sequence), 2 and 3
    processes, against the UNSHARDED oracle;
  * GPU: the device block vector (hip_block_vector) with reduction hooks on its workspace
    (include/nka_hip.h: nka_hip_vec_set_host_allreduce, nka_hip_vec_comm_init_rank), ranks
    sharing the one GPU of the box, against the unsharded oracle; config-5 shape (4 fields)
    included.

Decisions (num_vec after every call) exact; the digest of the replicated scalar state bitwise
equal on all ranks after every call; values within the tolerance rule of parity_util.
The cross-process sum is tests/c/shm_allreduce.c (test infrastructure: this image has no MPI).
"""
import os
import subprocess

import numpy as np
import pytest

import parity_util as P
import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FC = "/opt/rocm/bin/amdflang"


def _shm_file(tmp_path, world):
    path = tmp_path / "allreduce.shm"
    path.write_bytes(bytes(4096 + 8 * 64 * world))
    return path


def _run_ranks(cmds, timeout=300):
    procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for c in cmds]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return outs


def _parse(path, n, nfield, ncalls):
    """-> lo, hi, per call (global input, num_vec, digest, local result)"""
    raw = np.fromfile(path, dtype=np.uint8)
    lo, hi = (int(v) for v in raw[:16].view(np.int64))
    nloc = nfield * (hi - lo)
    rec = 8 * (n + 2 + nloc)
    body = raw[16:]
    assert body.size == ncalls * rec, (body.size, ncalls, rec)
    calls = []
    for t in range(ncalls):
        b = body[t * rec:(t + 1) * rec]
        x = b[:8 * n].view(np.float64)
        nv = int(b[8 * n:8 * n + 8].view(np.float64)[0])
        dig = int(b[8 * n + 8:8 * n + 16].view(np.int64)[0])
        loc = b[8 * n + 16:].view(np.float64)
        calls.append((x, nv, dig, loc))
    return lo, hi, calls


def _check_against_unsharded_oracle(oracle, files, nfield, nper, mvec, ncalls, key):
    n = nfield * nper
    ranks = [_parse(f, n, nfield, ncalls) for f in files]
    # the slices tile every field exactly
    assert ranks[0][0] == 0 and ranks[-1][1] == nper
    for a, b in zip(ranks[:-1], ranks[1:]):
        assert a[1] == b[0]
    ora = oracle.OracleNKA(n, mvec, oracle.F08_VECTOR)
    spread = P.Spread(oracle, n, mvec)
    for t in range(ncalls):
        x = ranks[0][2][t][0]
        for r in ranks[1:]:
            assert np.array_equal(r[2][t][0], x)
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        st = ora.state()
        if t + 1 == 7:
            ora.relax()
            spread.relax()
        nvs = {r[2][t][1] for r in ranks}
        digs = {r[2][t][2] for r in ranks}
        assert nvs == {ora.num_vec()}, (key, t, nvs, ora.num_vec())
        assert len(digs) == 1, (key, t, digs)              # replicated scalar state: the same bits on every rank
        got = np.empty(n)
        for lo, hi, calls in ranks:
            loc = calls[t][3].reshape(nfield, hi - lo)
            for k in range(nfield):
                got[k * nper + lo:k * nper + hi] = loc[k]
        P.check(S.rel_err(got, f, x), st, key, where=t, spread=spread.value,
                truth=spread.truth(got, x))


@pytest.fixture(scope="module")
def host_driver(tmp_path_factory):
    """The CPU driver, built here from the product's vector_class + vector-flavour nka_type and the
    test-only vector type (no libnka_hip.so, no GPU)."""
    out = tmp_path_factory.mktemp("sharded_host")
    vec = os.path.join(ROOT, "nka_amd", "fortran", "vector")
    # NKA_TEST_SANITIZE=1 (tests/test_sanitizers_cpu.py): the C all-reduce instrumented with AddressSanitizer + UBSan (clang:
    # the same runtime amdflang links) and the allocator of the whole executable interposed.  flang itself emits no
    # instrumentation for Fortran statements (probed in round 6: an out-of-bounds store goes unreported), so this covers
    # tests/c/shm_allreduce.c, the heap and the libc calls of the run, not the array statements of the Fortran modules.
    san = ["-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if os.environ.get("NKA_TEST_SANITIZE") == "1" else []
    cc = ["/opt/rocm/lib/llvm/bin/clang", "-O1"] + san if san else ["gcc", "-O2"]
    subprocess.run(cc + ["-c", "-o", str(out / "shm_allreduce.o"), os.path.join(ROOT, "tests", "c", "shm_allreduce.c")],
                   check=True)
    exe = out / "sharded_host_driver"
    subprocess.run([FC, "-O2", "-ffp-contract=off", "-cpp"] + san + ["-module-dir", str(out), "-o", str(exe),
                    os.path.join(vec, "vector_class.F90"), os.path.join(vec, "nka_type.F90"),
                    os.path.join(ROOT, "tests", "fortran", "host_slice_vector_type.F90"),
                    os.path.join(ROOT, "tests", "fortran", "sharded_host_driver.F90"), str(out / "shm_allreduce.o")],
                   check=True)
    return str(exe)


@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("world,n,mvec,ncalls", [(2, 501, 4, 16), (3, 1000, 6, 24), (2, 1, 2, 9)])
def test_vector_flavour_sharded_over_cpu_ranks_with_parallel_aware_reductions(host_driver, oracle, tmp_path, world, n,
                                                                              mvec, ncalls, compact):
    shm = _shm_file(tmp_path, world)
    files = [tmp_path / f"rank{r}.bin" for r in range(world)]
    _run_ranks([[host_driver, str(n), str(mvec), str(ncalls), str(files[r]), str(compact), str(r), str(world), str(shm)]
                for r in range(world)])
    _check_against_unsharded_oracle(oracle, files, 1, n, mvec, ncalls,
                                    f"sharded vector flavour, CPU vector, world {world} n={n} m={mvec} compact={compact}")


@pytest.fixture(scope="module")
def fortran_build():
    import nka_amd
    if not os.path.exists(nka_amd.lib_path()):
        nka_amd.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    return os.path.join(ROOT, "nka_amd", "fortran", "build")


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("world,rccl,nfield,nper,mvec,ncalls",
                         [(2, 1, 4, 2503, 6, 24),        # config-5 shape, small; RCCL (one-rank communicator) + host hook
                          (2, 0, 4, 2503, 20, 45),       # config-5 mvec; host hook only
                          (3, 1, 1, 12289, 30, 50),      # lists beyond one launch (> 24): the chunked reductions
                          (2, 1, 2, 1, 2, 9)])           # rank 0 holds an EMPTY slice and still joins every collective
def test_vector_flavour_sharded_on_device_block_vectors(fortran_build, oracle, tmp_path, world, rccl, nfield, nper,
                                                        mvec, ncalls, compact):
    """Ranks share the one GPU of the box; every reduction of hip_block_vector goes through
    the hooks of its workspace before the (per-rank) accelerator sees it."""
    shm = _shm_file(tmp_path, world)
    files = [tmp_path / f"rank{r}.bin" for r in range(world)]
    exe = os.path.join(fortran_build, "nka_vector_driver")
    _run_ranks([[exe, "shard", str(nfield), str(nper), str(mvec), str(ncalls), str(files[r]), str(compact), str(r),
                 str(world), str(shm), str(rccl)] for r in range(world)])
    _check_against_unsharded_oracle(oracle, files, nfield, nper, mvec, ncalls,
                                    f"sharded vector flavour, device block vector, world {world} {nfield}x{nper} m={mvec} "
                                    f"compact={compact} rccl={rccl}")
