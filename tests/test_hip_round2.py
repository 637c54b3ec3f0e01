"""GPU tests of what round 2 added to the boundary: the user dot-product
callback (set_dot_prod compatibility), the replicated-state digest, the large
mvec range, the scratch hygiene of red[], and a real two-GPU RCCL run."""
import math
import os
import socket
import subprocess

from launch_util import run_ranks  # noqa: E402
import sys

import numpy as np
import pytest

import parity_util as P
import scenarios as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def pairwise_dot(x, y):
    """A user dot product that is NOT the intrinsic one: pairwise summation."""
    p = x * y
    while p.size > 1:
        if p.size % 2:
            p = np.append(p, 0.0)
        p = p[0::2] + p[1::2]
    return float(p[0]) if p.size else 0.0


@pytest.mark.parametrize("flavor", [0, 1, 2])
def test_user_dot_product_callback_matches_the_oracles_set_dot_prod_run(torch_cuda, oracle, flavor):
    """set_dot_prod(dp) (F08:209-219): with the same user dp installed on the
    oracle (its reference-faithful set_dot_prod) and on the HIP path
    (nka_hip_set_host_dot), the device evaluates the reference's own dp calls on
    host copies of bit-identical operands, so results agree BIT FOR BIT --
    dependent inputs (drops), a repeated input (s == 0 -> relax), relax and
    restart included.  And the SEQUENCE of dp calls is the reference's: the same
    number of calls, in the same order, each on the same two operands (a dp with
    side effects -- call counters, per-call message tags -- cannot tell the two
    apart): the projection row is asked for AFTER the drop decisions (F08:371
    behind F08:295-347), never for an entry the update drops."""
    import nka_amd
    n, m = 1031, 5
    X = oracle.lcg_vectors(14, n, seed=5)
    basis = oracle.lcg_vectors(3, n, seed=9)
    acc = nka_amd.nka().init(n, m, flavor=flavor)
    ora = oracle.OracleNKA(n, m, flavor)
    calls = [0, 0]
    seq_a, seq_o = [], []

    def dp_a(x, y):
        calls[0] += 1
        seq_a.append((hash(np.asarray(x).tobytes()), hash(np.asarray(y).tobytes())))
        return pairwise_dot(x, y)

    def dp_o(x, y):
        calls[1] += 1
        seq_o.append((hash(np.asarray(x).tobytes()), hash(np.asarray(y).tobytes())))
        return pairwise_dot(np.asarray(x), np.asarray(y))

    acc.set_host_dot(dp_a)
    ora.set_dot_prod(dp_o)
    coef = oracle.lcg_vectors(4, 3, seed=2)
    for t in range(18):
        if t % 5 == 3:
            x = coef[t % 4] @ basis
        elif t == 7:
            x = prev.copy()
        else:
            x = X[t % 14].copy() * (1.0 + t)
        prev = x
        f = x.copy()
        ora.accel_update(f)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        out = ft.cpu().numpy()
        assert acc.num_vec() == ora.num_vec(), t
        assert acc.state().list_order() == ora.state().list_order(), t
        assert np.array_equal(out, f), (flavor, t, np.abs(out - f).max())
        assert seq_a == seq_o, (flavor, t, len(seq_a), len(seq_o))      # the reference's dp calls, no more, no fewer, in order
        if t == 10:
            acc.relax(); ora.relax()
        if t == 14:
            acc.restart(); ora.restart()
    assert calls[0] > 0 and calls[1] > 0
    # back to the device sums: results stay within tolerance of the oracle with the default dp
    acc.set_host_dot(None)
    ora2 = oracle.OracleNKA(n, m, flavor)
    spread = P.Spread(oracle, n, m)
    acc.restart()
    for t in range(8):
        x = X[t].copy()
        f = x.copy()
        ora2.accel_update(f)
        spread.update(x)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        P.check(S.rel_err(ft.cpu().numpy(), f, x), acc.state(), f"after set_host_dot(None) flavor {flavor}", where=t,
                spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))


def test_state_digest_tracks_the_replicated_state(torch_cuda):
    import nka_amd
    n, m = 4099, 4
    rng = np.random.default_rng(1)
    X = rng.standard_normal((9, n))
    a, b = nka_amd.nka().init(n, m), nka_amd.nka().init(n, m)
    assert a.state_digest() == b.state_digest()
    for t in range(9):
        for acc in (a, b):
            acc.accel_update(torch_cuda.from_numpy(X[t].copy()).cuda())
        assert a.state_digest() == b.state_digest(), t
    d0 = a.state_digest()
    y = X[3].copy()
    y[17] += 1e-9
    a.accel_update(torch_cuda.from_numpy(X[3].copy()).cuda())
    b.accel_update(torch_cuda.from_numpy(y).cuda())
    assert a.state_digest() != d0
    assert a.state_digest() != b.state_digest()     # one perturbed element changes the sums


@pytest.mark.parametrize("m", [47, 48, 62, 63, 90, 140])
def test_large_mvec_up_to_the_lds_limit(torch_cuda, oracle, m):
    """mvec up to the documented limit of 140: the wavefront solve (mvec+1 <= 63 since round 5: 47 / 48 straddle the
    48-row instantiation, 62 is the last subspace of the 63-row one, which needs more than 64 KiB of dynamic LDS), the
    one-lane solve beyond (63, 90, 140), and its opt-in above 64 KiB (mvec >= 88).  Lists longer than 32 also exercise the
    balanced passes of the window kernels (33..64 entries in two passes, 65..96 in three, ...)."""
    import nka_amd
    n = 600
    rng = np.random.default_rng(m)
    acc = nka_amd.nka().init(n, m)
    ora = oracle.OracleNKA(n, m, acc.flavor())
    spread = P.Spread(oracle, n, m)
    basis = rng.standard_normal((3, n))
    for t in range(m + 6):
        x = rng.standard_normal(n) if t % 11 != 7 else rng.standard_normal(3) @ basis
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        assert acc.num_vec() == ora.num_vec(), t
        if t % 8 == 0 or t > m:
            assert acc.state().list_order() == ora.state().list_order(), t
            P.check(S.rel_err(ft.cpu().numpy(), f, x), acc.state(), f"large mvec={m} n={n}", where=t, spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))
    assert acc.defined() and acc.num_vec() == ora.num_vec() and acc.num_vec() >= m - 8   # a few dependence drops


@pytest.mark.parametrize("m", [141, 150])
def test_mvec_beyond_the_lds_limit_works_from_global_memory(torch_cuda, oracle, m):
    """The reference accepts any mvec (F08:185-200).  Above 140 the (mvec+2)^2 matrix no longer fits
    the 160 KiB LDS of one CU and the one-lane scalar kernels work on the control block in global
    memory: slow, but the same loops, hence the same decisions and (given the same sums) bits."""
    import nka_amd
    n = 400
    rng = np.random.default_rng(m)
    acc = nka_amd.nka().init(n, m)
    ora = oracle.OracleNKA(n, m, acc.flavor())
    spread = P.Spread(oracle, n, m)
    basis = rng.standard_normal((3, n))
    for t in range(34):
        x = rng.standard_normal(n) if t % 11 != 7 else rng.standard_normal(3) @ basis
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        assert acc.num_vec() == ora.num_vec(), t
        st = acc.state()
        assert st.list_order() == ora.state().list_order() and st.free_order() == ora.state().free_order(), t
        P.check(S.rel_err(ft.cpu().numpy(), f, x), st, f"mvec={m} beyond the LDS limit, n={n}", where=t, spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))
        # the scalar step given the device's own sums: bit for bit (h by slot, coefficients)
        if t == 20:
            acc.relax(); ora.relax(); spread.relax()
    assert acc.defined()
    acc.restart(); ora.restart()
    assert acc.num_vec() == 0 and acc.defined()


def test_reduction_scratch_is_rewritten_or_zeroed_every_update(torch_cuda):
    """red[] is summed in place by the all-reduce: entries the current (short) list
    does not cover must not keep sums of an earlier, longer list."""
    import nka_amd
    n, m = 3000, 12
    rng = np.random.default_rng(3)
    acc = nka_amd.nka().init(n, m)
    for t in range(m + 3):
        acc.accel_update(torch_cuda.from_numpy(rng.standard_normal(n)).cuda())
    red = acc.reductions()
    assert np.all(red[2:2 + m] != 0.0) and np.all(red[2 + m:] != 0.0)
    acc.restart()
    for t in range(3):
        acc.accel_update(torch_cuda.from_numpy(rng.standard_normal(n)).cuda())
    red = acc.reductions()
    live = 1                                           # one older entry besides the pending pair at the third call
    assert np.all(red[2 + live:2 + m] == 0.0), red
    assert np.all(red[2 + m + live:] == 0.0), red


def test_rccl_is_bound_once_and_reports_its_library(torch_cuda):
    import nka_amd
    path = nka_amd.nka.rccl_library()
    assert "librccl" in path
    maps = open("/proc/self/maps").read()
    loaded = {ln.split()[-1] for ln in maps.splitlines() if "librccl" in ln}
    assert len(loaded) == 1, loaded                    # exactly one RCCL in this process (torch's, imported first)
    assert os.path.realpath(path) in {os.path.realpath(p) for p in loaded}


def test_vec_tol_on_a_null_handle_is_an_error_not_zero(torch_cuda):
    import nka_amd
    L = nka_amd.load()
    assert L.nka_hip_vec_tol(None) == -1.0
    assert b"null handle" in L.nka_hip_last_error()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_gpu_rccl_sharded_run(torch_cuda):
    """Row (e) with the real collective: two ranks on two GPUs, the library's own
    RCCL communicator (ncclAllReduce over xGMI on the kernel stream), against the
    unsharded oracle; replicated state digests equal on both ranks.  Skips on the
    one-GPU boxes."""
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NKA_TEST_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_sharded_gpu_worker.py")]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-5000:]
    assert p.stdout.count(" OK") == 2
    assert "hook=rccl" in p.stdout


def test_bench_two_gpus(torch_cuda):
    """bench.py --gpus 2 launched the way the driver launches it (small n)."""
    import json
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--vlen",
           "4e6", "--mvec", "6", "--steps", "5"]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["steady_state"]
    assert all(c["identical"] for c in d["replica_check"])
    assert 0.0 < d["roofline"]["frac"] <= 1.0


@pytest.mark.parametrize("world", [2, 3])
def test_bench_multi_rank_logic_rehearsed_with_ranks_sharing_the_gpu(torch_cuda, world):
    """bench.py --gpus N launched the way the driver launches it, with N ranks on the ONE GPU of the box
    (NKA_BENCH_SHARE_GPU=1, torch.distributed over gloo, the all-reduce staged through the host: RCCL refuses two
    ranks on a device).  Everything around the library that only runs with more than one rank -- slicing, the
    collective choice of the hook and its self-test, max-over-ranks timing, replica digests after warm-up and after
    the timed steps, the per-rank record gathered on rank 0 -- is executed for real; the numbers mean nothing."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", NKA_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--vlen",
           "3000001", "--mvec", "6", "--steps", "6", "--backend", "gloo", "--allreduce", "staged", "--no-cpu-baseline"]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == world and d["config"]["steady_state"] and d["scaling"] == "strong"
    assert d["config"]["parallelism"].startswith("REHEARSAL")        # (first words: a record that truncates strings still says so)
    assert len(d["replica_check"]) == 2 and all(c["identical"] and c["ranks"] == world for c in d["replica_check"])
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == list(range(world))
    assert len({r["state_digest"] for r in ranks}) == 1                    # the replicated state: the same bits on every rank
    assert all(r["hook"] == "staged" for r in ranks)
    # the slices tile the global vector
    assert ranks[0]["slice"][0] == 0 and ranks[-1]["slice"][1] == 3000001
    assert all(a["slice"][1] == b["slice"][0] for a, b in zip(ranks[:-1], ranks[1:]))
    assert sum(r["n_local"] for r in ranks) == 3000001


def test_bench_plain_form_launches_its_own_ranks(torch_cuda):
    """`python3 bench.py --gpus 2` in the PLAIN form (no torch.distributed.run in front, WORLD_SIZE unset): the parent
    touches no GPU, starts the rank group as a child process, relays rank 0's ONE line and returns the child's code
    (bench.py: launch_ranks).  Ranks share the box's one GPU, hence gloo + the host-staged hook (a rehearsal)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(NKA_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--allreduce", "staged",
           "--vlen", "3000001", "--mvec", "6", "--steps", "6", "--no-cpu-baseline"]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                      # stdout carries the one JSON line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["config"]["steady_state"]
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all("comm_nranks_rank" in r for r in d["ranks"])
    assert len(d["replica_check"]) == 2 and all(c["identical"] and c["ranks"] == 2 for c in d["replica_check"])
    assert len({r["state_digest"] for r in d["ranks"]}) == 1
    assert "all-reduce=staged" in d["config"]["parallelism"] and "gloo" in d["config"]["control_plane"]


def test_bench_plain_form_falls_back_to_the_staged_hook_when_rccl_cannot_work(torch_cuda):
    """Two ranks on ONE GPU with the default --allreduce rccl: RCCL refuses two ranks on a device, so the library's
    communicator (and torch's nccl group behind it) cannot be created.  Whichever way that surfaces -- an error the
    in-process ladder catches (rccl -> torch -> staged, decided collectively over gloo) or a dead / hung rank group that
    the supervising parent replaces by its second attempt -- the run must END with one line that names the hook that
    ran, never hang (VERDICT r3 task 1b)."""
    import json
    if torch_cuda.cuda.device_count() >= 2:
        pytest.skip("with two GPUs RCCL works: nothing to fall back from")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(NKA_BENCH_SHARE_GPU="1", NKA_BENCH_WATCHDOG_S="90")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--vlen", "2000001", "--mvec", "5", "--steps", "5",
           "--no-cpu-baseline", "--launch-timeout", "150"]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["steady_state"]
    assert all(r["hook"] == "staged" for r in d["ranks"])
    assert "all-reduce=staged" in d["config"]["parallelism"]
    assert "FALLBACK" in d["config"].get("parallelism_detail", d["config"]["parallelism"]) or "launch" in d            # says how it got there
    assert all(c["identical"] for c in d["replica_check"])


@pytest.mark.parametrize("flavor", [0, 1, 2])
@pytest.mark.parametrize("n,m", [(40961, 8), (100003, 20), (5003, 5)])
def test_every_kernel_variant_is_bit_identical(torch_cuda, flavor, n, m):
    """The streaming passes exist in two forms (every load of a tile in flight,
    rolling window) and the scalar step in two (one wavefront, or the reference's loops on one lane);
    nka_hip_set_tuning switches between them.  All of them restate the same
    arithmetic in the same order: outputs, stored vectors and the replicated state
    must agree BIT FOR BIT with the automatic choice, through list growth, capacity
    and dependence drops, a repeated input, relax and restart."""
    import nka_amd
    rng = np.random.default_rng(100 * m + flavor)
    basis = rng.standard_normal((3, n))
    X = []
    for t in range(m + 9):
        X.append(rng.standard_normal(3) @ basis if t % 6 == 4 else rng.standard_normal(n))
    X[m + 3] = X[m + 2].copy()                      # s == 0 -> relax inside the update

    def run(settings):
        # the automatic choice runs in the PRODUCT library, every forced variant in the diagnostic build of the same
        # sources (include/nka_hip_diag.h): the two builds are thereby held to the same bits as well
        acc = nka_amd.nka(diagnostic=bool(settings)).init(n, m, flavor=flavor)
        for k, v in settings.items():
            acc.set_tuning(k, v)
        outs = []
        for t, x in enumerate(X):
            ft = torch_cuda.from_numpy(x.copy()).cuda()
            acc.accel_update(ft)
            outs.append(ft.cpu().numpy())
            if t == m + 5:
                acc.relax()
            if t == m + 7:
                acc.restart()
        st = acc.state()
        live = st.list_order()
        ix = np.ix_([k - 1 for k in live[1:]], [k - 1 for k in live[1:]])
        return (outs, acc.state_digest(), acc.w(st.first), acc.v(st.first),
                (live, st.free_order(), st.subspace, st.pending, st.h[ix].tobytes(), st.c[[k - 1 for k in live[1:]]].tobytes()))

    ref = run({})
    variants = [{"pa_pipe": 0, "pb_pipe": 0}, {"pa_pipe": 201, "pb_pipe": 201}, {"pa_pipe": 202, "pb_pipe": 202},
                {"pa_pipe": 201, "pb_pipe": 0}, {"pa_pipe": 0, "pb_pipe": 201},
                {"serial_solve": 1}]
    for settings in variants:
        got = run(settings)
        for t, (a, b) in enumerate(zip(ref[0], got[0])):
            assert np.array_equal(a, b), (settings, t, np.abs(a - b).max())
        assert got[4] == ref[4], settings            # lists, free list, flags, factor and coefficients of the live entries
        if "serial_solve" not in settings:           # (the one-lane solve leaves other bits in entries nobody reads)
            assert got[1] == ref[1], settings        # digest of the whole control blocks incl. the reduced sums
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), settings
    with pytest.raises(nka_amd.NKAError):
        nka_amd.nka(diagnostic=True).init(16, 2).set_tuning("pa_pipe", 7)
    with pytest.raises(nka_amd.NKAError):                # the product has no such switch
        nka_amd.nka().init(16, 2).set_tuning("pa_pipe", 0)
    assert not hasattr(nka_amd.load(), "nka_hip_set_tuning") and not hasattr(nka_amd.load(), "nka_hip_set_grid")


@pytest.mark.parametrize("flavor", [0, 1, 2])
def test_tile_tickets_leave_every_bit_unchanged(torch_cuda, flavor):
    """PB's rolling-window kernel can take its tiles from global ticket counters
    (k_combine_win, `pb_tickets` = number of counters) instead of the static
    tile -> block mapping.  The pass is elementwise, so which block handles a tile
    must not change a bit -- and the counters must be back at zero after every
    launch (a stale counter would skip tiles of the next update).  Several tiles
    per block, ragged tail, list growth, drops, relax and restart included."""
    import nka_amd
    n, m = 256 * 512 * 5 + 77, 6
    rng = np.random.default_rng(7 + flavor)
    basis = rng.standard_normal((3, n))
    X = [rng.standard_normal(3) @ basis if t % 5 == 3 else rng.standard_normal(n) for t in range(m + 8)]
    X[m + 2] = X[m + 1].copy()                      # s == 0 -> relax inside the update

    def run(tickets, tile=1):
        acc = nka_amd.nka(diagnostic=True).init(n, m, flavor=flavor)
        acc.set_tuning("pb_pipe", 201)
        acc.set_tuning("pb_tickets", tickets)
        acc.set_tuning("pb_tile", tile)            # 2: double-width tiles (short lists only)
        outs = []
        for t, x in enumerate(X):
            ft = torch_cuda.from_numpy(x.copy()).cuda()
            acc.accel_update(ft)
            outs.append(ft.cpu().numpy())
            if t == m + 4:
                acc.relax()
            if t == m + 6:
                acc.restart()
        st = acc.state()
        return outs, acc.state_digest(), acc.w(st.first), acc.v(st.first)

    ref = run(0)
    for tickets, tile in ((1, 1), (2, 1), (4, 1), (8, 1), (-1, 1), (0, 2), (1, 2), (2, 2), (-1, -1)):
        got = run(tickets, tile)
        for t, (a, b) in enumerate(zip(ref[0], got[0])):
            assert np.array_equal(a, b), (tickets, tile, t, np.abs(a - b).max())
        assert got[1] == ref[1], (tickets, tile)
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), (tickets, tile)
    with pytest.raises(nka_amd.NKAError):
        nka_amd.nka(diagnostic=True).init(16, 2).set_tuning("pb_tickets", 3)


def test_tile_tickets_soak(torch_cuda):
    """300 consecutive updates with PB's tiles taken from the ticket counter (one
    counter; then two with double-width tiles) in lock step with an accelerator on the
    static mapping: every output bit for bit, every launch -- a rare hand-off slip
    (a tile skipped or done twice, a counter not back at zero) would show as a
    difference that persists."""
    import nka_amd
    torch = torch_cuda
    n, m = 256 * 512 * 6 + 311, 4
    accs = []
    for tickets, tile in ((0, 1), (1, 1), (2, 2)):
        a = nka_amd.nka(diagnostic=True).init(n, m, flavor=nka_amd.FLAVOR_C)
        a.set_tuning("pb_pipe", 201)
        a.set_tuning("pb_tickets", tickets)
        a.set_tuning("pb_tile", tile)
        accs.append(a)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    for t in range(300):
        x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
        outs = []
        for a in accs:
            f = x.clone()
            a.accel_update(f)
            outs.append(f)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), t
    assert accs[0].state_digest() == accs[1].state_digest() == accs[2].state_digest()


@pytest.mark.parametrize("n", [64 * 256 * 512, 64 * 256 * 512 + 1, 64 * 256 * 512 + 1023, 64 * 256 * 512 + 513 * 1024 + 77])
@pytest.mark.parametrize("flavor,m", [(2, 2), (0, 9)])
def test_automatic_tickets_at_the_threshold_sizes(torch_cuda, flavor, m, n):
    """At the sizes where the automatic rule switches the tickets on (64 tiles per
    block), with and without a ragged tail, odd and even tile counts (double-width
    tiles pair them up), the automatic choice must reproduce the static mapping bit
    for bit."""
    import nka_amd
    torch = torch_cuda
    g = torch.Generator(device="cuda")
    g.manual_seed(n % 1000 + m)
    X = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1 for _ in range(m + 3)]
    outs = []
    for tickets in (0, -1):
        acc = nka_amd.nka(diagnostic=(tickets == 0)).init(n, m, flavor=flavor)      # -1: the product's automatic choice
        if tickets == 0:
            acc.set_tuning("pb_tickets", 0)
            acc.set_tuning("pb_pipe", 201)
            acc.set_tuning("pb_tile", 1)
        res = []
        for x in X:
            f = x.clone()
            acc.accel_update(f)
            res.append(f)
        outs.append((res, acc.state_digest()))
        acc.delete()
    for t, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        assert torch.equal(a, b), t
    assert outs[0][1] == outs[1][1]


def test_failed_allreduce_leaves_the_update_undone_and_the_call_repeatable(torch_cuda):
    """include/nka_hip.h (distribution hook): "If the hook fails, nka_hip_accel_update returns
    NKA_HIP_ECOMM with the update NOT done: only scratch sums were written; f, the stored vectors,
    the lists and the host bookkeeping are as before the call" -- so the SAME call may be repeated.
    A hook that fails on its 7th call (growth phase, a pair pending, drops still to come), then a
    good hook and the call again: every later result is bit-identical to an accelerator that was
    never disturbed, through dependence drops, relax and restart."""
    import nka_amd
    torch = torch_cuda
    n, m = 5003, 4
    rng = np.random.default_rng(11)
    basis = rng.standard_normal((2, n))
    X = [rng.standard_normal(n) if t % 4 != 3 else rng.standard_normal(2) @ basis for t in range(16)]
    calls = {"n": 0, "fail_at": 7}

    def good(ptr, count, stream):          # one rank: the global sum is the local one
        assert count in (1, 1 + 2 * m, 2 + 2 * m)      # (default sums since round 6: the norm, then the rows; NKA_HIP_SUMS_BLOCKED: all at once)

    def flaky(ptr, count, stream):
        calls["n"] += 1
        if calls["n"] == calls["fail_at"]:
            raise RuntimeError("injected communication failure")

    for flavor in (nka_amd.FLAVOR_DEFAULT, nka_amd.FLAVOR_F08):
        calls["n"] = 0
        a = nka_amd.nka().init(n, m, flavor=flavor)
        b = nka_amd.nka().init(n, m, flavor=flavor)
        a.set_dot_prod(flaky)
        b.set_dot_prod(good)
        failed = 0
        for t, x in enumerate(X):
            fb = torch.from_numpy(x.copy()).cuda()
            b.accel_update(fb)
            fa = torch.from_numpy(x.copy()).cuda()
            before = (a.num_vec(), a.state().list_order(), a.state().free_order())
            try:
                a.accel_update(fa)
            except nka_amd.NKAError as exc:
                failed += 1
                assert "(-4)" in str(exc), str(exc)                      # NKA_HIP_ECOMM
                torch.cuda.synchronize()
                assert np.array_equal(fa.cpu().numpy(), x)               # f untouched
                assert (a.num_vec(), a.state().list_order(), a.state().free_order()) == before
                assert a.defined()
                a.set_dot_prod(good)                                     # re-install a working hook ...
                a.accel_update(fa)                                       # ... and repeat the SAME call
            assert np.array_equal(fa.cpu().numpy(), fb.cpu().numpy()), (flavor, t)
            assert a.num_vec() == b.num_vec()
            sa, sb = a.state(), b.state()
            assert sa.list_order() == sb.list_order() and sa.free_order() == sb.free_order()
            assert np.array_equal(sa.h, sb.h) and np.array_equal(sa.c, sb.c)
            if t == 9:
                a.relax(); b.relax()
            if t == 12:
                a.restart(); b.restart()
        assert failed == 1
        assert a.state_digest() == b.state_digest()


@pytest.mark.parametrize("flavor", [0, 1, 2])
def test_copy_in_mid_stream_gives_two_independent_accelerators(torch_cuda, oracle, flavor):
    """nka_hip_clone / nka.copy(): the reference's `b = a` is a deep copy (F08:154-168).  Copied in
    steady state with a pair pending, the two objects follow different inputs, relax and restart
    independently, and each tracks its own oracle: decisions exact, values within tolerance; the copy
    starts from the same bits (state digest, stored vectors)."""
    import nka_amd
    n, m = 3001, 5
    rng = np.random.default_rng(40 + flavor)
    a = nka_amd.nka().init(n, m, flavor=flavor)
    oa, ob = oracle.OracleNKA(n, m, flavor), oracle.OracleNKA(n, m, flavor)
    sa, sb = P.Spread(oracle, n, m), P.Spread(oracle, n, m)
    basis = rng.standard_normal((2, n))

    def step(acc, ora, spread, x, key, t):
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        assert acc.num_vec() == ora.num_vec(), (key, t)
        st = acc.state()
        assert st.list_order() == ora.state().list_order() and st.free_order() == ora.state().free_order(), (key, t)
        P.check(S.rel_err(ft.cpu().numpy(), f, x), st, f"deep copy flavor {flavor} object {key}", where=t, spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))

    for t in range(m + 3):
        x = rng.standard_normal(n)
        step(a, oa, sa, x, "a", t)
        fb = x.copy()
        ob.accel_update(fb)
        sb.update(x)
    b = a.copy()
    assert b.state_digest() == a.state_digest() and b.defined()
    for slot in a.state().list_order():
        assert np.array_equal(a.w(slot), b.w(slot)) and np.array_equal(a.v(slot), b.v(slot))
    for t in range(m + 3, m + 15):
        xa = rng.standard_normal(n) if t % 4 else rng.standard_normal(2) @ basis
        xb = rng.standard_normal(n) if t % 3 else rng.standard_normal(2) @ basis
        step(a, oa, sa, xa, "a", t)
        step(b, ob, sb, xb, "b", t)
        if t == m + 6:
            a.relax(); oa.relax(); sa.relax()
        if t == m + 9:
            b.restart(); ob.restart(); sb.restart()
    assert a.state_digest() != b.state_digest()
    a.delete()                                  # the copy owns its storage
    step(b, ob, sb, rng.standard_normal(n), "b", 99)
    assert b.defined()
