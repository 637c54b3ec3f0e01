"""Parity of the HIP path (through the C ABI of libnka_hip.so) against the oracle
and against the fixtures generated from the compiled reference.

The bar:
  * decisions -- num_vec trace, list order, slot ids, free list, s == 0 -- EXACT;
  * the scalar step (Cholesky with drops + substitutions) BIT-EXACT given the
    same dot products;
  * the elementwise statements BIT-EXACT given the same scalars;
  * dot products and the returned f within fp64 tolerance:
        ||f_hip - f_ref||_2 / ||f_in||_2 <= 1e-12   (n <= 1e5)
    (the reductions are summed in a different -- blocked, fixed -- order).
"""
import os

import numpy as np
import pytest

from launch_util import run_ranks  # noqa: E402

import parity_util as P
import scenarios as S

pytestmark = pytest.mark.gpu

TOL_SMALL = 1e-12   # n <= 1e5, well-conditioned subspace
FLAVORS = {0: "f_out_f08", 2: "f_out_c", 1: "f_out_f08vec"}


def cond_tol(state, base=TOL_SMALL):
    """1e-12 unscaled while the smallest pivot is > 0.5, else 1e-12 / pivot^2
    (tests/parity_util.py explains the rule)."""
    return P.tolerance(state, base)[0]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def make_acc(n, m, flavor=0, sums=None):
    """This file holds the FAST passes to their bars at every n (blocked, fused sums; raw-sum Gram row), also up to 64
    elements, where a single-rank accelerator would sum in the reference's order by itself (nka_hip_set_sum_order; a
    sharded one keeps the fast passes there).  Reference-order sums have their own file: tests/test_hip_reference_order.py."""
    import nka_amd
    return nka_amd.nka().init(n, m, flavor=flavor).set_sum_order(nka_amd.SUMS_BLOCKED if sums is None else sums)


def dev_update(torch):
    def update(acc, f):
        t = torch.from_numpy(f).cuda()
        acc.accel_update(t)
        return t.cpu().numpy()
    return update


@pytest.mark.parametrize("name", S.scenario_names())
@pytest.mark.parametrize("flavor", [0, 1, 2])
def test_scenarios_decisions_exact_values_within_tolerance(torch_cuda, oracle, name, flavor):
    g = S.load(name)
    key = FLAVORS[flavor]
    if key not in g.files:
        pytest.skip("fixture has no output for this flavour")
    n, m = int(g["n"]), int(g["mvec"])
    acc = make_acc(n, m, flavor)
    states = []
    outs, trace = S.replay(acc, g, update=dev_update(torch_cuda), after_update=lambda u, a: states.append(a.state()))
    assert acc.defined()
    assert np.array_equal(trace, g["num_vec"])                       # decisions: exact
    inputs = [g["inputs"][int(i)] for op, i, _ in g["ops"] if int(op) == S.OP_UPDATE]
    spreads = P.fixture_spreads(g)          # the three REFERENCE outputs of the same calls (diagnostic "K needed")
    # THE rule (parity_util): the device against the extended-precision trajectory, allowed twice the distance the
    # COMPILED reference's own outputs (held by the fixture) have from it
    exact, err_ref = P.fixture_truth(g, oracle)
    for u in range(len(outs)):
        truth = (S.rel_err(outs[u], exact[u], inputs[u]), err_ref[u], n, m)
        P.check(S.rel_err(outs[u], g[key][u], inputs[u]), states[u], f"scenario {name} flavor {flavor} vs own reference",
                where=u, spread=spreads[u], truth=truth)
        # every flavour -- the front ends' default C/compact one included -- against the
        # reference FORTRAN path (src-F08) on the same inputs, same rule
        P.check(S.rel_err(outs[u], g["f_out_f08"][u], inputs[u]), states[u],
                f"scenario {name} flavor {flavor} vs src-F08 reference", where=u, spread=spreads[u], truth=truth)
    if "first" in g.files:                                           # list state of the C reference
        for u, st in enumerate(states):
            assert (st.first, st.last, st.free) == (g["first"][u], g["last"][u], g["free"][u]), (name, u)
            assert st.subspace == bool(g["subspace"][u]) and st.pending
            assert np.array_equal(st.next, g["next"][u]), (name, u)
            order = st.list_order()
            for k in order:
                assert st.prev[k - 1] == g["prev"][u][k - 1]
            live = order[1:]
            hh, gg = st.h[np.ix_([k - 1 for k in live], [k - 1 for k in live])], \
                g["h"][u][np.ix_([k - 1 for k in live], [k - 1 for k in live])]
            assert np.allclose(hh, gg, rtol=0, atol=cond_tol(st)), (name, u)


@pytest.mark.parametrize("name", ["S2_dependence", "S3_zero_difference", "S9_near_dependence", "S1_capacity"])
def test_scalar_step_bit_exact_given_same_dots(torch_cuda, oracle, name):
    """Feed the oracle's scalar step the dot products the device produced: the
    device solve kernel (one wavefront) must then agree BIT FOR BIT on the
    factor h, the coefficients c and every list decision."""
    g = S.load(name)
    n, m = int(g["n"]), int(g["mvec"])
    acc = make_acc(n, m, 0)
    ora = oracle.OracleNKA(n, m, oracle.F08)

    def check(u, a):
        red = a.reductions()
        so = ora.state()                      # oracle state at entry of this update
        hrow = np.zeros(m + 2)
        b = np.zeros(m + 2)
        order = so.list_order()
        older = order[1:] if so.pending else order
        # the device solve forms the Gram row of w1' = d/s by ONE division of the
        # raw sums <d,w_k>, <f,d> by s = sqrt(<d,d>)  (numpy: IEEE sqrt and divide)
        s = np.sqrt(red[0]) if so.pending else 0.0
        for p, k in enumerate(older):
            hrow[k] = red[2 + p] / s if (so.pending and s != 0.0) else 0.0
            b[k] = red[2 + m + p]
        if so.pending and s != 0.0:
            b[so.first] = red[1] / s
        new = ora.scalar_step(float(s), hrow, b)
        sd, sn = a.state(), ora.state()
        assert (sd.first, sd.last, sd.free, sd.subspace, sd.pending) == (sn.first, sn.last, sn.free, sn.subspace, sn.pending)
        assert sd.first == new
        assert np.array_equal(sd.next, sn.next)
        live = sn.list_order()[1:]
        for i in live:
            assert sd.prev[i - 1] == sn.prev[i - 1]
            assert sd.c[i - 1] == sn.c[i - 1], (name, u, i)            # bit for bit
            for j in live:
                assert sd.h[i - 1, j - 1] == sn.h[i - 1, j - 1], (name, u, i, j)

    u = 0
    for op, idx, val in g["ops"]:
        op = int(op)
        if op == S.OP_UPDATE:
            t = torch_cuda.from_numpy(g["inputs"][int(idx)].copy()).cuda()
            acc.accel_update(t)
            check(u, acc)
            u += 1
        elif op == S.OP_RESTART:
            acc.restart(); ora.restart()
        elif op == S.OP_RELAX:
            acc.relax(); ora.relax()
        elif op == S.OP_SET_VEC_TOL:
            acc.set_vec_tol(float(val)); ora.set_vec_tol(float(val))
        assert acc.num_vec() == ora.num_vec()


@pytest.mark.parametrize("flavor", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 2, 255, 513, 4099])
def test_elementwise_statements_bit_exact_given_same_scalars(torch_cuda, flavor, n):
    """w1' = (w1-f)/s, v1' = v1/s, the combine and both ring stores, recomputed
    with numpy (IEEE, no FMA) from the device's own scalars, must match the
    device bit for bit -- including ragged tails and the 1-element case."""
    m = 3
    rng = np.random.default_rng(100 + n)
    acc = make_acc(n, m, flavor)
    prev_in = None
    for t in range(7):
        f_in = rng.standard_normal(n)
        st0 = acc.state()
        first0 = st0.first
        w1_raw = acc.w(first0) if st0.pending else None
        v1_raw = acc.v(first0) if st0.pending else None
        ft = torch_cuda.from_numpy(f_in.copy()).cuda()
        acc.accel_update(ft)
        f_out = ft.cpu().numpy()
        st = acc.state()
        red = acc.reductions()
        new = st.first
        assert np.array_equal(acc.w(new), f_in)                    # w(:,new) = f       F08:361
        assert np.array_equal(acc.v(new), f_out)                   # v(:,new) = f_out   F08:404
        if st0.pending:
            assert np.array_equal(w1_raw, prev_in)
            s = np.sqrt(red[0])
            d = w1_raw - f_in
            if flavor == 1:
                r = 1.0 / s
                wn, vn = r * d, r * v1_raw
            else:
                wn, vn = d / s, v1_raw / s
            if flavor == 2:
                vn = vn - wn          # compact storage: the v array keeps v' - w' (C .c:423 operand)
            assert np.array_equal(acc.w(first0), wn)
            assert np.array_equal(acc.v(first0), vn)
        x = f_in.copy()
        for k in st.list_order()[1:]:
            c, wk, vk = st.c[k - 1], acc.w(k), acc.v(k)
            if flavor == 0:
                x = (x - c * wk) + c * vk
            elif flavor == 1:
                x = ((-c) * wk + c * vk) + x
            else:
                x = x + c * vk        # vk is the stored difference v' - w'
        assert np.array_equal(f_out, x)
        prev_in = f_in


def test_dot_products_within_tolerance(torch_cuda):
    n, m = 100003, 6
    rng = np.random.default_rng(5)
    acc = make_acc(n, m)
    for t in range(9):
        f_in = rng.standard_normal(n)
        st0 = acc.state()
        order0 = st0.list_order()
        olders = order0[1:] if st0.pending else order0
        W = {k: acc.w(k) for k in olders}
        w1_raw = acc.w(st0.first) if st0.pending else None
        ft = torch_cuda.from_numpy(f_in.copy()).cuda()
        acc.accel_update(ft)
        red = acc.reductions()
        if st0.pending:
            d = w1_raw - f_in
            nd, nf = np.linalg.norm(d), np.linalg.norm(f_in)
            assert red[0] == pytest.approx(float(d @ d), rel=1e-13)
            assert red[1] == pytest.approx(float(f_in @ d), abs=1e-13 * nf * nd)
            for p, k in enumerate(olders):
                assert red[2 + p] == pytest.approx(float(d @ W[k]), abs=1e-13 * nd)
                assert red[2 + m + p] == pytest.approx(float(f_in @ W[k]), abs=1e-13 * nf)


def test_medium_case_against_f08_reference_fixture(torch_cuda):
    g = np.load(os.path.join(S.GOLD, "medium_n100000_m10.npz"))
    n, m, calls = int(g["n"]), int(g["mvec"]), int(g["calls"])
    rng = np.random.Generator(np.random.PCG64(int(g["seed"])))
    acc = make_acc(n, m)
    probe = np.cos(np.arange(n) * 0.001)
    for t in range(calls):
        f = rng.random(n) * 2.0 - 1.0
        fin_norm = np.linalg.norm(f)
        ft = torch_cuda.from_numpy(f).cuda()
        acc.accel_update(ft)
        out = ft.cpu().numpy()
        assert acc.num_vec() == g["num_vec"][t]
        # sampled entries / norm / probe functional of the reference output
        assert np.linalg.norm(out[g["idx"]] - g["out_samples"][t]) <= TOL_SMALL * fin_norm
        assert abs(np.linalg.norm(out) - g["out_norm"][t]) <= TOL_SMALL * fin_norm
        assert abs(float(out @ probe) - g["out_probe"][t]) <= TOL_SMALL * fin_norm * np.linalg.norm(probe)


@pytest.mark.parametrize("flavor", [0, 1, 2])
@pytest.mark.parametrize("n,m", [(0, 2), (1, 1), (7, 8), (1000, 1), (777, 40), (2048, 33), (5000, 64)])
def test_edge_shapes_against_oracle(torch_cuda, oracle, n, m, flavor):
    """Empty and tiny vectors, mvec = 1, and mvec beyond one unrolled pass (the
    PA/PB kernels then run several passes of 32)."""
    rng = np.random.default_rng(n * 131 + m)
    acc, ora = make_acc(n, m, flavor), oracle.OracleNKA(n, m, flavor)
    spread = P.Spread(oracle, n, m)
    ncall = min(m + 4, 45)
    basis = rng.standard_normal((3, n))
    for t in range(ncall):
        x = rng.standard_normal(n) if (t % 5) else rng.standard_normal(3) @ basis
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        out = ft.cpu().numpy()
        assert acc.num_vec() == ora.num_vec(), (t,)
        assert acc.state().list_order() == ora.state().list_order()
        assert acc.state().free_order() == ora.state().free_order()
        if n:
            P.check(S.rel_err(out, f, x), acc.state(), f"edge shape n={n} m={m} flavor {flavor}", where=t, spread=spread.value,
                truth=spread.truth(out, x))
    assert acc.defined()


def test_unaligned_device_pointer_takes_scalar_path(torch_cuda, oracle):
    n, m = 3001, 5
    rng = np.random.default_rng(9)
    acc, ora = make_acc(n, m), oracle.OracleNKA(n, m)
    buf = torch_cuda.zeros(n + 1, dtype=torch_cuda.float64, device="cuda")
    view = buf[1:]                                   # 8-byte but not 16-byte aligned
    assert view.data_ptr() % 16 == 8
    for t in range(9):
        x = rng.standard_normal(n)
        f = x.copy()
        ora.accel_update(f)
        view.copy_(torch_cuda.from_numpy(x))
        acc.accel_update(view)
        assert acc.num_vec() == ora.num_vec()
        P.record(S.rel_err(view.cpu().numpy(), f, x), TOL_SMALL, "unaligned pointer n=3001 m=5")


def test_host_array_compat_entry_and_config1_example(torch_cuda, oracle):
    """BASELINE config 1: the 50x50 example driven through the HIP path (host
    array entry point, like the reference signature) reproduces reference_output."""
    import json
    with open(os.path.join(S.GOLD, "example_tables.json")) as fh:
        tables = json.load(fh)
    for mvec, nsweep, key in ((5, 2, "f08 --nka-vec 5"), (5, 4, "f08 --sweeps 4 --nka-vec 5")):
        acc = make_acc(2500, mvec)
        rn, _ = oracle.example_solve(nsweep=nsweep, accel=acc)
        lines = [f"{0:3d}:{rn[0]:14.6E}"] + [oracle.format_example_line(i, rn[i], rn[0]) for i in range(1, len(rn))]
        assert lines[-1] == tables[key][-1]
        assert lines == tables[key][1:]


def test_api_surface_defaults_and_errors(torch_cuda):
    import nka_amd
    a = nka_amd.nka()
    with pytest.raises(nka_amd.NKAError):
        a.num_vec()                                   # used before init
    with pytest.raises(nka_amd.NKAError):
        a.init(10, 0)                                 # mvec > 0      (F08:190)
    with pytest.raises(nka_amd.NKAError):
        a.init(-1, 3)                                 # vlen >= 0     (F08:191)
    a.init(10, 3)
    assert (a.vec_len(), a.max_vec(), a.num_vec(), a.vec_tol()) == (10, 3, 0, 0.01)
    assert a.defined()
    with pytest.raises(nka_amd.NKAError):
        a.set_vec_tol(0.0)                            # vtol > 0      (F08:205)
    a.set_vec_tol(0.5)
    assert a.vec_tol() == 0.5
    f = torch_cuda.arange(10, dtype=torch_cuda.float64, device="cuda")
    g = f.clone()
    a.accel_update(g)
    assert torch_cuda.equal(f, g) and a.num_vec() == 0      # first call returns f unchanged
    a.relax()
    assert a.num_vec() == 0 and a.state().first == 0 and a.defined()
    with pytest.raises(nka_amd.NKAError):
        a.accel_update(torch_cuda.zeros(11, dtype=torch_cuda.float64, device="cuda"))   # size(f) == vlen (F08:258)
    # the C ABI itself refuses a buffer that is too short or not device memory (no kernel is launched)
    import ctypes as C
    L = nka_amd.load()
    short = torch_cuda.zeros(4, dtype=torch_cuda.float64, device="cuda")
    big = nka_amd.nka().init(1 << 22, 2)
    assert L.nka_hip_accel_update(big._handle(), C.c_void_p(short.data_ptr())) == -1
    assert b"shorter" in L.nka_hip_last_error()
    host = np.zeros(1 << 22)
    assert L.nka_hip_accel_update(big._handle(), C.c_void_p(host.ctypes.data)) == -1
    assert big.num_vec() == 0 and big.defined()
    a.init(10, 3)                                     # re-init resets vtol (intent(out), F08:186)
    assert a.vec_tol() == 0.01
    # an allocation the device cannot satisfy (3 slots x 2 arrays x 160 GB) fails cleanly with NKA_HIP_ENOMEM:
    # nothing is left allocated, the library stays usable
    free0 = torch_cuda.cuda.mem_get_info()[0]
    h = C.c_void_p()
    assert L.nka_hip_create(C.byref(h), 20_000_000_000, 2, 0.01, nka_amd.FLAVOR_DEFAULT, 0, None) == -3
    assert not h.value and b"hipMalloc" in L.nka_hip_last_error()
    assert torch_cuda.cuda.mem_get_info()[0] >= free0 - (64 << 20)
    assert nka_amd.nka().init(10, 3).defined()
    # the flavour every front end runs by default, and its override
    assert nka_amd.nka().init(10, 3).flavor() == nka_amd.nka.default_flavor()
    assert nka_amd.nka().init(10, 3, flavor=nka_amd.FLAVOR_F08).flavor() == nka_amd.FLAVOR_F08
    assert L.nka_hip_create(C.byref(h), 10, 3, 0.01, 7, 0, None) == -1                   # unknown flavour


def test_flavor_environment_variable(torch_cuda):
    """NKA_HIP_FLAVOR decides what NKA_HIP_FLAVOR_DEFAULT resolves to (include/nka_hip.h); a value that is not a
    flavour is refused with a message instead of silently running something else."""
    import subprocess
    import sys
    root = os.path.dirname(S.GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    code = ("import nka_amd, torch; torch.cuda.set_device(0); a = nka_amd.nka().init(16, 2); print('FLAVOR', a.flavor())")
    for val, want in (("f08", "FLAVOR 0"), ("f08vec", "FLAVOR 1"), ("c", "FLAVOR 2"), ("", "FLAVOR 2")):
        env = dict(os.environ, NKA_HIP_FLAVOR=val, PYTHONPATH=root)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and want in p.stdout, (val, p.stdout[-300:], p.stderr[-600:])
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NKA_HIP_FLAVOR="fortran", PYTHONPATH=root),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "NKA_HIP_FLAVOR" in p.stderr


def test_runs_are_bitwise_reproducible(torch_cuda):
    n, m = 200001, 7
    outs = []
    for rep in range(2):
        rng = np.random.default_rng(77)
        acc = make_acc(n, m)
        res = []
        for t in range(12):
            ft = torch_cuda.from_numpy(rng.standard_normal(n)).cuda()
            acc.accel_update(ft)
            res.append(ft.cpu().numpy())
        outs.append(res)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_rccl_hook_single_rank_and_python_hook(torch_cuda, oracle):
    """The distribution hook with one rank: the built-in RCCL all-reduce and a
    Python hook must both leave the results unchanged."""
    import nka_amd
    n, m = 5003, 4
    rng = np.random.default_rng(2)
    X = rng.standard_normal((8, n))
    ref = make_acc(n, m)
    a_rccl = make_acc(n, m)
    a_rccl.use_rccl(nka_amd.nka.rccl_unique_id(), 1, 0)
    a_py = make_acc(n, m)
    calls = []
    a_py.set_dot_prod(lambda ptr, count, stream: calls.append(count))
    for t in range(8):
        outs = []
        for acc in (ref, a_rccl, a_py):
            ft = torch_cuda.from_numpy(X[t].copy()).cuda()
            acc.accel_update(ft)
            outs.append(ft.cpu().numpy())
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert set(calls) == {2 + 2 * m}          # ONE exchange per update: the norm and both Gram rows


@pytest.mark.parametrize("hook", ["rccl", "torch"])
def test_bench_multi_gpu_plumbing_rehearsal_single_rank(torch_cuda, hook):
    """bench.py with the N > 1 plumbing forced on at world size 1 (process group,
    id broadcast, RCCL communicator or torch all-reduce hook on the library's
    device buffer), launched the way the driver launches it."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(S.GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NKA_BENCH_FORCE_HOOK="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--vlen", "2e6",
           "--mvec", "6", "--steps", "5", "--no-cpu-baseline", "--allreduce", hook]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["config"]["steady_state"]
    assert hook in d["config"]["parallelism"]
    assert d["value"] > 0


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_input_takes_the_same_decisions_as_the_oracle(torch_cuda, oracle, bad):
    """A NaN / Inf in f: the reference has no guard (s == 0 is false for NaN,
    hkk > vtol**2 is false for NaN -> the entry is dropped); the device takes the
    same branches, and restart() recovers a clean object."""
    n, m = 257, 3
    rng = np.random.default_rng(17)
    acc, ora = make_acc(n, m), oracle.OracleNKA(n, m)
    for t in range(7):
        x = rng.standard_normal(n)
        if t == 3:
            x[5] = bad
        f = x.copy()
        ora.accel_update(f)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        out = ft.cpu().numpy()
        assert acc.num_vec() == ora.num_vec(), t
        assert acc.state().list_order() == ora.state().list_order(), t
        assert np.array_equal(np.isnan(out), np.isnan(f)), t
    acc.restart(); ora.restart()
    for t in range(4):
        x = rng.standard_normal(n)
        f = x.copy()
        ora.accel_update(f)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        assert acc.num_vec() == ora.num_vec()
        P.record(S.rel_err(ft.cpu().numpy(), f, x), TOL_SMALL, "debug mode n=1000 m=3")


@pytest.mark.skipif(not __import__("oracle.oracle_py", fromlist=["x"]).have_ref(),
                    reason="compiled reference (oracle/_ref) did not travel to this box")
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("flavor", [0, 1, 2])
def test_against_the_live_compiled_reference(torch_cuda, oracle, seed, flavor):
    """The reference's own src-F08 module (compiled from /root/reference into
    oracle/_ref and shipped with the repo) and the HIP path, side by side, on
    fresh random streams with dependent vectors, relax and restart mixed in.
    All three roundings of the HIP path -- flavour 2 is the bench's headline
    (src-C combine, compact storage) -- are held to the FORTRAN reference:
    decisions exact, values within the stated tolerance."""
    n, m = 20011, 8
    rng = np.random.default_rng(seed)
    ref = oracle.RefF08(n, m)
    acc = make_acc(n, m, flavor)
    spread = P.Spread(oracle, n, m)
    basis = rng.standard_normal((4, n))
    worst = 0.0
    for t in range(40):
        x = rng.standard_normal(4) @ basis if t % 6 == 4 else rng.standard_normal(n)
        if t == 11:
            x = prev.copy()                      # repeated input -> s == 0 -> relax inside the update
        prev = x
        f = x.copy()
        ref.accel_update(f)
        spread.update(x)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        assert acc.num_vec() == ref.num_vec(), (seed, flavor, t)
        err = P.check(S.rel_err(ft.cpu().numpy(), f, x), acc.state(),
                      f"live src-F08 reference n={n} m={m} flavor {flavor}", where=(seed, t), spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))
        worst = max(worst, err)
        if t == 20:
            ref.relax(); acc.relax(); spread.relax()
        if t == 30:
            ref.restart(); acc.restart(); spread.restart()
    assert ref.defined() and acc.defined()


def test_sharded_hip_path_two_ranks_one_gpu(torch_cuda):
    """Row (e) on real hardware: two ranks (sharing the one GPU of the test box)
    run the HIP path on contiguous slices with the all-reduce hook; decisions,
    replicated scalars and the returned slices are checked against the
    unsharded oracle (tests/_sharded_gpu_worker.py)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(S.GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "_sharded_gpu_worker.py")]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-5000:]
    assert p.stdout.count(" OK") == 2


def test_debug_mode_checks_invariants_every_update(torch_cuda, oracle, monkeypatch):
    """NKA_HIP_DEBUG=1 is the counterpart of the reference's non-NDEBUG build
    (ASSERT(defined(this)) on entry of accel_update, F08:257)."""
    monkeypatch.setenv("NKA_HIP_DEBUG", "1")
    n, m = 1000, 3
    rng = np.random.default_rng(4)
    acc, ora = make_acc(n, m), oracle.OracleNKA(n, m)
    for t in range(8):
        x = rng.standard_normal(n)
        f = x.copy()
        ora.accel_update(f)
        ft = torch_cuda.from_numpy(x.copy()).cuda()
        acc.accel_update(ft)
        assert S.rel_err(ft.cpu().numpy(), f, x) <= TOL_SMALL


def test_follows_the_current_torch_stream(torch_cuda, oracle):
    """Updates issued under `with torch.cuda.stream(side)` run on that stream and
    stay ordered with the updates issued before and after on the default stream."""
    torch = torch_cuda
    n, m = 50021, 4
    rng = np.random.default_rng(8)
    acc, ora = make_acc(n, m), oracle.OracleNKA(n, m)
    side = torch.cuda.Stream()
    for t in range(10):
        x = rng.standard_normal(n)
        f = x.copy()
        ora.accel_update(f)
        if t % 3 == 1:
            with torch.cuda.stream(side):
                ft = torch.from_numpy(x.copy()).cuda()
                acc.accel_update(ft)
                out = ft.cpu().numpy()
        else:
            ft = torch.from_numpy(x.copy()).cuda()
            acc.accel_update(ft)
            out = ft.cpu().numpy()
        assert acc.num_vec() == ora.num_vec()
        P.record(S.rel_err(out, f, x), TOL_SMALL, "torch stream following n=50021 m=4")


@pytest.mark.parametrize("n,tickets", [(200003, -1), (400037, 1)])
def test_steady_state_update_is_hipgraph_capturable(torch_cuda, n, tickets):
    """accel_update only enqueues kernels; once capture_safe() is true, ONE update
    captured into a hipGraph and replayed with fresh inputs reproduces the eager
    results bit for bit (launch-bound small-n loops can be replayed as graphs).
    Second case: PB takes its tiles from the global ticket counter -- device-side
    state that every launch, replayed ones included, must leave at zero."""
    import nka_amd
    torch = torch_cuda
    m = 6
    rng = np.random.default_rng(0)
    X = [rng.standard_normal(n) for _ in range(m + 12)]
    ref = nka_amd.nka().init(n, m)
    want = []
    for x in X:
        t = torch.from_numpy(x.copy()).cuda()
        ref.accel_update(t)
        want.append(t.cpu().numpy())
    acc = nka_amd.nka(diagnostic=tickets >= 0).init(n, m)
    if tickets >= 0:
        acc.set_tuning("pb_pipe", 201)
        acc.set_tuning("pb_tickets", tickets)
    side = torch.cuda.Stream()
    static = torch.empty(n, dtype=torch.float64, device="cuda")
    got = []
    with torch.cuda.stream(side):
        for i, x in enumerate(X[:m + 3]):
            assert acc.capture_safe() == (i >= m + 1)
            static.copy_(torch.from_numpy(x))
            acc.accel_update(static)
            got.append(static.cpu().numpy())
    torch.cuda.synchronize()
    assert acc.capture_safe()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        acc.accel_update(static)
    for x in X[m + 3:]:
        static.copy_(torch.from_numpy(x))
        g.replay()
        torch.cuda.synchronize()
        got.append(static.cpu().numpy())
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    assert acc.num_vec() == ref.num_vec() == m
    acc.relax()
    assert not acc.capture_safe()


@pytest.mark.parametrize("m,seed", [(2, 0), (5, 1), (9, 2), (5, 3), (33, 4)])
def test_randomised_call_sequences_against_oracle(torch_cuda, oracle, m, seed):
    """Long random sequences of every public operation -- updates with fresh,
    dependent, repeated and zero inputs, relax, restart, set_vec_tol -- must keep
    the device state machine (and the host's bounds that size the kernels) in
    lock step with the oracle: num_vec, list order and free list after EVERY
    call, values within the conditioning-aware tolerance."""
    n = 257
    rng = np.random.default_rng(1000 + seed)
    acc = make_acc(n, m)
    ora = oracle.OracleNKA(n, m, acc.flavor())
    spread = P.Spread(oracle, n, m)
    basis = rng.standard_normal((3, n))
    prev = rng.standard_normal(n)
    nupd = 0
    for step in range(300):
        r = rng.random()
        if r < 0.80:
            kind = rng.random()
            if kind < 0.55:
                x = rng.standard_normal(n)
            elif kind < 0.85:
                x = rng.standard_normal(3) @ basis          # dependent: forces drops
            elif kind < 0.95:
                x = prev.copy()                             # repeated: s == 0 -> relax
            else:
                x = np.zeros(n)
            prev = x
            f = x.copy()
            ora.accel_update(f)
            spread.update(x)
            ft = torch_cuda.from_numpy(x.copy()).cuda()
            acc.accel_update(ft)
            nupd += 1
            st = acc.state()
            nx = np.linalg.norm(x)
            if nx > 0:
                P.check(S.rel_err(ft.cpu().numpy(), f, x), st, f"random call sequence m={m} seed={seed}",
                        where=(step, nupd), spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))
        elif r < 0.88:
            acc.relax(); ora.relax(); spread.relax()
        elif r < 0.93:
            acc.restart(); ora.restart(); spread.restart()
        else:
            vt = float(10.0 ** rng.uniform(-3, -0.3))
            acc.set_vec_tol(vt); ora.set_vec_tol(vt); spread.set_vec_tol(vt)
        assert acc.num_vec() == ora.num_vec(), step
        sa, so = acc.state(), ora.state()
        assert sa.list_order() == so.list_order(), step
        assert sa.free_order() == so.free_order(), step
        assert (sa.subspace, sa.pending) == (so.subspace, so.pending), step
    assert acc.defined()
