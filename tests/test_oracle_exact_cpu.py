"""CPU tests of the error-measurement instruments (checker code only): the extended-precision restatement
(oracle/nka_oracle_exact.c) that serves as the truth of the parity rule, and the error-attribution dot products."""
import math

import numpy as np
import pytest

import parity_util as P
import scenarios as S


@pytest.mark.parametrize("name", S.scenario_names())
def test_extended_precision_run_takes_the_compiled_references_decisions(oracle, name):
    """Same statements, same list logic (the pinned source compiled a second time): on every fixture the
    extended-precision run must reproduce the num_vec trace and the list state of the COMPILED reference, and
    stay within the reference flavours' own neighbourhood (it is the limit they all round towards)."""
    g = S.load(name)
    n, m = int(g["n"]), int(g["mvec"])
    acc = oracle.OracleExact(n, m)
    states = []
    outs, trace = S.replay(acc, g, after_update=lambda u, a: states.append(a.state()))
    assert np.array_equal(trace, g["num_vec"])
    assert acc.defined()
    if "first" in g.files:
        for u, st in enumerate(states):
            assert (st.first, st.last, st.free) == (g["first"][u], g["last"][u], g["free"][u]), (name, u)
            assert np.array_equal(st.next, g["next"][u]), (name, u)
    exact, err_ref = P.fixture_truth(g, oracle)
    assert all(np.array_equal(a, b) for a, b in zip(exact, outs))          # deterministic
    spreads = P.fixture_spreads(g)
    # the reference's distance from the truth is of the order of its own inter-flavour spread, never wildly more
    assert err_ref[-1] <= max(1e-13, 50 * spreads[-1]), (name, err_ref[-1], spreads[-1])


def test_extended_precision_run_is_a_rounding_level_neighbour_of_the_double_run(oracle):
    """Well-conditioned input: the two runs of the same statements differ at the rounding level of double (the
    extended run carries 11 more bits through every sum), not at all in the decisions."""
    rng = np.random.default_rng(1)
    n, m = 4000, 5
    a, b = oracle.OracleNKA(n, m), oracle.OracleExact(n, m)
    worst = 0.0
    for t in range(12):
        x = rng.standard_normal(n)
        fa, fb = x.copy(), x.copy()
        a.accel_update(fa)
        b.accel_update(fb)
        assert a.state().list_order() == b.state().list_order() and a.num_vec() == b.num_vec()
        worst = max(worst, np.linalg.norm(fa - fb) / np.linalg.norm(x))
    assert 0 < worst < 5e-14


@pytest.mark.parametrize("n", [1, 5, 255, 512, 513, 1300, 70001])
def test_attribution_dots_are_the_sums_they_claim_to_be(oracle, n):
    import ctypes as C
    L = oracle.lib()
    for name in ("nka_oracle_dot_fma", "nka_oracle_dot_blocked", "nka_oracle_dot_device"):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [C.c_void_p, C.c_int64, oracle._dp, oracle._dp]
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    true = math.fsum(float(a) * float(b) for a, b in zip(x, y))          # (products rounded: good to ~n eps)
    scale = float(np.abs(x * y).sum())
    seq = 0.0
    for a, b in zip(x, y):
        seq += float(a) * float(b)
    for name in ("nka_oracle_dot_fma", "nka_oracle_dot_blocked", "nka_oracle_dot_device"):
        got = getattr(L, name)(None, n, oracle._ptr(x), oracle._ptr(y))
        assert abs(got - true) <= 4 * n * 2.2e-16 * scale + 1e-300, (name, got, true)
    # the blocked order WITHOUT fma on fewer than 512 elements: one product per thread, then the butterfly over
    # 256 "threads" in four wavefronts -- reproduce it literally
    if n <= 256:
        acc = np.zeros(256)
        acc[:n] = x * y
        waves = []
        for w in range(4):
            v = acc[64 * w:64 * w + 64].copy()
            off = 32
            while off >= 1:
                v[:off] += v[off:2 * off]
                off //= 2
            waves.append(v[0])
        want = ((waves[0] + waves[1]) + waves[2]) + waves[3]
        lanes = np.zeros(64)
        lanes[0] = want
        off = 32
        while off >= 1:
            lanes[:off] += lanes[off:2 * off]
            off //= 2
        assert L.nka_oracle_dot_blocked(None, n, oracle._ptr(x), oracle._ptr(y)) == lanes[0]


def test_raw_sums_switch_changes_last_bits_only_and_no_decision(oracle):
    rng = np.random.default_rng(5)
    n, m = 3000, 6
    a, b = oracle.OracleNKA(n, m), oracle.attribution_oracle(n, m, raw_sums=True, fma=True, blocked=True)
    basis = rng.standard_normal((3, n))
    worst = 0.0
    for t in range(20):
        x = rng.standard_normal(3) @ basis if t % 5 == 3 else rng.standard_normal(n)
        fa, fb = x.copy(), x.copy()
        a.accel_update(fa)
        b.accel_update(fb)
        assert a.state().list_order() == b.state().list_order()
        worst = max(worst, np.linalg.norm(fa - fb) / np.linalg.norm(x))
    assert 0 < worst < 1e-11


# ---- the truth of the parity rule, checked INDEPENDENTLY: 60-digit restatement in mpmath ------------------
@pytest.mark.parametrize("name", S.scenario_names())
def test_extended_precision_truth_agrees_with_an_independent_60_digit_restatement(oracle, name):
    """oracle/oracle_mp.py: the algorithm written a third time (other data structures, no shared code) in 60-digit
    arithmetic.  On every golden scenario the extended-precision run -- the `exact` of tests/parity_util.py -- takes the
    same decisions and returns the same vectors to FAR below the distances the rule measures: the truth's own uncertainty
    is at most a thousandth of the reference's distance from it (or the rounding of its double outputs)."""
    from oracle import oracle_mp
    g = S.load(name)
    n, m = int(g["n"]), int(g["mvec"])
    ext, ref = oracle.OracleExact(n, m), oracle_mp.MpNKA(n, m)
    orders = {"ext": [], "mp": []}
    outs_e, tr_e = S.replay(ext, g, after_update=lambda u, a: orders["ext"].append(a.state().list_order()))
    outs_m, tr_m = S.replay(ref, g, after_update=lambda u, a: orders["mp"].append(a.list_order()))
    assert np.array_equal(tr_e, tr_m) and np.array_equal(tr_e, g["num_vec"])
    assert orders["ext"] == orders["mp"]                                     # slot for slot
    _, err_ref = P.fixture_truth(g, oracle)
    ups = [int(i) for op, i, _ in g["ops"] if int(op) == 0]
    for u, idx in enumerate(ups):
        nx = max(float(np.linalg.norm(g["inputs"][idx])), 1e-300)
        d = float(np.linalg.norm(outs_e[u] - outs_m[u])) / nx
        assert d <= max(5e-16, 1e-3 * err_ref[u]), (name, u, d, err_ref[u])


@pytest.mark.parametrize("seed,n,m", [(3069, 3, 21), (985, 5, 21), (1610, 7, 8), (11, 40, 6), (12, 9, 3)])
def test_truth_on_random_sequences_of_the_sizes_where_the_soak_recorded_exceedances(oracle, seed, n, m):
    """The exceedances of the soak (tests/golden/soak_cases.json) sit at n <= 9: ill-conditioned by construction (a handful
    of elements, up to 21 vectors).  There the truth itself must be beyond doubt: random sequences of the soak's mix (fresh,
    dependent, repeated and zero inputs, relax, restart, set_vec_tol) through the extended-precision run and the 60-digit one."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_ops import array_ops
    from oracle import oracle_mp
    rng = np.random.default_rng(seed)
    ext, ref, dbl = oracle.OracleExact(n, m), oracle_mp.MpNKA(n, m), oracle.OracleNKA(n, m)
    err_ref = 0.0
    for step, op in enumerate(array_ops(rng, n, 60)):
        if op[0] == "update":
            x = op[1]
            fe, fm, fd = x.copy(), x.copy(), x.copy()
            ext.accel_update(fe); ref.accel_update(fm); dbl.accel_update(fd)
            nx = float(np.linalg.norm(x))
            if nx > 0 and ext.state().list_order() == dbl.state().list_order():
                err_ref = max(err_ref, float(np.linalg.norm(fd - fm)) / nx)
                d = float(np.linalg.norm(fe - fm)) / nx
                assert d <= max(5e-16, 1e-3 * err_ref), (seed, step, d, err_ref)
        elif op[0] == "relax":
            ext.relax(); ref.relax(); dbl.relax()
        elif op[0] == "restart":
            ext.restart(); ref.restart(); dbl.restart()
        elif op[0] == "set_vec_tol":
            ext.set_vec_tol(op[1]); ref.set_vec_tol(op[1]); dbl.set_vec_tol(op[1])
        assert ext.num_vec() == ref.num_vec(), (seed, step)
        assert ext.state().list_order() == ref.list_order(), (seed, step)
