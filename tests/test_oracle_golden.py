"""Pins the oracle (oracle/nka_oracle.c, the CPU restatement) to the reference:
  * the reference's own golden files (reference_output) through the restated
    example problem,
  * fixtures produced by the compiled reference (tests/golden/, made by
    oracle/make_golden.py),
  * and, where oracle/_ref exists (build container), the compiled reference
    live, bit for bit.
CPU only."""
import json
import os
import re

import numpy as np
import pytest

import scenarios as S


def _tables():
    with open(os.path.join(S.GOLD, "example_tables.json")) as fh:
        return json.load(fh)


def _golden_last_lines():
    txt = open(os.path.join(S.GOLD, "reference_output_F08.txt")).read().splitlines()
    return [ln for ln in txt if re.match(r"^\s*\d+:", ln)]


@pytest.mark.parametrize("mvec,nsweep,key,gold_idx", [
    (0, 2, "f08", 0), (5, 2, "f08 --nka-vec 5", 1), (5, 4, "f08 --sweeps 4 --nka-vec 5", 2)])
@pytest.mark.parametrize("flavor", [0, 1, 2])
def test_config1_example_reproduces_reference_output(oracle, mvec, nsweep, key, gold_idx, flavor):
    acc = oracle.OracleNKA(2500, mvec, flavor) if mvec else None
    rn, _ = oracle.example_solve(nsweep=nsweep, accel=acc)
    lines = [f"{0:3d}:{rn[0]:14.6E}"] + [oracle.format_example_line(i, rn[i], rn[0]) for i in range(1, len(rn))]
    # (1) the published golden: final line of src-F08/reference_output
    assert lines[-1] == _golden_last_lines()[gold_idx]
    # (2) every iteration of the table printed by the compiled reference here
    ref_lines = _tables()[key][1:]
    assert lines == ref_lines


def test_config1_matches_c_reference_output_scale_free_columns(oracle):
    """src-C/reference_output pins every iteration (Reduction and Rate columns
    are scale free; its norms carry an extra hx*hy factor)."""
    txt = open(os.path.join(S.GOLD, "reference_output_C.txt")).read().splitlines()
    rows = [ln for ln in txt if re.match(r"^\s*\d+:", ln)]
    accel_rows = rows[:27]   # iterations 0..26 of the accelerated solve
    plain_rows = rows[27:]   # iterations 0..367 without acceleration
    for mvec, gold in ((5, accel_rows), (0, plain_rows)):
        acc = oracle.OracleNKA(2500, mvec, oracle.C_FLAVOR) if mvec else None
        rn, _ = oracle.example_solve(accel=acc)
        assert len(rn) == len(gold)
        for i in range(1, len(rn)):
            mine = oracle.format_example_line(i, rn[i], rn[0])
            assert mine[18:] == gold[i][18:], (i, mine, gold[i])  # Reduction + Rate columns
            # norms: golden = mine / 2500 to the printed 7 digits
            assert float(gold[i][4:18]) == pytest.approx(rn[i] / 2500.0, rel=2e-6)


@pytest.mark.parametrize("name", S.scenario_names())
def test_scenarios_bit_exact_against_compiled_reference_fixtures(oracle, name):
    g = S.load(name)
    n, m = int(g["n"]), int(g["mvec"])
    for flavor, key in ((oracle.F08, "f_out_f08"), (oracle.C_FLAVOR, "f_out_c"), (oracle.F08_VECTOR, "f_out_f08vec")):
        if key not in g.files:
            continue
        acc = oracle.OracleNKA(n, m, flavor)
        states = []
        outs, trace = S.replay(acc, g, after_update=lambda u, a: states.append(a.state()))
        assert acc.defined()
        assert np.array_equal(trace, g["num_vec"]), (name, flavor)
        # bit for bit: same arithmetic, same order, no FMA on either side
        assert np.array_equal(outs, g[key]), (name, flavor, np.abs(outs - g[key]).max())
        if flavor == oracle.C_FLAVOR and "first" in g.files:
            for u, st in enumerate(states):
                assert (st.first, st.last, st.free) == (g["first"][u], g["last"][u], g["free"][u])
                assert st.subspace == bool(g["subspace"][u])
                assert np.array_equal(st.next, g["next"][u])
                order = st.list_order()
                for k in order:  # prev is defined for list members only
                    assert st.prev[k - 1] == g["prev"][u][k - 1]
                for i in order[1:] if st.pending else order:  # factor + raw entries of live slots
                    for j in order[1:] if st.pending else order:
                        assert st.h[i - 1, j - 1] == g["h"][u][i - 1, j - 1], (name, u, i, j)


def test_medium_case_against_f08_reference_fixture(oracle):
    g = np.load(os.path.join(S.GOLD, "medium_n100000_m10.npz"))
    n, m, calls = int(g["n"]), int(g["mvec"]), int(g["calls"])
    rng = np.random.Generator(np.random.PCG64(int(g["seed"])))
    acc = oracle.OracleNKA(n, m)
    probe = np.cos(np.arange(n) * 0.001)
    for t in range(calls):
        f = rng.random(n) * 2.0 - 1.0
        assert np.linalg.norm(f) == g["in_norm"][t]      # the generator is reproducible
        acc.accel_update(f)
        assert acc.num_vec() == g["num_vec"][t]
        assert np.array_equal(f[g["idx"]], g["out_samples"][t])
        assert np.linalg.norm(f) == g["out_norm"][t]
        assert float(f @ probe) == g["out_probe"][t]


def test_reference_api_surface_and_defaults(oracle):
    """src-F08/nka_type.F90:160 (vtol default), :185-246 (accessors)."""
    a = oracle.OracleNKA(10, 3)
    assert (a.vec_len(), a.max_vec(), a.num_vec(), a.vec_tol()) == (10, 3, 0, 0.01)
    assert a.defined()
    a.set_vec_tol(0.5)
    assert a.vec_tol() == 0.5
    f = np.arange(10.0)
    g = f.copy()
    a.accel_update(g)          # first call: f returned unchanged (nka_type.F90:263,366 both false)
    assert np.array_equal(f, g) and a.num_vec() == 0
    a.relax()                  # drops the pending pair
    assert a.num_vec() == 0 and a.state().first == 0 and a.defined()


def test_user_dot_product_hook(oracle):
    """set_dot_prod (src-F08/nka_type.F90:209-214): a pairwise dot changes
    roundings only; decisions and values stay within tolerance."""
    n, m = 257, 4
    X = oracle.lcg_vectors(8, n, seed=5)
    a, b = oracle.OracleNKA(n, m), oracle.OracleNKA(n, m)
    calls = []
    b.set_dot_prod(lambda x, y: (calls.append(1), float(np.dot(x, y)))[1])
    for t in range(8):
        f, g = X[t].copy(), X[t].copy()
        a.accel_update(f)
        b.accel_update(g)
        assert a.num_vec() == b.num_vec()
        assert np.linalg.norm(f - g) <= 1e-13 * np.linalg.norm(X[t])
    assert len(calls) > 0


@pytest.mark.skipif(not __import__("oracle.oracle_py", fromlist=["x"]).have_ref(),
                    reason="compiled reference (oracle/_ref) not built here")
def test_live_compiled_reference_bit_exact(oracle):
    n, m = 300, 6
    rng = np.random.default_rng(3)
    a, r = oracle.OracleNKA(n, m), oracle.RefF08(n, m)
    c, rc = oracle.OracleNKA(n, m, oracle.C_FLAVOR), oracle.RefC(n, m)
    basis = rng.standard_normal((3, n))
    for t in range(30):
        x = rng.standard_normal(n) if t % 3 else rng.standard_normal(3) @ basis
        f = [x.copy() for _ in range(4)]
        a.accel_update(f[0]); r.accel_update(f[1]); c.accel_update(f[2]); rc.accel_update(f[3])
        assert np.array_equal(f[0], f[1]) and np.array_equal(f[2], f[3])
        assert a.num_vec() == r.num_vec() == c.num_vec() == rc.num_vec()
        assert c.state().list_order() == rc.state().list_order()
        if t == 17:
            for o in (a, r, c, rc):
                o.relax()
        if t == 23:
            for o in (a, r, c, rc):
                o.restart()


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(S.GOLD), "..", "oracle", "_ref", "libnka_ref_c.so")),
                    reason="compiled reference absent")
def test_user_dot_product_call_sequence_is_pinned_to_the_compiled_reference(oracle):
    """The sequence of user-dp calls -- how many, in which order, on which operands -- is part of the reference's
    behaviour for a dp with side effects.  The oracle's set_dot_prod run against the COMPILED src-C reference
    constructed with the same dp (nka_init(..., dp), .c:211, 227-231): identical call sequences, operand by
    operand, and bit-identical outputs, through dependence drops, a repeated input (s == 0 -> relax), relax and
    restart.  (tests/test_hip_round2.py then holds the device path to the oracle's sequence.)"""
    n, m = 257, 4
    X = oracle.lcg_vectors(12, n, seed=3)
    basis = oracle.lcg_vectors(2, n, seed=4)
    seq_o, seq_r = [], []

    def fp(seq):
        def dp(x, y):
            x, y = np.asarray(x), np.asarray(y)
            seq.append((hash(x.tobytes()), hash(y.tobytes())))
            return float(np.add.reduce((x * y)[::-1]))          # not the default order: really used
        return dp

    ora = oracle.OracleNKA(n, m, oracle.C_FLAVOR)
    ora.set_dot_prod(fp(seq_o))
    ref = oracle.RefC(n, m, 0.01, dp=fp(seq_r))
    for t in range(20):
        if t % 4 == 3:
            x = (t + 1.0) * basis[0] - 0.5 * basis[1]
        elif t == 9:
            x = prev.copy()
        else:
            x = X[t % 12] * (1.0 + 0.1 * t)
        prev = x
        fo, fr = x.copy(), x.copy()
        ora.accel_update(fo)
        ref.accel_update(fr)
        assert np.array_equal(fo, fr), t
        assert ora.num_vec() == ref.num_vec(), t
        assert seq_o == seq_r, (t, len(seq_o), len(seq_r))
        if t == 12:
            ora.relax(); ref.relax()
        if t == 16:
            ora.restart(); ref.restart()
    assert len(seq_o) > 60
