#!/usr/bin/env python3
"""Which of the device path's deliberate deviations carries its distance from the reference?  (CPU only.)

The device's inner products differ from a literal transcription of src-F08/nka_type.F90:263-290, 371 in three
ways: (a) fused multiply-adds, (b) a blocked, tree-shaped summation order, (c) the Gram row
of the normalised new vector as fl(<d,w_k>/s) from raw sums.  This tool replays a call sequence through

  reference    the double restatement as pinned to the compiled reference (sequential sums, no FMA)
  +fma / +blocked / +raw_sums     the same with ONE deviation switched on (oracle/nka_oracle_probe.c)
  device-like  all three
  exact        the same statements in extended precision (oracle/nka_oracle_exact.c)

and prints, per variant, the largest ||f_variant - f_exact|| / ||f_in|| over the sequence, next to the largest
||f_variant - f_reference|| / ||f_in|| -- the quantity the parity tests used to bound by K x (reference spread).

  tools/error_attribution.py [--scenario S9_near_dependence] [--fuzz-seed 4] [--out profiles/r04/error_attribution.txt]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

VARIANTS = [("reference", {}), ("+fma", {"fma": True}), ("+blocked", {"blocked": True}), ("+raw_sums", {"raw_sums": True}),
            ("device-like", {"fma": True, "blocked": True, "raw_sums": True})]


def run(ops, n, m, flavor, O):
    accs = {name: O.attribution_oracle(n, m, flavor, **kw) for name, kw in VARIANTS}
    exact = O.OracleExact(n, m, flavor)
    e_exact = {name: 0.0 for name, _ in VARIANTS}
    e_ref = {name: 0.0 for name, _ in VARIANTS}
    where = {name: -1 for name, _ in VARIANTS}
    updates = 0
    for step, op in enumerate(ops):
        everyone = list(accs.values()) + [exact]
        if op[0] == "update":
            x = op[1]
            fx = x.copy()
            exact.accel_update(fx)
            outs = {}
            for name, a in accs.items():
                f = x.copy()
                a.accel_update(f)
                outs[name] = f
            updates += 1
            order = exact.state().list_order()
            for name, a in accs.items():
                if a.state().list_order() != order:
                    raise SystemExit(f"step {step}: '{name}' took another drop decision than the extended-precision run")
            nx = np.linalg.norm(x)
            if nx > 0:
                for name in accs:
                    d = np.linalg.norm(outs[name] - fx) / nx
                    if d > e_exact[name]:
                        e_exact[name], where[name] = d, step
                    e_ref[name] = max(e_ref[name], np.linalg.norm(outs[name] - outs["reference"]) / nx)
        elif op[0] == "relax":
            [a.relax() for a in everyone]
        elif op[0] == "restart":
            [a.restart() for a in everyone]
        elif op[0] == "set_vec_tol":
            [a.set_vec_tol(op[1]) for a in everyone]
    return e_exact, e_ref, where, updates


def scenario_ops(g):
    import scenarios as S
    for op, idx, val in g["ops"]:
        op, idx = int(op), int(idx)
        if op == S.OP_UPDATE:
            yield ("update", g["inputs"][idx].copy())
        elif op == S.OP_RESTART:
            yield ("restart",)
        elif op == S.OP_RELAX:
            yield ("relax",)
        elif op == S.OP_SET_VEC_TOL:
            yield ("set_vec_tol", float(val))


def report(title, res, out):
    e_exact, e_ref, where, updates = res
    lines = [f"## {title} ({updates} updates)",
             f"{'variant':<14s} {'max err vs EXACT':>18s} {'(at step)':>10s} {'max diff vs reference':>24s} {'err / err(reference)':>22s}"]
    for name, _ in VARIANTS:
        ratio = e_exact[name] / e_exact["reference"] if e_exact["reference"] > 0 else float("nan")
        lines.append(f"{name:<14s} {e_exact[name]:18.3e} {where[name]:10d} {e_ref[name]:24.3e} {ratio:22.2f}")
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        out.write(text + "\n")


def survey(nseeds, O, out):
    """The truth rule's statistics without a GPU: for fuzz seeds 0..N-1, the worst distance from the extended-precision
    trajectory of (a) the reference -- its three flavours, the largest -- and (b) the device-like arithmetic (all three
    deviations on), per sequence; prints the distribution of  max err_dev / (2 max err_ref)."""
    from fuzz_ops import array_ops, array_shape
    import parity_util as P
    rows = []
    for seed in range(nseeds):
        rng, n, m, flavor = array_shape(seed)
        ops = list(array_ops(rng, n, 120))
        refs = [O.OracleNKA(n, m, fl) for fl in (0, 1, 2)]
        dev = O.attribution_oracle(n, m, flavor, fma=True, blocked=True, raw_sums=True)
        exact = O.OracleExact(n, m)
        e_ref = e_dev = 0.0
        for op in ops:
            everyone = refs + [dev, exact]
            if op[0] == "update":
                x = op[1]
                fx = x.copy()
                exact.accel_update(fx)
                nx = np.linalg.norm(x)
                for a in refs:
                    f = x.copy()
                    a.accel_update(f)
                    if nx > 0:
                        e_ref = max(e_ref, np.linalg.norm(f - fx) / nx)
                f = x.copy()
                dev.accel_update(f)
                if nx > 0:
                    e_dev = max(e_dev, np.linalg.norm(f - fx) / nx)
            elif op[0] == "relax":
                [a.relax() for a in everyone]
            elif op[0] == "restart":
                [a.restart() for a in everyone]
            elif op[0] == "set_vec_tol":
                [a.set_vec_tol(op[1]) for a in everyone]
        rows.append((seed, n, m, e_dev, e_ref))
    lines = [f"## survey: fuzz seeds 0..{nseeds - 1}, device-like arithmetic against the truth rule (CPU emulation)",
             f"{'seed':>5s} {'n':>6s} {'m':>3s} {'max err_dev':>12s} {'max err_ref':>12s} {'allowance used':>14s}"]
    used = []
    for seed, n, m, ed, er in rows:
        r = ed / max(P.truth_factor(n, m) * er, 1e-300) if ed > 1e-12 else 0.0
        used.append((r, n))
        lines.append(f"{seed:5d} {n:6d} {m:3d} {ed:12.3e} {er:12.3e} {r:14.2f}")
    rs = sorted(r for r, _ in used)
    q = lambda p: rs[min(len(rs) - 1, int(p * len(rs)))]       # noqa: E731
    lines.append(f"# share of the allowance used (0 where err_dev <= 1e-12): median {q(0.5):.2f}, 90 % {q(0.9):.2f}, largest {q(1.0):.2f}; "
                 f"sequences above 1: {[ (round(r, 2), n) for r, n in used if r > 1]}")
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        out.write(text + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--survey", type=int, default=0, help="N > 0: the truth rule's statistics over fuzz seeds 0..N-1 (see survey)")
    ap.add_argument("--scenario", nargs="*", default=["S9_near_dependence", "S8_n7_m8"])
    ap.add_argument("--fuzz-seed", type=int, nargs="*", default=[4])
    ap.add_argument("--flavor", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import scenarios as S
    from fuzz_ops import array_ops, array_shape
    from oracle import oracle_py as O
    out = open(a.out, "w") if a.out else None
    if out:
        out.write("# tools/error_attribution.py -- see its docstring.  err = ||f - f_exact|| / ||f_in||, largest over the sequence.\n\n")
    for name in a.scenario:
        g = S.load(name)
        n, m = int(g["n"]), int(g["mvec"])
        report(f"fixture {name}: n={n} mvec={m} flavour {a.flavor}", run(scenario_ops(g), n, m, a.flavor, O), out)
    if a.survey > 0:
        survey(a.survey, O, out)
    for seed in a.fuzz_seed:
        rng, n, m, flavor = array_shape(seed)
        report(f"fuzz seed {seed}: n={n} mvec={m} flavour {flavor} (tools/fuzz_ops.py)", run(array_ops(rng, n, 120), n, m, flavor, O), out)


if __name__ == "__main__":
    main()
