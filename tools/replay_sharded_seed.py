#!/usr/bin/env python3
"""CPU only: the sharded soak's sequence of one seed (tools/fuzz_gpu.py one_seed_sharded: the same draws in the same order)
through the reference's three flavours, the attribution variants of oracle/nka_oracle_probe.c -- the reference's own double
arithmetic with ONE of the device's deviations switched on, and all three ("device-like") -- and the extended-precision
trajectory; prints every update's distance from that trajectory, per variant.  Answers, for a seed the soak flags, whether
the device's figure is its arithmetic (any re-ordering of the sums lands there) or a defect of the path that ran.
  tools/replay_sharded_seed.py SEED [LAST_STEP]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

seed, steps = int(sys.argv[1]), 60
last = int(sys.argv[2]) if len(sys.argv) > 2 else steps - 1
rng = np.random.default_rng(90_000 + seed)
n = int(rng.choice([1, 2, 3, 4, 5, 7, 255, 512, 513, 1025, 2049, 4099])) if rng.random() < 0.7 else int(rng.integers(1, 9000))
m = int(rng.integers(1, 25))
flavor = int(rng.integers(0, 3))
print(f"# sharded soak seed {seed}: n={n} mvec={m} flavor {flavor}; ||f_variant - f_exact|| / ||f_in|| per update")
refs = {f"ref{fl}": O.OracleNKA(n, m, fl) for fl in (0, 1, 2)}
var = {"+fma": O.attribution_oracle(n, m, flavor, fma=True), "+blocked": O.attribution_oracle(n, m, flavor, blocked=True),
       "+raw_sums": O.attribution_oracle(n, m, flavor, raw_sums=True),
       "device-like": O.attribution_oracle(n, m, flavor, fma=True, blocked=True, raw_sums=True)}
exact = O.OracleExact(n, m, flavor)
everyone = list(refs.values()) + list(var.values()) + [exact]
names = list(refs) + list(var)
print(f"{'step':>4s} {'input':<7s} {'nvec':>4s} " + " ".join(f"{k:>11s}" for k in names))
worst = {k: 0.0 for k in names}
basis = rng.standard_normal((3, n))
prev = rng.standard_normal(n)
for step in range(last + 1):
    r = rng.random()
    if r < 0.80:
        kind = rng.random()
        if kind < 0.55:
            x, what = rng.standard_normal(n), "fresh"
        elif kind < 0.85:
            x, what = rng.standard_normal(3) @ basis, "span-3"
        elif kind < 0.95:
            x, what = prev.copy(), "repeat"
        else:
            x, what = np.zeros(n), "zero"
        prev = x
        fx = x.copy()
        exact.accel_update(fx)
        nx = np.linalg.norm(x)
        row = {}
        for name, a in list(refs.items()) + list(var.items()):
            f = x.copy()
            a.accel_update(f)
            assert a.state().list_order() == exact.state().list_order(), (step, name)
            row[name] = np.linalg.norm(f - fx) / nx if nx > 0 else 0.0
            worst[name] = max(worst[name], row[name])
        print(f"{step:4d} {what:<7s} {exact.num_vec():4d} " + " ".join(f"{row[k]:11.2e}" for k in names))
    elif r < 0.87:
        [a.relax() for a in everyone]
        print(f"{step:4d} relax")
    elif r < 0.91:
        [a.restart() for a in everyone]
        print(f"{step:4d} restart")
    elif r < 0.96:
        vt = float(10.0 ** rng.uniform(-3, -0.3))
        [a.set_vec_tol(vt) for a in everyone]
        print(f"{step:4d} set_vec_tol {vt:.3g}")
    else:
        print(f"{step:4d} deep copy")
print(f"{'max':>4s} {'':<7s} {'':>4s} " + " ".join(f"{worst[k]:11.2e}" for k in names))
