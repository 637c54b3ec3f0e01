#!/usr/bin/env python3
"""What reference-order sums cost (nka_hip_set_sum_order): microseconds per update, blocked passes against
k_dots_ordered, full subspace, wall clock over a run of back-to-back updates with one synchronisation at the end.
  tools/sum_order_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nka_amd  # noqa: E402
from nka_amd import synth  # noqa: E402


SPAN = 0          # tools/sum_order_cost.py N MVEC REPS SPAN: every input in the span of SPAN fixed random vectors (correlated
BASIS = None      # inputs: the inner products are sums that DRIFT instead of zero-mean random walks; a dependence drop per update)


def make_input(buf, t, n):
    global BASIS
    if SPAN <= 0:
        synth.fill_torch(buf, 4321, t, 0, n)
        return
    if BASIS is None or BASIS.shape[1] != n:
        BASIS = torch.empty((SPAN, n), dtype=torch.float64, device="cuda")
        for q in range(SPAN):
            synth.fill_torch(BASIS[q], 8765, q, 0, n)
    coef = torch.from_numpy(synth.fill_numpy(99, t, 0, SPAN, SPAN)).cuda()
    torch.mv(BASIS.t(), coef, out=buf)


def us_per_update(n, m, order, reps):
    """Every input is a fresh vector of the generator (repeating a pool would make the differences dependent: the subspace
    would shrink to the pool's size and both modes would look cheaper than they are at mvec vectors)."""
    acc = nka_amd.nka().init(n, m).set_sum_order(order)
    buf = torch.empty(n, dtype=torch.float64, device="cuda")
    for t in range(m + 4):
        make_input(buf, t, n)
        acc.accel_update(buf)
    work = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(reps)]
    for r, w in enumerate(work):
        make_input(w, m + 4 + r, n)
    torch.cuda.synchronize()
    if acc.num_vec() != m and SPAN <= 0:
        print(f"# n={n} m={m}: the subspace holds {acc.num_vec()} vectors", flush=True)
    t0 = time.perf_counter()
    for r in range(reps):
        acc.accel_update(work[r])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


print(f"{'n':>9s} {'mvec':>5s} {'blocked us':>11s} {'reference-order us':>19s}")
CASES = ((64, 5, 2000), (64, 20, 2000), (512, 5, 2000), (512, 20, 2000), (512, 40, 1000), (4096, 20, 500), (10**4, 20, 300),
         (10**5, 20, 100), (10**6, 20, 20))
if len(sys.argv) > 1:                      # tools/sum_order_cost.py N MVEC REPS [SPAN]   (one case, e.g. under rocprofv3)
    CASES = ((int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])),)
    if len(sys.argv) > 4:
        SPAN = int(sys.argv[4])
        print(f"# every input in the span of {SPAN} fixed vectors")
for n, m, reps in CASES:
    b = us_per_update(n, m, nka_amd.SUMS_BLOCKED, reps)
    r = us_per_update(n, m, nka_amd.SUMS_REFERENCE_ORDER, max(5, reps // (1 if n <= 4096 else 4)))
    print(f"{n:9d} {m:5d} {b:11.1f} {r:19.1f}", flush=True)
