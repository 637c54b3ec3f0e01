#!/usr/bin/env python3
"""One tree of this repository (its own package and library) timed at one size: tools/tree_ab.py TREE_ROOT N MVEC.
Run for two trees alternately on one box to compare rounds (profiles/r04/tree_ab_r03_r04.txt)."""
import os, sys, time
root = sys.argv[1]; n = int(float(sys.argv[2])); m = int(sys.argv[3])
sys.path.insert(0, root)
import torch, nka_amd
from nka_amd import synth
assert os.path.realpath(nka_amd.__file__).startswith(os.path.realpath(root)), nka_amd.__file__
acc = nka_amd.nka().init(n, m)
f = torch.empty(n, dtype=torch.float64, device="cuda")
step = 0
def one():
    global step
    synth.fill_torch(f, 12345, step, 0, n)          # a fresh, independent input every update: the subspace stays full
    acc.accel_update(f)
    step += 1
for t in range(m + 6): one()
assert acc.num_vec() == m, acc.num_vec()
acc.set_timing(64)
best = []
for rep in range(3):
    for t in range(40): one()
    torch.cuda.synchronize()
    ms = [acc.timing_ms(b) for b in range(30)]
    pa = sum(x[0] for x in ms) / 30; so = sum(x[1] for x in ms) / 30; pb = sum(x[2] for x in ms) / 30; wh = sum(x[3] for x in ms) / 30
    best.append((wh * 1e3, pa * 1e3, so * 1e3, pb * 1e3))
b = min(best)
print(f"{os.path.basename(os.path.realpath(root)) or 'repo':8s} n={n} m={m} (full subspace): device whole update {b[0]:7.1f} us  PA {b[1]:6.1f}  solve {b[2]:5.1f}  PB {b[3]:6.1f}", flush=True)
