#!/usr/bin/env python3
"""One soak seed (tools/fuzz_ops.py) twice through the C ABI: every update in place, and with the out-of-place updates the soak
tool mixes into that seed -- every output and the state digest after every operation compared BIT FOR BIT.
  tools/swap_vs_inplace_seed.py SEED [SEED ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import nka_amd  # noqa: E402
from fuzz_ops import array_ops, array_shape  # noqa: E402

for seed in [int(s) for s in sys.argv[1:]]:
    rng, n, m, flavor = array_shape(seed)
    ops = list(array_ops(rng, n, 120))
    rng_swap = np.random.default_rng(777_000 + seed)
    share = float(rng_swap.choice([0.3, 0.7, 1.0]))
    a, b = nka_amd.nka().init(n, m, flavor=flavor), nka_amd.nka().init(n, m, flavor=flavor)
    nswap = 0
    for step, op in enumerate(ops):
        if op[0] == "update":
            fa, fb = torch.from_numpy(op[1].copy()).cuda(), torch.from_numpy(op[1].copy()).cuda()
            a.accel_update(fa)
            if rng_swap.random() < share:
                _, fb = b.accel_update_swap(fb)
                nswap += 1
                if rng_swap.random() < 0.5:
                    torch.cuda.synchronize()
            else:
                b.accel_update(fb)
            assert torch.equal(fa, fb), (seed, step, float((fa - fb).abs().max()))
        elif op[0] == "relax":
            a.relax(); b.relax()
        elif op[0] == "restart":
            a.restart(); b.restart()
        elif op[0] == "set_vec_tol":
            a.set_vec_tol(op[1]); b.set_vec_tol(op[1])
        else:
            a, b = a.copy(), b.copy()
        assert a.state_digest() == b.state_digest(), (seed, step, op[0])
    print(f"seed {seed}: n={n} m={m} flavor {flavor}, {len(ops)} operations, {nswap} updates out of place (share {share}): every output and every "
          f"state digest bit-identical to the in-place run", flush=True)
