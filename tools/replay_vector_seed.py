#!/usr/bin/env python3
"""Replays recorded seeds of the abstract-vector soak (tests/golden/soak_cases.json) twice -- the norm stage fused into the
inner-product pass (raw-sum Gram row) and on its own (the Gram row on the rounded pair: what NKA_HIP_SUMS_BLOCKED_ROUNDED
selects for the workspace) -- and prints both distances from the truth.  tools/replay_vector_seed.py [SEED ...]"""
import os, sys, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tools')
import parity_util as P, scenarios as S
from oracle import oracle_py as oracle
import fuzz_gpu
oracle.lib()
for seed in ([int(a) for a in sys.argv[1:]] or (500196, 220309, 20203)):
    for fuse in ("1", "0"):
        os.environ["NKA_FUZZ_VECTOR_FUSE"] = fuse
        with tempfile.TemporaryDirectory() as d:
            key = fuzz_gpu.one_seed_vector(seed, oracle, P, S, d, strict=False)
        r = P.WORST[key]
        print(key, "dev", r["err_dev_exact"], "ref", r["err_ref_exact"], "ratio %.2f" % (r["err_dev_exact"] / r["err_ref_exact"]), flush=True)
