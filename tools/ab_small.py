#!/usr/bin/env python3
"""Small-n A/B by WALL CLOCK: where an update is a few launches long, per-phase HIP events
distort what they measure (four event records widen the kernel boundaries by ~15 us).  One
accelerator per vector length, the variants of ONE tuning key alternated in blocks of B updates
on fresh copies of the same B inputs, each block bracketed by stream synchronisation:
microseconds per update = block wall time / B (host enqueue included: at these sizes it can be
the limit, which is part of the answer).

  tools/ab_small.py --key pa_pipe --values 0 201 --vlens 1e4 1e5 1e6 1e7 1.25e7 --mvecs 5 10 20
"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", default="pa_pipe")
    ap.add_argument("--values", type=int, nargs="+", default=[0, 201])
    ap.add_argument("--vlens", type=float, nargs="+", default=[1e4, 1e5, 1e6, 1e7, 1.25e7])
    ap.add_argument("--mvecs", type=int, nargs="+", default=[5, 10, 20])
    ap.add_argument("--flavor", default="default", choices=["default", "f08", "c", "f08vec"])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--block", type=int, default=48)
    a = ap.parse_args()
    import torch
    import nka_amd
    from nka_amd import synth
    fl = {"default": nka_amd.FLAVOR_DEFAULT, "f08": nka_amd.FLAVOR_F08, "c": nka_amd.FLAVOR_C,
          "f08vec": nka_amd.FLAVOR_F08_VECTOR}[a.flavor]
    print(f"# key={a.key} values={a.values} flavor={a.flavor} block={a.block} rounds={a.rounds}: us per update (median of rounds; min)")
    for m in a.mvecs:
        for nf in a.vlens:
            n = int(nf)
            B = a.block if n <= 2e7 else 16
            acc = nka_amd.nka(diagnostic=True).init(n, m, flavor=fl)
            pool0 = torch.empty((B, n + (n % 2)), dtype=torch.float64, device="cuda")
            for j in range(B):
                synth.fill_torch(pool0[j, :n], 12345, j, 0, n)
            pool = pool0.clone()
            for j in range(m + 3):
                acc.accel_update(pool[j % B, :n])
            assert acc.num_vec() == m
            res = {v: [] for v in a.values}
            for r in range(a.rounds):
                order = a.values if r % 2 == 0 else list(reversed(a.values))
                for v in order:
                    acc.set_tuning(a.key, v)
                    pool.copy_(pool0)
                    acc.accel_update(pool[0, :n])          # first launch of a variant outside the timed block
                    pool[0].copy_(pool0[0])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for j in range(B):
                        acc.accel_update(pool[j, :n])
                    torch.cuda.synchronize()
                    res[v].append((time.perf_counter() - t0) / B * 1e6)
            assert acc.num_vec() == m
            line = f"n={n:>9d} m={m:>2d} "
            base = statistics.median(res[a.values[0]])
            for v in a.values:
                med = statistics.median(res[v])
                line += f" | {a.key}={v}: {med:8.1f} (min {min(res[v]):8.1f})"
            line += f" | {100 * (statistics.median(res[a.values[-1]]) / base - 1):+5.1f} %"
            print(line, flush=True)
            acc.delete()
            del pool, pool0
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
