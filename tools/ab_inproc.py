#!/usr/bin/env python3
"""In-process A/B of kernel variants: ONE accelerator (same allocations, same
inputs, same thermal state), the variants alternated in blocks of K updates for
R rounds; per-phase device times from the library's HIP events.

  tools/ab_inproc.py --key pb_pipe --values 0 201 [--flavor f08] [--vlen 1e8] [--mvec 20] [--rounds 8] [--steps 10]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", default="pb_pipe")
    ap.add_argument("--values", type=int, nargs="+", default=[0, 201])
    ap.add_argument("--combos", nargs="+", default=None,
                    help="variants as key=value[,key=value...] (overrides --key/--values), e.g. pb_tickets=1,pb_tile=2")
    ap.add_argument("--flavor", default="f08", choices=["f08", "c", "f08vec"])
    ap.add_argument("--vlen", type=float, default=1e8)
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--span-dim", type=int, default=0,
                    help="D > 0: every input in a D-dimensional span (a dependence drop per update, the subspace holds D "
                         "vectors) and one synchronisation per update, e.g. --key list_word --values 0 1")
    a = ap.parse_args()
    import torch
    import nka_amd
    from nka_amd import synth
    n, m = int(a.vlen), a.mvec
    fl = {"f08": nka_amd.FLAVOR_F08, "c": nka_amd.FLAVOR_C, "f08vec": nka_amd.FLAVOR_F08_VECTOR}[a.flavor]
    acc = nka_amd.nka(diagnostic=True).init(n, m, flavor=fl)      # libnka_hip_diag.so: the A/B switches
    P = min(m + 6, 30)
    pool = torch.empty((P, n + (n % 2)), dtype=torch.float64, device="cuda")
    D = a.span_dim
    if D > 0:
        B = torch.empty((D, n), dtype=torch.float64, device="cuda")
        for q in range(D):
            synth.fill_torch(B[q], 777, q, 0, n)

    def fill(j, t):
        if D > 0:
            coef = torch.from_numpy(synth.fill_numpy(60, t, 0, D, D)).cuda()
            torch.mv(B.t(), coef, out=pool[j, :n])
        else:
            synth.fill_torch(pool[j, :n], 12345, t, 0, n)

    def step(t):
        fill(t % P, t)
        acc.accel_update(pool[t % P, :n])
        if D > 0:
            torch.cuda.synchronize()

    k_want = min(D, m) if D > 0 else m
    t = 0
    for _ in range(m + 3):
        step(t)
        t += 1
    assert acc.num_vec() == k_want
    if a.combos:
        a.values = a.combos
        a.key = "combo"
    res = {v: {"PA": [], "PB": [], "all": []} for v in a.values}
    acc.set_timing(a.steps)
    for r in range(a.rounds):
        order = a.values if r % 2 == 0 else list(reversed(a.values))
        for v in order:
            if a.combos:
                for kv in v.split(","):
                    k_, v_ = kv.split("=")
                    acc.set_tuning(k_, int(v_))
            elif a.key == "sum_order":       # (a product switch, nka_hip_set_sum_order: 2 = blocked, 3 = blocked on the rounded w1')
                acc.set_sum_order(v)
            else:
                acc.set_tuning(a.key, v)
            step(t)                     # (one update under the new setting before the timed ones)
            t += 1
            for _ in range(a.steps):
                step(t)
                t += 1
            ph = [acc.timing_ms(b) for b in range(a.steps)]
            res[v]["PA"].append(statistics.mean(p[0] for p in ph))
            res[v]["PB"].append(statistics.mean(p[2] for p in ph))
            res[v]["all"].append(statistics.mean(p[3] for p in ph))
    assert acc.num_vec() == k_want
    if D > 0:
        print(f"inputs in a {D}-dimensional span: num_vec = {k_want} of mvec = {m}; host bound on the list: {acc.list_bound()}")
    print(f"in-process A/B  key={a.key}  flavor={a.flavor} n={n} m={m}  {a.rounds} rounds x {a.steps} updates per variant")
    for v in a.values:
        d = res[v]
        print(f"  {a.key}={v}:  PB {statistics.mean(d['PB']):.3f} ms (min {min(d['PB']):.3f}, max {max(d['PB']):.3f}, "
              f"sd {statistics.pstdev(d['PB']):.3f})   PA {statistics.mean(d['PA']):.3f}   update {statistics.mean(d['all']):.3f} ms")


if __name__ == "__main__":
    main()
