#!/usr/bin/env python3
"""In-process A/B of the slot skew (NKA_HIP_SLOT_PAD_BYTES): one accelerator per
pad value in ONE process (same inputs, same thermal state), alternated in blocks
of K updates for R rounds; per-phase device times from the library's HIP events.

  tools/ab_pad.py --pads 0 256 2048 [--flavor c] [--vlen 1e8] [--mvec 20] [--rounds 6] [--steps 10]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pads", type=int, nargs="+", default=[0, 256, 2048])
    ap.add_argument("--flavor", default="c", choices=["f08", "c", "f08vec"])
    ap.add_argument("--vlen", type=float, default=1e8)
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import nka_amd
    from nka_amd import synth
    n, m = int(a.vlen), a.mvec
    fl = {"f08": nka_amd.FLAVOR_F08, "c": nka_amd.FLAVOR_C, "f08vec": nka_amd.FLAVOR_F08_VECTOR}[a.flavor]
    accs = {}
    for p in a.pads:
        os.environ["NKA_HIP_SLOT_PAD_BYTES"] = str(p)
        accs[p] = nka_amd.nka().init(n, m, flavor=fl)
    P = m + 4
    pool = torch.empty((P, n + (n % 2)), dtype=torch.float64, device="cuda")
    f = torch.empty(n, dtype=torch.float64, device="cuda")
    for j in range(P):
        synth.fill_torch(pool[j, :n], 12345, j, 0, n)
    t = {p: 0 for p in a.pads}

    def step(p):
        f.copy_(pool[t[p] % P, :n])
        accs[p].accel_update(f)
        t[p] += 1

    for p in a.pads:
        for _ in range(m + 3):
            step(p)
        assert accs[p].num_vec() == m
        accs[p].set_timing(a.steps)
    res = {p: {"PA": [], "PB": [], "all": []} for p in a.pads}
    for r in range(a.rounds):
        order = a.pads if r % 2 == 0 else list(reversed(a.pads))
        for p in order:
            for _ in range(a.steps):
                step(p)
            ph = [accs[p].timing_ms(b) for b in range(a.steps)]
            res[p]["PA"].append(statistics.mean(x[0] for x in ph))
            res[p]["PB"].append(statistics.mean(x[2] for x in ph))
            res[p]["all"].append(statistics.mean(x[3] for x in ph))
    print(f"in-process slot-pad A/B  flavor={a.flavor} n={n} m={m}  {a.rounds} rounds x {a.steps} updates per pad; f at {f.data_ptr() % 4096} mod 4096")
    for p in a.pads:
        d = res[p]
        print(f"  pad={p}:  PB {statistics.mean(d['PB']):.3f} ms (min {min(d['PB']):.3f}, max {max(d['PB']):.3f})"
              f"   PA {statistics.mean(d['PA']):.3f} (min {min(d['PA']):.3f})   update {statistics.mean(d['all']):.3f} ms")


if __name__ == "__main__":
    main()
