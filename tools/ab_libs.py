#!/usr/bin/env python3
"""In-process A/B of BUILDS of the library (compile-time variants), optionally crossed with run-time switches.

One accelerator per build, all on the same device, fed from ONE pool of inputs; the variants are alternated in blocks
of K updates for R rounds (order reversed every other round), per-phase device times from the library's HIP events.

  tools/ab_libs.py --libs nka_amd/libnka_hip_diag.so nka_amd/libnka_hip_diag_ft.so \
                   --combos pb_reverse=0 pb_reverse=1 --flavor c --vlen 1.25e7 --mvec 20

Every (build, combo) pair is one variant.  Round 5: do f and the pending pair's w, which PA and PB both read, come out of
the Infinity Cache for PB when they are loaded with the default cache policy (libnka_hip_diag_ft.so:
-DNKA_F_TEMPORAL=1) and / or when PB walks its tiles in the reverse of PA's order (pb_reverse=1)?
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--combos", nargs="+", default=["pb_reverse=0"], help="key=value[,key=value...] per variant")
    ap.add_argument("--flavor", default="c", choices=["f08", "c", "f08vec"])
    ap.add_argument("--vlen", type=float, default=1.25e7)
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--check-bits", action="store_true", help="compare the outputs of all variants bit for bit during the fill")
    a = ap.parse_args()
    import torch
    import nka_amd
    from nka_amd import synth
    n, m = int(a.vlen), a.mvec
    fl = {"f08": nka_amd.FLAVOR_F08, "c": nka_amd.FLAVOR_C, "f08vec": nka_amd.FLAVOR_F08_VECTOR}[a.flavor]
    accs = [nka_amd.nka(lib=os.path.join(ROOT, p) if not os.path.isabs(p) else p).init(n, m, flavor=fl) for p in a.libs]
    P = m + 6                    # (distinct inputs for the whole fill: a repeated input would be dropped as dependent)
    pool = torch.empty((P, n + (n % 2)), dtype=torch.float64, device="cuda")
    work = torch.empty(n + (n % 2), dtype=torch.float64, device="cuda")

    def apply(acc, combo):
        for kv in combo.split(","):
            k_, v_ = kv.split("=")
            acc.set_tuning(k_, int(v_))

    # fill: every accelerator sees the same inputs; with --check-bits each variant's output is compared with the first's
    t = 0
    for j in range(P):
        synth.fill_torch(pool[j, :n], 12345, j, 0, n)
    nbad = 0
    for _ in range(m + 3):
        ref = None
        for i, acc in enumerate(accs):
            # (with --check-bits the builds run DIFFERENT combos on the same input, rotating over the updates)
            apply(acc, a.combos[(t + i) % len(a.combos)] if a.check_bits else a.combos[0])
            work[:n].copy_(pool[t % P, :n])
            acc.accel_update(work[:n])
            if a.check_bits:
                if ref is None:
                    ref = work[:n].clone()
                elif not torch.equal(ref, work[:n]):
                    nbad += 1
        t += 1
    for acc in accs:
        assert acc.num_vec() == m
    variants = [(i, c) for i in range(len(accs)) for c in a.combos]
    res = {v: {"PA": [], "PB": [], "all": []} for v in variants}
    for acc in accs:
        acc.set_timing(a.steps)
    for r in range(a.rounds):
        order = variants if r % 2 == 0 else list(reversed(variants))
        for v in order:
            acc = accs[v[0]]
            apply(acc, v[1])
            for s in range(a.steps + 1):         # (one update under the new setting before the timed ones)
                acc.accel_update(pool[t % P, :n])
                t += 1
            ph = [acc.timing_ms(b) for b in range(a.steps)]
            res[v]["PA"].append(statistics.mean(p[0] for p in ph))
            res[v]["PB"].append(statistics.mean(p[2] for p in ph))
            res[v]["all"].append(statistics.mean(p[3] for p in ph))
    print(f"in-process A/B of builds  flavor={a.flavor} n={n} m={m}  {a.rounds} rounds x {a.steps} updates per variant"
          + (f"  [bit check during the fill: {nbad} mismatches]" if a.check_bits else ""))
    base = statistics.mean(res[variants[0]]["all"])
    for v in variants:
        d = res[v]
        print(f"  {os.path.basename(a.libs[v[0]]):28s} {v[1]:14s} PA {statistics.mean(d['PA']):.4f}  PB {statistics.mean(d['PB']):.4f} "
              f"(min {min(d['PB']):.4f} sd {statistics.pstdev(d['PB']):.4f})  update {statistics.mean(d['all']):.4f} ms "
              f"({100.0 * (statistics.mean(d['all']) / base - 1.0):+.2f} %)")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
