#!/usr/bin/env python3
"""In-place against out-of-place updates (nka_hip_accel_update vs nka_hip_accel_update_swap): two accelerators on the
same inputs in one process, blocks of K updates alternated for R rounds; per-phase device times from the library's
HIP events; one synchronisation per update on both sides (the out-of-place entry learns the displaced buffers from
the list word's record without waiting only if the caller has synchronised since).

  tools/ab_swap.py [--flavor c] [--vlen 1e8] [--mvec 20] [--rounds 6] [--steps 10]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flavor", default="c", choices=["f08", "c", "f08vec"])
    ap.add_argument("--vlen", type=float, default=1e8)
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import time
    import torch
    import nka_amd
    from nka_amd import synth
    n, m = int(a.vlen), a.mvec
    fl = {"f08": nka_amd.FLAVOR_F08, "c": nka_amd.FLAVOR_C, "f08vec": nka_amd.FLAVOR_F08_VECTOR}[a.flavor]
    inp = nka_amd.nka().init(n, m, flavor=fl)
    oop = nka_amd.nka().init(n, m, flavor=fl)
    x = torch.empty(n, dtype=torch.float64, device="cuda")          # the input of the step
    f = torch.empty(n, dtype=torch.float64, device="cuda")          # in-place side
    buf = torch.empty(n, dtype=torch.float64, device="cuda")        # out-of-place side: the buffer the caller owns
    acc = None

    def step(t, which):
        nonlocal buf, acc
        synth.fill_torch(x, 12345, t, 0, n)
        if which == "in":
            f.copy_(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inp.accel_update(f)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        buf.copy_(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        buf, acc = oop.accel_update_swap(buf)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    t = 0
    for _ in range(m + 3):
        step(t, "in"); step(t, "out")
        assert torch.equal(f, acc), t                      # same bits
        t += 1
    assert inp.num_vec() == m and oop.num_vec() == m and inp.state_digest() == oop.state_digest()
    res = {w: {"PA": [], "PB": [], "all": [], "wall": []} for w in ("in", "out")}
    inp.set_timing(a.steps)
    oop.set_timing(a.steps)
    for r in range(a.rounds):
        for w in (("in", "out") if r % 2 == 0 else ("out", "in")):
            obj = inp if w == "in" else oop
            walls = []
            for k in range(a.steps):
                walls.append(step(t + k, w))
            ph = [obj.timing_ms(b) for b in range(a.steps)]
            res[w]["PA"].append(statistics.mean(p[0] for p in ph))
            res[w]["PB"].append(statistics.mean(p[2] for p in ph))
            res[w]["all"].append(statistics.mean(p[3] for p in ph))
            res[w]["wall"].append(1e3 * statistics.mean(walls))
        # keep the two accelerators in lock step: the other side runs the same inputs untimed
        t += a.steps
    # (each side saw its own inputs only; states differ, traffic does not)
    words = {"c": (27, 25), "f08": (46, 44), "f08vec": (46, 44)}[a.flavor]
    print(f"in place vs out of place  flavor={a.flavor} n={n} m={m}  {a.rounds} rounds x {a.steps} updates, one synchronisation per update")
    for w, nm, wd in (("in", "nka_hip_accel_update      ", words[0]), ("out", "nka_hip_accel_update_swap ", words[1])):
        d = res[w]
        pb = statistics.mean(d["PB"])
        print(f"  {nm}: PB {pb:.3f} ms ({wd} words -> {8e-9 * n * wd / (pb * 1e-3):.0f} GB/s; min {min(d['PB']):.3f}, max {max(d['PB']):.3f})   "
              f"PA {statistics.mean(d['PA']):.3f}   update {statistics.mean(d['all']):.3f} ms device, {statistics.mean(d['wall']):.3f} ms wall")
    g = 1.0 - statistics.mean(res["out"]["all"]) / statistics.mean(res["in"]["all"])
    print(f"  out of place: {100 * g:+.1f} % per update (device time)")


if __name__ == "__main__":
    main()
