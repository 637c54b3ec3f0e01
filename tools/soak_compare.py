#!/usr/bin/env python3
"""Round 6: the two fast sum modes on the SAME seeds, side by side.

    tools/soak_compare.py OUT.txt DIR        (DIR = gpurun_out/paired: fuzz_<kind>_<first>_{blocked,rounded}.txt[.rank<r>])

`tools/evidence.sh soak-paired` runs every seed of a span once with NKA_FUZZ_FORCE_SUMS=blocked (the fast passes as rounds 1-5 ran them by default: the
Gram row taken as fl(<d,w_k>/s) from raw sums) and once with =rounded (NKA_HIP_SUMS_BLOCKED_ROUNDED: the norm first, the Gram
row as the inner product of the STORED fl(d/s), /root/reference/src-F08/nka_type.F90:282-290).  A record = one sequence on one
rank, paired by (kind, seed, rank).  Judged as tests/parity_util.py judges: per sequence, err_dev <= max(1e-12, F err_ref)
against the extended-precision trajectory, F = 2 beyond one tile (n > 512), 4 within.  The table the default is re-decided
from: exceedances per mode and size class, the distribution of err_dev / err_ref, how often a seed is beyond the allowance in
one mode and not in the other, and the smallest per-sequence factor that NO record of either mode exceeds."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as P  # noqa: E402

BASE = 1e-12
CLASSES = ((1, 16), (17, 512), (513, 2048), (2049, 10**9))


def parse(path):
    """-> {seed: (err_dev, err_ref, n, mvec, stop_tripped, line)}"""
    out = {}
    for ln in open(path, errors="replace"):
        if not ln.startswith(("ok", "stop")):
            continue
        m = re.search(r"dev-exact ([0-9.e+-]+) ref-exact ([0-9.e+-]+)", ln)
        s = re.search(r"seed (\d+)", ln)
        if not m or not s:
            continue
        k = re.search(r" n=(\d+) m=(\d+)", ln)
        if k:
            n, mv = int(k.group(1)), int(k.group(2))
        else:
            k = re.search(r" (\d+)x(\d+) m=(\d+)", ln)
            n, mv = int(k.group(1)) * int(k.group(2)), int(k.group(3))
        out[int(s.group(1))] = (float(m.group(1)), float(m.group(2)), n, mv, ln.startswith("stop"), ln.strip())
    return out


def label(lo, hi):
    return f"{lo}..{hi}" if hi < 10**9 else f"> {lo - 1}"


def beyond(rec):
    dev, ref, n, mv = rec[:4]
    return dev > max(BASE, P.truth_factor(n, mv) * ref)


def main():
    out, root = sys.argv[1], sys.argv[2]
    pairs = {}          # (kind, first, rank) -> {mode: records}
    for f in sorted(glob.glob(os.path.join(root, "fuzz_*_*_*.txt*"))):
        m = re.match(r"fuzz_([a-z-]+)_(\d+)_(blocked|rounded)\.txt(?:\.rank(\d+))?$", os.path.basename(f))
        if not m:
            continue
        kind, first, mode, rank = m.group(1), int(m.group(2)), m.group(3), int(m.group(4) or 0)
        if kind in ("sharded",) and m.group(4) is None:
            continue        # (the sharded runs write one file per rank)
        pairs.setdefault((kind, first, rank), {})[mode] = parse(f)
    L = []
    L.append("# tools/soak_compare.py: the two fast sum modes on the same seeds (tools/evidence.sh soak-paired; NKA_FUZZ_FORCE_SUMS)")
    L.append("#   blocked = NKA_HIP_SUMS_BLOCKED, the default of rounds 1-5 (opt-in fast mode since): Gram row fl(<d,w_k>/s) from raw sums, one pass, one exchange")
    L.append("#   rounded = NKA_HIP_SUMS_BLOCKED_ROUNDED (the default since round 6): the norm first, Gram row = inner product of the stored fl(d/s) (F08:282-290)")
    L.append(f"# rule per sequence: err_dev <= max({BASE:g}, F x err_ref) against the extended-precision trajectory, F = {P.TRUTH_FACTOR:g} beyond "
             f"{P.TINY_N} elements, {P.TRUTH_FACTOR_TINY:g} within; a record = one sequence on one rank")
    L.append("")
    allrec = {"blocked": [], "rounded": []}
    both = []           # (kind, seed, rank, rec_blocked, rec_rounded)
    L.append(f"{'kind':>15s} {'first seed':>10s} {'rank':>4s} {'paired':>7s} | {'beyond: blocked':>15s} {'rounded':>8s} | per-call stop tripped: blocked rounded")
    for (kind, first, rank), modes in sorted(pairs.items()):
        b, r = modes.get("blocked", {}), modes.get("rounded", {})
        seeds = sorted(set(b) & set(r))
        for s in seeds:
            both.append((kind, s, rank, b[s], r[s]))
            allrec["blocked"].append(b[s])
            allrec["rounded"].append(r[s])
        L.append(f"{kind:>15s} {first:10d} {rank:4d} {len(seeds):7d} | {sum(beyond(b[s]) for s in seeds):15d} {sum(beyond(r[s]) for s in seeds):8d} | "
                 f"{sum(b[s][4] for s in seeds):28d} {sum(r[s][4] for s in seeds):7d}"
                 + (f"   (unpaired: {len(set(b) ^ set(r))})" if set(b) ^ set(r) else ""))
    n_rec = len(both)
    L.append("")
    L.append(f"## {n_rec} paired records ({len({(k, s) for k, s, _, _, _ in both})} sequences); records beyond the allowance, by vector length")
    L.append(f"{'elements':>12s} {'records':>8s} | {'blocked':>8s} {'rounded':>8s} | {'only blocked':>12s} {'only rounded':>12s} {'both':>5s}")
    tot = [0, 0]
    for lo, hi in CLASSES:
        sel = [x for x in both if lo <= x[3][2] <= hi]
        nb = sum(beyond(x[3]) for x in sel)
        nr = sum(beyond(x[4]) for x in sel)
        ob = sum(beyond(x[3]) and not beyond(x[4]) for x in sel)
        orr = sum(beyond(x[4]) and not beyond(x[3]) for x in sel)
        bb = sum(beyond(x[3]) and beyond(x[4]) for x in sel)
        if lo > 512:
            tot[0] += nb
            tot[1] += nr
        L.append(f"{label(lo, hi):>12s} {len(sel):8d} | {nb:8d} {nr:8d} | {ob:12d} {orr:12d} {bb:5d}")
    L.append(f"beyond one tile (> 512 elements): blocked {tot[0]}, rounded {tot[1]}"
             + (f" -- ratio {tot[0] / tot[1]:.2f}" if tot[1] else " -- rounded: none" if tot[0] else ""))
    L.append("")
    for mode in ("blocked", "rounded"):
        L.append(f"## {mode}: err_dev / err_ref of the records that end with err_dev > {BASE:g}")
        L.append(f"{'elements':>12s} {'records':>8s} {'with ratio':>10s} {'median':>7s} {'90 %':>6s} {'99 %':>6s} {'max':>8s} {'> 2':>5s} {'> 4':>5s} {'> 8':>5s}")
        for lo, hi in CLASSES:
            sel = [r for r in allrec[mode] if lo <= r[2] <= hi]
            rat = sorted(r[0] / r[1] for r in sel if r[0] > BASE and r[1] > 0)
            if rat:
                q = lambda p: rat[min(len(rat) - 1, int(p * len(rat)))]      # noqa: E731
                L.append(f"{label(lo, hi):>12s} {len(sel):8d} {len(rat):10d} {q(.5):7.2f} {q(.9):6.2f} {q(.99):6.2f} {rat[-1]:8.2f} "
                         f"{sum(x > 2 for x in rat):5d} {sum(x > 4 for x in rat):5d} {sum(x > 8 for x in rat):5d}")
            else:
                L.append(f"{label(lo, hi):>12s} {len(sel):8d} {0:10d}")
        L.append("")
    L.append("## the two modes against EACH OTHER on the same sequence (records in which either ends above the base)")
    L.append(f"{'elements':>12s} {'records':>8s} {'blocked further from the truth':>31s} {'rounded further':>16s} {'median err_blocked/err_rounded':>31s}")
    for lo, hi in CLASSES:
        sel = [x for x in both if lo <= x[3][2] <= hi and max(x[3][0], x[4][0]) > BASE and min(x[3][0], x[4][0]) > 0]
        if not sel:
            L.append(f"{label(lo, hi):>12s} {0:8d}")
            continue
        q = sorted(x[3][0] / x[4][0] for x in sel)
        L.append(f"{label(lo, hi):>12s} {len(sel):8d} {sum(x[3][0] > x[4][0] for x in sel):31d} {sum(x[4][0] > x[3][0] for x in sel):16d} {q[len(q) // 2]:31.2f}")
    L.append("")
    # the bound that no record exceeds: err_dev <= max(A x base, F x err_ref); smallest F per class with A = 1 and with A = 10
    L.append("## the smallest factor F with err_dev <= max(A x 1e-12, F x err_ref) for EVERY record of the class (per sequence)")
    L.append(f"{'elements':>12s} | {'blocked: A=1':>13s} {'A=10':>8s} | {'rounded: A=1':>13s} {'A=10':>8s}")
    for lo, hi in CLASSES:
        row = f"{label(lo, hi):>12s} |"
        for mode in ("blocked", "rounded"):
            sel = [r for r in allrec[mode] if lo <= r[2] <= hi]
            for A in (1.0, 10.0):
                need = [r[0] / r[1] if r[1] > 0 else float("inf") for r in sel if r[0] > A * BASE]
                row += f" {max(need) if need else 0.0:{13 if A == 1.0 else 8}.2f}"
            row += " |"
        L.append(row)
    L.append("")
    L.append("## every record beyond the allowance with more than 512 elements")
    for kind, s, rank, b, r in both:
        if b[2] > 512 and (beyond(b) or beyond(r)):
            L.append(f"{kind} seed {s} rank {rank} n={b[2]} m={b[3]}: blocked {b[0]:.2e} ({b[0] / max(b[1], 1e-300):.2f} x err_ref)"
                     f"{' BEYOND' if beyond(b) else ''}; rounded {r[0]:.2e} ({r[0] / max(r[1], 1e-300):.2f} x){' BEYOND' if beyond(r) else ''}; err_ref {b[1]:.2e}")
    open(out, "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
