#!/usr/bin/env python3
"""Summarise the per-sequence records of soak runs (tools/fuzz_gpu.py --out files) under the truth rule of
tests/parity_util.py: share of the allowance used, sequences beyond it, and err_dev / err_ref by vector length.
  tools/soak_summary.py OUT.txt FILE [FILE ...]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as P  # noqa: E402


def records(path):
    for ln in open(path):
        if not ln.startswith(("ok", "stop")):
            continue
        m = re.search(r"dev-exact ([0-9.e+-]+) ref-exact ([0-9.e+-]+)", ln)
        if not m:
            continue
        k = re.search(r" n=(\d+) m=(\d+)", ln)
        if k:
            n, mv = int(k.group(1)), int(k.group(2))
        else:
            k = re.search(r" (\d+)x(\d+) m=(\d+)", ln)
            n, mv = int(k.group(1)) * int(k.group(2)), int(k.group(3))
        yield float(m.group(1)), float(m.group(2)), n, mv, ln.startswith("stop")


def main():
    out, files = sys.argv[1], sys.argv[2:]
    lines = ["# tools/fuzz_gpu.py on the final tree of round 4, judged by the truth rule (tests/parity_util.py): per sequence,",
             f"#   max err_dev <= max(1e-12, F x max err_ref)  against the extended-precision trajectory, F = {P.TRUTH_FACTOR:g} beyond one tile "
             f"(n > {P.TINY_N}), {P.TRUTH_FACTOR_TINY:g} within;",
             "# decisions exact after every call in every sequence listed; sequences beyond the allowance are recorded, not fatal.",
             "# (Re-evaluated from the per-sequence records of the runs by tools/soak_summary.py; the sharded runs count one record per rank.)", ""]
    allr = []
    for f in files:
        rec = list(records(f))
        failed = sum(1 for ln in open(f) if ln.startswith("FAIL"))
        used, beyond = [], []
        for dev, ref, n, mv, _ in rec:
            fac = P.truth_factor(n, mv)
            used.append(dev / (fac * ref) if dev > 1e-12 and ref > 0 else 0.0)
            if dev > max(1e-12, fac * ref):
                beyond.append(f"n={n} mvec={mv} err_dev={dev:.2e} err_ref={ref:.2e} ({dev / ref:.1f} x)")
        allr += rec
        used.sort()
        if not used:
            lines.append(f"## {os.path.basename(f)}: no records")
            continue
        tripped = sum(1 for r in rec if r[4])
        lines.append(f"## {os.path.basename(f)}: {len(rec)} sequences" + (f" (+ {failed} ended by the per-call stop or an assertion: see the file)" if failed else "")
                     + (f"; per-call stop tripped in {tripped} (recorded, the sequence ran on and is judged here)" if tripped else ""))
        lines.append(f"share of the allowance used: median {used[len(used) // 2]:.2f}, 90 % {used[int(.9 * len(used))]:.2f}, largest {used[-1]:.2f}; "
                     f"beyond the allowance: {len(beyond)} {beyond}")
        lines.append("")
    lines.append(f"## err_dev / err_ref by vector length, all {len(allr)} records (only sequences that end with err_dev > 1e-12 have a ratio)")
    lines.append(f"{'n':>14s} {'sequences':>10s} {'with ratio':>11s} {'median':>7s} {'90 %':>6s} {'max':>6s} {'> 2':>5s} {'> 4':>5s}")
    for lo, hi in ((1, 16), (17, 128), (129, 512), (513, 2048), (2049, 10**9)):
        sel = [r for r in allr if lo <= r[2] <= hi]
        rat = sorted(r[0] / r[1] for r in sel if r[0] > 1e-12 and r[1] > 0)
        lab = f"{lo}..{hi}" if hi < 10**9 else f"> {lo - 1}"
        if rat:
            lines.append(f"{lab:>14s} {len(sel):10d} {len(rat):11d} {rat[len(rat) // 2]:7.2f} {rat[int(.9 * len(rat))]:6.2f} {rat[-1]:6.2f} "
                         f"{sum(1 for x in rat if x > 2):5d} {sum(1 for x in rat if x > 4):5d}")
        else:
            lines.append(f"{lab:>14s} {len(sel):10d} {0:11d}")
    lines.append("")
    lines.append("## both directions: records in which either err_dev or err_ref ends above 1e-12 -- how often is the DEVICE further from the")
    lines.append("## extended-precision trajectory than the reference by more than 2 x / 4 x, and how often the REFERENCE further than the device")
    lines.append(f"{'n':>14s} {'records':>8s} {'device > 2 x':>13s} {'device > 4 x':>13s} {'reference > 2 x':>16s} {'reference > 4 x':>16s}")
    for lo, hi in ((1, 16), (17, 512), (513, 2048), (2049, 10**9)):
        sel = [r for r in allr if lo <= r[2] <= hi and max(r[0], r[1]) > 1e-12 and min(r[0], r[1]) > 0]
        lab = f"{lo}..{hi}" if hi < 10**9 else f"> {lo - 1}"
        lines.append(f"{lab:>14s} {len(sel):8d} {sum(1 for r in sel if r[0] > 2 * r[1]):13d} {sum(1 for r in sel if r[0] > 4 * r[1]):13d} "
                     f"{sum(1 for r in sel if r[1] > 2 * r[0]):16d} {sum(1 for r in sel if r[1] > 4 * r[0]):16d}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
