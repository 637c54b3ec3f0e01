#!/usr/bin/env python3
"""Table of memory-system counters per launch for the steady-state instances of
k_dots and k_combine, from the passes of tools/pmc_memsys.sh.

  tools/pmc_memsys_summary.py <memsys_dir>

Per counter: mean over the steady-state launches (the widest template instance,
upper half by kernel duration) and the same divided by the launch duration in
shader-clock cycles where that makes a rate (GRBM_GUI_ACTIVE / 8 XCDs ~ cycles)."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    root = sys.argv[1]
    table = collections.defaultdict(dict)          # kernel stem -> counter -> mean
    dur = collections.defaultdict(list)
    for pdir in sorted(glob.glob(os.path.join(root, "*/"))):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(pdir, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not re.search(r"nka::k_(dots|combine)(_win|_pipe)?<", k):
                    continue
                name = k.split("(")[0].replace("void ", "")
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(pdir, "**", "*_kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if re.search(r"nka::k_(dots|combine)(_win|_pipe)?<", k):
                    name = k.split("(")[0].replace("void ", "")
                    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        for stem in ("k_dots", "k_combine"):
            best, key = None, (-1, -1)
            for name in agg:
                m = re.match(rf"nka::{stem}(?:_win|_pipe)?<(\d+)", name)
                nlaunch = max(len(v) for v in agg[name].values())
                if m and (int(m.group(1)), nlaunch) > key:      # widest instance, then the most launched variant
                    best, key = name, (int(m.group(1)), nlaunch)
            if best is None:
                continue
            for ctr, vals in agg[best].items():
                top = sorted(vals)[-max(1, len(vals) // 2):]
                table[best][ctr] = sum(top) / len(top)
    for name in sorted(table):
        d = sorted(dur.get(name, [0.0]))
        top = d[-max(1, len(d) // 2):]
        us = sum(top) / len(top)
        print(f"== {name}   mean steady-state launch {us:.1f} us (under the profiler)")
        for ctr in sorted(table[name]):
            v = table[name][ctr]
            print(f"   {ctr:<42s} {v:18.1f}   per us {v / us if us else 0:14.2f}")
        t = table[name]
        if "TCC_EA0_RDREQ_LEVEL_sum" in t and "TCC_EA0_RDREQ_sum" in table[name]:
            pass
    # derived figures where both operands were collected (different passes of the same workload)
    for name in sorted(table):
        t = table[name]
        print(f"-- derived, {name}")
        if "TCP_TCC_READ_REQ_LATENCY_sum" in t and t.get("TCP_TCC_READ_REQ_sum"):
            print(f"   mean L1->L2 read latency  {t['TCP_TCC_READ_REQ_LATENCY_sum'] / t['TCP_TCC_READ_REQ_sum']:10.1f} cycles")
        if "TCP_TCC_WRITE_REQ_LATENCY_sum" in t and t.get("TCP_TCC_WRITE_REQ_sum"):
            print(f"   mean L1->L2 write latency {t['TCP_TCC_WRITE_REQ_LATENCY_sum'] / t['TCP_TCC_WRITE_REQ_sum']:10.1f} cycles")
        if "SQ_WAIT_ANY" in t and t.get("SQ_WAVE_CYCLES"):
            print(f"   waves parked (s_waitcnt)  {100 * t['SQ_WAIT_ANY'] / t['SQ_WAVE_CYCLES']:10.1f} % of wave cycles")
        if "SQ_ACTIVE_INST_ANY" in t and t.get("SQ_WAVE_CYCLES"):
            print(f"   issuing instructions      {100 * t['SQ_ACTIVE_INST_ANY'] / t['SQ_WAVE_CYCLES']:10.1f} % of wave cycles")


if __name__ == "__main__":
    main()
