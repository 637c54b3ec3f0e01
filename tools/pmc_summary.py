#!/usr/bin/env python3
"""Summarise rocprofv3 output of bench.py into profiles/<tag>/.

  tools/pmc_summary.py <prof_dir> <out_dir> [--flavor c] [--n 1e8] [--mvec 20]

<prof_dir> holds trace/, pmc_fetch/, pmc_write/ as written by tools/rocprof_bench.sh.
Writes  kernel_stats.csv (nka kernels only, from --kernel-trace --stats) and
        pmc_traffic.json: HBM bytes per launch of the two streaming kernels.
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes, are in KiB, and on
gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read -> doubled here; WRITE_SIZE is exact for 16-B stores.
Only steady-state launches (the widest template instance of k_dots* / k_combine*,
whichever variant the library chose) are averaged.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def load_counter(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "nka::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


def widest(agg, stem):
    """Steady-state kernel of a pass: the widest template instance; among equal widths (the
    library switches variants while the list fills) the one launched most often."""
    best, key = None, (-1, -1)
    for k in agg:
        m = re.match(rf"nka::{stem}(?:_win|_pipe)?<(\d+)", k)     # k_dots / k_dots_pipe / k_dots_win: the same pass
        if m and (int(m.group(1)), len(agg[k])) > key:
            best, key = k, (int(m.group(1)), len(agg[k]))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prof_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--flavor", default="c")
    ap.add_argument("--n", type=float, default=1e8)
    ap.add_argument("--mvec", type=int, default=20)
    a = ap.parse_args()
    os.makedirs(a.out_dir, exist_ok=True)

    stats = glob.glob(os.path.join(a.prof_dir, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(a.out_dir, f"kernel_stats_{a.flavor}.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(rows[0])
            for r in rows[1:]:
                if "nka::" in r[0] or "rccl" in r[0].lower():
                    w.writerow(r)

    fetch = load_counter(os.path.join(a.prof_dir, "pmc_fetch"), "FETCH_SIZE")
    write = load_counter(os.path.join(a.prof_dir, "pmc_write"), "WRITE_SIZE")
    out = {"flavor": a.flavor, "n": int(a.n), "mvec": a.mvec, "unit": "bytes per launch",
           "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 halves wide coalesced reads); WRITE_SIZE [KiB] x 1024",
           "kernels": {}}
    total = 0.0
    def plain(agg, stem):           # a kernel that is not a template: the norm pass of the default sum mode (k_norm_diff)
        ks = [k for k in agg if re.match(rf"nka::{stem}\b", k)]
        return max(ks, key=lambda k: len(agg[k])) if ks else None
    for stem in ("k_norm_diff", "k_dots", "k_combine"):
        kf, kw = (plain(fetch, stem), plain(write, stem)) if stem == "k_norm_diff" else (widest(fetch, stem), widest(write, stem))
        if not kf or not kw:
            continue
        # steady state = the launches with the most traffic of the widest instance
        fv = sorted(fetch[kf])[-max(1, len(fetch[kf]) // 2):]
        wv = sorted(write[kw])[-max(1, len(write[kw]) // 2):]
        rb = sum(fv) / len(fv) * 1024 * 2
        wb = sum(wv) / len(wv) * 1024
        out["kernels"][stem] = {"kernel": kf, "read_bytes": rb, "write_bytes": wb, "launches_averaged": len(fv),
                                "words_per_element": (rb + wb) / 8.0 / a.n}
        total += rb + wb
    out["hbm_bytes_per_update"] = total
    out["algorithmic_bytes_per_update"] = 8.0 * a.n * (11 + 3 * a.mvec)
    with open(os.path.join(a.out_dir, f"pmc_traffic_{a.flavor}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
