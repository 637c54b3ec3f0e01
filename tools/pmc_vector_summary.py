#!/usr/bin/env python3
"""HBM traffic of the abstract-vector kernels (config 5) from two rocprofv3 PMC passes over nka_vector_driver.

  tools/pmc_vector_summary.py <prof_dir> <out.json> --n 40000000 --mvec 20 --compact 0|1

<prof_dir> holds pmc_fetch/ and pmc_write/ (tools/rocprof_vector.sh).  Counter handling as tools/pmc_summary.py and
/opt/skills/guides/MI355X_MICROARCH.md prescribe: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, KiB, FETCH_SIZE x 2 on
gfx950 for wide coalesced streaming reads.  Per kernel family the steady-state instance (the widest template instance,
launched most often) is averaged over the upper half of its launches and set against the byte MODEL bench.py's
config5_abstract_vector uses (until round 6 that fraction rested on the model alone):
   pure-read stage   k_scale_dot_pair_many_win<L, false, true, true, W>   2 + m words per element (x, z and the m stored w)
   combine stage     k_update_many_keep_win<L, true, W>    6 + 2m  (reference rounding: f, 2m stored vectors, the raw pair; 5 stores)
                     k_update_many_keep_win<L, false, W>   7 + m   (compact storage)
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def load(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                mt = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])      # "void (anonymous namespace)::k_x<20, true, 2>(long, ...)"
                if mt:
                    agg[mt.group(1)].append(float(r["Counter_Value"]))
    return agg


def steady(agg, stem):
    best, key = None, (-1, -1)
    for k in agg:
        m = re.match(rf"{stem}\w*<(\d+)", k)
        if m and (int(m.group(1)), len(agg[k])) > key:
            best, key = k, (int(m.group(1)), len(agg[k]))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prof_dir")
    ap.add_argument("out")
    ap.add_argument("--n", type=float, default=4e7)
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--compact", type=int, default=0)
    a = ap.parse_args()
    fetch = load(os.path.join(a.prof_dir, "pmc_fetch"), "FETCH_SIZE")
    write = load(os.path.join(a.prof_dir, "pmc_write"), "WRITE_SIZE")
    n, m = a.n, a.mvec
    # the kernels config 5 runs in steady state (fused norm stage): k_scale_dot_pair_many_win<L, false, true, true, W> is the
    # PURE-READ form of the scale-and-dot stage (d and w1' formed in registers only: x, z and the m stored w = 2 + m words);
    # k_update_many_keep_win<L, true, W> combines with two stored vectors per pair (6 + 2m), <L, false, W> is the compact
    # (axpy) form with one (7 + m)
    combine = (6 + 2 * m) if a.compact == 0 else (7 + m)
    model = {"k_diff_norm_dot_pair_many": 2 + m, "k_update_norm2_dots": 2 + m, "k_scale_dot_pair_many": 2 + m, "k_dot_pair_many_scaled": 2 + m,
             "k_update_many_keep": combine, "k_axpy_many_keep": combine}
    out = {"workload": f"nka_vector_driver bench 4 x {int(n) // 4}, mvec {m}, compact {a.compact}", "n": int(n), "mvec": m,
           "unit": "bytes per launch", "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 halves wide coalesced reads); WRITE_SIZE [KiB] x 1024",
           "kernels": {}, "all_kernels_seen": sorted(set(fetch) | set(write))[:80]}
    total = model_total = 0.0
    for stem, words in model.items():
        kf, kw = steady(fetch, stem), steady(write, stem)
        if not kf or not kw:
            continue
        fv = sorted(fetch[kf])[-max(1, len(fetch[kf]) // 2):]
        wv = sorted(write[kw])[-max(1, len(write[kw]) // 2):]
        rb, wb = sum(fv) / len(fv) * 1024 * 2, sum(wv) / len(wv) * 1024
        if (rb + wb) < 0.2 * 8 * n:            # (a family with no steady-state launches of config-5 size)
            continue
        out["kernels"][stem] = {"kernel": kf, "launches_averaged": len(fv), "read_bytes": rb, "write_bytes": wb,
                                "words_per_element": (rb + wb) / (8 * n), "model_words_per_element": words,
                                "traffic_over_model": (rb + wb) / (8 * n * words)}
        total += rb + wb
        model_total += 8 * n * words
    out["hbm_bytes_per_update"] = total
    out["model_bytes_per_update"] = model_total
    out["traffic_over_model"] = total / model_total if model_total else None
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "all_kernels_seen"}, indent=1))


if __name__ == "__main__":
    main()
