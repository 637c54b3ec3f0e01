#!/bin/bash
# Run on the GPU box: where does a SMALL update spend its time?  rocprofv3 kernel trace of
# bench.py at the given vector lengths; prints the kernel timeline of two steady-state
# updates (start, gap to the previous kernel's end, duration) and the mean gap / duration
# per kernel over the timed steps.
# Usage: tools/small_n_trace.sh <tag> [flavor] [n ...]
set -u
TAG=${1:-r02}; FL=${2:-c}; shift 2 || true
NS=${@:-1e5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out/profiles_$TAG"
cd /tmp && export TMPDIR=/tmp
for N in $NS; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_small_$N
  rm -rf "$OUT"; mkdir -p "$OUT"
  NKA_BENCH_SECONDARY=0 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --flavor $FL --vlen $N --mvec 20 --steps 50 > "$OUT/bench.log" 2>&1
  echo "n=$N trace rc=$?"
  python3 - "$OUT" "$N" <<'PY' | tee "$ROOT/gpurun_out/profiles_$TAG/small_n_timeline_$N.txt"
import csv, glob, json, re, sys
out, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads([ln for ln in open(out + "/bench.log").read().splitlines() if ln.startswith("{")][-1])
    k = d["roofline"]["kernels"]
    print(f"# n={n}: bench under trace: {d['value']:.0f} updates/s, {1e3*d['ms_per_step']:.1f} us/update (wall); HIP-event phases "
          f"PA {1e3*k['PA_k_dots']['mean_ms']:.1f} solve {1e3*k['k_solve']['mean_ms']:.1f} PB {1e3*k['PB_k_combine']['mean_ms']:.1f} us")
except Exception as exc:
    print("# no bench line:", exc)
rows = []
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:50]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_combine")]
if len(idx) >= 12:
    a, b = idx[-12] + 1, idx[-10] + 1          # two updates well inside the timed steps
    t0 = rows[a][0]
    prev = rows[a - 1][1]
    for s, e, k in rows[a:b]:
        print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  +{(e - s) / 1e3:8.1f} us  {k[:60]}")
        prev = e
    # means over the last 40 updates
    a = idx[-41] + 1
    agg = {}
    prev = rows[a - 1][1]
    for s, e, k in rows[a:idx[-1] + 1]:
        g = agg.setdefault(k.split("<")[0], [0, 0.0, 0.0])
        g[0] += 1; g[1] += (s - prev) / 1e3; g[2] += (e - s) / 1e3
        prev = e
    tot = (rows[idx[-1]][1] - rows[idx[-41]][1]) / 1e3 / 40
    print(f"# means over 40 updates: {tot:.1f} us per update (device timeline)")
    for k, (c, g, dur) in agg.items():
        print(f"#   {k:28s} x{c / 40:.0f}  gap before {g / c:6.1f} us   duration {dur / c:7.1f} us")
PY
done
