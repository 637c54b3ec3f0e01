#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace + stats of the abstract-vector path
# (BASELINE configs[4]: 4 fields x 1e7, mvec 20) through the Fortran driver.
# Usage: tools/rocprof_vector.sh <tag> [compact 0|1]
set -u
TAG=${1:-r02}; CP=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_vector$CP
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/profiles_$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$ROOT/nka_amd/fortran/build/nka_vector_driver" bench 4 10000000 20 20 $CP > "$OUT/bench.log" 2>&1
echo "trace rc=$?"; cat "$OUT/bench.log"
S=$(ls "$OUT"/trace/*/*_kernel_stats.csv | head -1)
{ head -1 "$S"; grep -E "k_[a-z_0-9]+<|nka::" "$S"; } > "$ROOT/gpurun_out/profiles_$TAG/kernel_stats_vector_compact$CP.csv"
cp "$OUT/bench.log" "$ROOT/gpurun_out/profiles_$TAG/vector_bench_under_trace_compact$CP.log"
cat "$ROOT/gpurun_out/profiles_$TAG/kernel_stats_vector_compact$CP.csv"
# HBM traffic by the PMC counters: two more runs, one counter each (never combined with other trace domains), summarised per kernel
# family against the byte model of bench.py's config5_abstract_vector (tools/pmc_vector_summary.py)
if [ "${NKA_VECTOR_PMC:-1}" = "1" ]; then
  for pass in "FETCH_SIZE pmc_fetch" "WRITE_SIZE pmc_write"; do
    set -- $pass
    rocprofv3 --kernel-trace --pmc $1 --output-format csv -d "$OUT/$2" -- "$ROOT/nka_amd/fortran/build/nka_vector_driver" bench 4 10000000 20 6 $CP > "$OUT/$2.log" 2>&1
    echo "$1 rc=$?"
  done
  python3 "$ROOT/tools/pmc_vector_summary.py" "$OUT" "$ROOT/gpurun_out/profiles_$TAG/pmc_traffic_vector_compact$CP.json" --n 40000000 --mvec 20 --compact $CP
fi
# timeline of two steady-state updates: kernel starts/ends relative to the first
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:60]))
rows.sort()
# last 2 updates = last 2 occurrences of k_update_norm2
idx = [i for i, r in enumerate(rows) if "k_update_norm2" in r[2]]
if len(idx) >= 3:
    a = idx[-3]
    t0 = rows[a][0]
    prev = None
    for s, e, k in rows[a:idx[-1]]:
        print(f"{(s - t0) / 1e3:10.1f} us  gap {((s - prev) / 1e3 if prev else 0):6.1f}  +{(e - s) / 1e3:9.1f} us  {k[:70]}")
        prev = e
PY
