#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of bench.py,
# then (separate passes, as the MI355X guide prescribes) the HBM byte counters.
# Usage: tools/rocprof_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_under_trace.log" 2>&1
echo "trace rc=$?"
if [ "${PMC:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 "$@" > "$OUT/bench_under_pmc_fetch.log" 2>&1
  echo "pmc fetch rc=$?"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 "$@" > "$OUT/bench_under_pmc_write.log" 2>&1
  echo "pmc write rc=$?"
fi
find "$OUT" -name "*.csv" | head -50
