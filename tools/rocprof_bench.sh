#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of bench.py,
# then -- in SEPARATE passes with --kernel-trace only, as the MI355X guide
# prescribes -- the HBM byte counters FETCH_SIZE and WRITE_SIZE.
# Usage: tools/rocprof_bench.sh <tag> <flavor> [bench args...]
set -u
TAG=${1:-r01}; FL=${2:-c}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_$FL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --flavor $FL "$@" > "$OUT/bench_under_trace.log" 2>&1
echo "trace rc=$?"; tail -1 "$OUT/bench_under_trace.log" | cut -c1-300
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --flavor $FL --steps 4 "$@" > "$OUT/bench_under_pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --flavor $FL --steps 4 "$@" > "$OUT/bench_under_pmc_write.log" 2>&1
echo "pmc write rc=$?"
mkdir -p "$ROOT/gpurun_out/profiles_$TAG"
# (the vector length / subspace size of the run, for the words-per-element columns: --vlen / --mvec among the bench arguments)
N=1e8; M=20; prev=""
for arg in "$@"; do
  [ "$prev" = "--vlen" ] && N=$arg
  [ "$prev" = "--mvec" ] && M=$arg
  prev=$arg
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$ROOT/gpurun_out/profiles_$TAG" --flavor $FL --n $N --mvec $M | tail -30
cp "$OUT/bench_under_trace.log" "$ROOT/gpurun_out/profiles_$TAG/bench_under_trace_$FL.log"
