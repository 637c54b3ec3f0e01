#!/bin/bash
# Run on the GPU box (through gpurun): memory-system counters of the two streaming
# kernels, to explain where k_combine (reads + 5 store streams) loses against the
# pure-read k_dots.  One rocprofv3 --pmc pass per counter group (the TCC block
# has 4 counter slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), each
# with --kernel-trace only, as the guide prescribes.
# Usage: tools/pmc_memsys.sh <tag> <flavor> [bench args...]
set -u
TAG=${1:-r02}; FL=${2:-f08}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/memsys_${TAG}_$FL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export NKA_BENCH_SECONDARY=0
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" \
      --no-cpu-baseline --flavor $FL --steps 4 ${EXTRA[@]+"${EXTRA[@]}"} > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
EXTRA=("$@")
pass ea_req      TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass ea_credit   TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum
pass ea_level    TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_REQ_sum TCC_IB_STALL_sum
pass tcp_lat     TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
pass sq_wait     SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL
pass grbm        GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum
pass utcl1       TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
mkdir -p "$ROOT/gpurun_out/profiles_$TAG"
python3 "$ROOT/tools/pmc_memsys_summary.py" "$OUT" > "$ROOT/gpurun_out/profiles_$TAG/memsys_counters_$FL.txt"
cat "$ROOT/gpurun_out/profiles_$TAG/memsys_counters_$FL.txt"
