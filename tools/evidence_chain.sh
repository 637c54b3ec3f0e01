#!/bin/bash
# Evidence of the reference-order sums of long vectors (k_chain_sums): on the GPU box, through gpurun.
#   make -C nka_amd/csrc chain_stamps   (here, before the call: the stamps build travels with the snapshot)
#   gpurun -- 'bash tools/evidence_chain.sh'        -> gpurun_out/reference_order_chain.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/reference_order_chain.txt
cd "$ROOT"
{
  echo "# 1. one VALU instruction of one wavefront (tools/micro/dep_add.hip)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/dep_add.hip -o /tmp/dep_add 2>/dev/null && timeout -k 10 120 /tmp/dep_add | grep "blocks=   1 "
  echo
  echo "# 2. one sum, device time of k_chain_sums (tools/chain_sum_time.py; stamps build: phases of wavefront 0 and block counts)"
  NKA_HIP_DIAG_LIB=$ROOT/nka_amd/libnka_hip_chain_stamps.so timeout -k 10 300 python tools/chain_sum_time.py 1e6 1e7 1e8 2>&1 | grep -v amdgpu.ids
  echo
  echo "# 3. whole updates, wall clock (tools/sum_order_cost.py): blocked passes against reference-order sums"
  timeout -k 10 300 python tools/sum_order_cost.py 2>&1 | grep -v amdgpu.ids
  timeout -k 10 300 python tools/sum_order_cost.py 1e7 20 8 2>&1 | grep -v amdgpu.ids | tail -1
  timeout -k 10 300 python tools/sum_order_cost.py 1e8 20 4 2>&1 | grep -v amdgpu.ids | tail -1
  echo
  echo "# 4. the same with CORRELATED inputs (every input in the span of 12 fixed vectors: the inner products are sums that drift"
  echo "#    instead of zero-mean random walks, so hardly a block meets an end of its binade; a dependence drop per update)"
  timeout -k 10 300 python tools/sum_order_cost.py 1e7 20 8 12 2>&1 | grep -v amdgpu.ids | tail -2
  timeout -k 10 300 python tools/sum_order_cost.py 1e8 20 4 12 2>&1 | grep -v amdgpu.ids | tail -1
} > "$OUT" 2>&1
tail -50 "$OUT"
