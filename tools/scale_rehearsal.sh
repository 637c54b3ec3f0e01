#!/bin/bash
# Rehearsal of the driver's multi-GPU runs on ONE GPU (no 8-GPU node is available to the builder):
# for N = 1, 2, 4, 8 the exact driver form
#     python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 ... bench.py --gpus 1 --vlen 1e8/N
# with NKA_BENCH_FORCE_HOOK=1 (the library's RCCL communicator with one rank, its all-reduce on the kernel
# stream between PA and the scalar step: the whole N > 1 code path except the xGMI hops), then the measured
# single-rank RCCL latency (tools/rccl_latency.py).  Prints the per-shard update time and the PROJECTED strong
# scaling  t(1) / t_shard(N)  -- a projection from one GPU, NOT a measured scaling curve.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/scale_rehearsal.txt}
HOOK=${2:-rccl}          # rccl (default) | p2p: the hook forced on the one rank (round 5: the peer-to-peer exchange with its own mailbox)
SUMS=${3:-default}       # default (round 6: the norm first, two exchanges per update) | blocked (the single-pass fast mode, one exchange)
cd "$ROOT"
: > "$OUT"
echo "# one MI355X running the shard of an N-GPU job (n_global = 1e8, mvec = 20, default flavour, sums $SUMS); hook '$HOOK' forced, one rank" | tee -a "$OUT"
PORT=29600
for N in 1 2 4 8; do
  NL=$((100000000 / N))
  PORT=$((PORT + 1))
  NKA_BENCH_FORCE_HOOK=1 NKA_BENCH_SECONDARY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
      --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --vlen $NL --steps 40 --no-cpu-baseline --allreduce $HOOK --sums $SUMS 2>/dev/null | grep '^{' | tail -1 > /tmp/scale_$N.json
  python3 - "$N" /tmp/scale_$N.json <<'PY' | tee -a "$OUT"
import json, sys
N, path = int(sys.argv[1]), sys.argv[2]
try:
    d = json.load(open(path))
    r = d["ranks"][0]
    print(f"N={N} n_local={d['config']['n_local']:>9d}: {1e3 * d['ms_per_step']:8.1f} us/update  hook={r['hook']} "
          f"comm(nranks,rank)={r['comm_nranks_rank']} replica_check={[c['identical'] for c in d.get('replica_check', [])]} "
          f"whole-update frac {d['roofline']['whole_update']['frac']:.3f} exchange back to back {d.get('exchange', {}).get('us_back_to_back', float('nan')):.1f} us")
except Exception as exc:
    print(f"N={N}: no bench line ({exc})")
PY
done
python3 - "$OUT" <<'PY' | tee -a "$OUT"
import re, sys
t = {}
for ln in open(sys.argv[1]):
    m = re.match(r"N=(\d+) n_local=\s*\d+:\s+([0-9.]+) us/update", ln)
    if m:
        t[int(m.group(1))] = float(m.group(2))
if 1 in t:
    print("# projected strong scaling t(1)/t_shard(N) (one rank: no xGMI hop in the all-reduce):",
          ", ".join(f"N={n}: {t[1] / t[n]:.2f}x" for n in sorted(t)))
PY
[ "$HOOK" = rccl ] && timeout -k 10 300 python tools/rccl_latency.py 12500000 2>/dev/null | grep -v amdgpu.ids | tee -a "$OUT"
exit 0
