#!/bin/bash
# Round 5, VERDICT item 6: the peer-to-peer exchange measured as far as one GPU allows.
set -e
OUT=gpurun_out/r5_p2p_latency.txt
: > $OUT
python tools/p2p_latency.py 12500000 2>&1 | grep -v amdgpu.ids | tee -a $OUT
python tools/p2p_latency.py 100000 2>&1 | grep -v amdgpu.ids | tee -a $OUT
for W in 2 4; do
  for N in 4096 1000000; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port 29577 \
        tools/p2p_latency.py share $N 2>&1 | grep -E "ranks sharing|hook " | tee -a $OUT
  done
done
