#!/bin/bash
# Round 5, VERDICT item 2: Infinity-Cache reuse of f and the pending pair between PA and PB at shard sizes.
# (a) temporal loads for those vectors (libnka_hip_diag_ft.so), (b) PB in the reverse of PA's tile order, (c) both.
set -e
OUT=gpurun_out/r5_mall_ab.txt
: > $OUT
L0=nka_amd/libnka_hip_diag.so
L1=nka_amd/libnka_hip_diag_ft.so
for spec in "c 1.25e7 20" "f08 1.25e7 20" "c 1e7 10" "f08 1e7 10" "c 5e6 20" "c 1e8 20"; do
  set -- $spec
  echo "=== flavor $1 n $2 m $3" | tee -a $OUT
  python tools/ab_libs.py --libs $L0 $L1 --combos pb_reverse=0 pb_reverse=1 --flavor $1 --vlen $2 --mvec $3 \
      --rounds 10 --steps 16 --check-bits 2>&1 | tee -a $OUT
done
