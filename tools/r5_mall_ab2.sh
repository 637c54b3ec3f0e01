#!/bin/bash
# Round 5, item 2, second step: WHICH pass's load policy matters?  ft1 = temporal loads of f / pending w in PA only,
# ft2 = in PB only, ft3 = both (the first run's `ft`).
set -e
OUT=gpurun_out/r5_mall_ab2.txt
: > $OUT
D=nka_amd/libnka_hip_diag
for spec in "c 1.25e7 20" "c 1e7 10" "f08 1.25e7 20" "c 1e8 20"; do
  set -- $spec
  echo "=== flavor $1 n $2 m $3" | tee -a $OUT
  python tools/ab_libs.py --libs $D.so ${D}_ft1.so ${D}_ft2.so ${D}_ft3.so --combos pb_reverse=0 --flavor $1 --vlen $2 --mvec $3 \
      --rounds 10 --steps 16 --check-bits 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
