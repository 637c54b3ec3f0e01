#!/bin/bash
# A/B several builds of libnka_hip.so on one box, interleaved, N rounds.
# usage: tools/ab_bench.sh ROUNDS "bench args" lib1 lib2 ...
R=$1; ARGS=$2; shift 2
for r in $(seq 1 $R); do
  for lib in "$@"; do
    NKA_HIP_LIB=$PWD/nka_amd/$lib python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$lib', 'round $r', round(d['value'],2), 'PA', round(k['PA_k_dots']['mean_ms'],3), 'PB', round(k['PB_k_combine']['mean_ms'],3), 'solve', round(k['k_solve']['mean_ms'],3))"
  done
done
