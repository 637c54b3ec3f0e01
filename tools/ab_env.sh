#!/bin/bash
# A/B environment-selected kernel variants on one box, interleaved, N rounds.
# usage: tools/ab_env.sh ROUNDS "bench args" "VAR=val ..." "VAR=val ..." ...   ("-" = no variables)
R=$1; ARGS=$2; shift 2
for r in $(seq 1 $R); do
  for envs in "$@"; do
    e="$envs"; [ "$e" = "-" ] && e=""
    env $e python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
line='%-40s round $r  %7.2f upd/s  PA %.3f ms (%.0f GB/s)  PB %.3f ms (%.0f GB/s)  solve %.3f' % ('$envs', d['value'], k['PA_k_dots']['mean_ms'], k['PA_k_dots']['achieved'], k['PB_k_combine']['mean_ms'], k['PB_k_combine']['achieved'], k['k_solve']['mean_ms'])
a=d.get('also_f08_rounding')
if a: line += '   | f08 %7.2f upd/s PB %.3f ms' % (a['value'], a['roofline']['kernels']['PB_k_combine']['mean_ms'])
print(line)"
  done
done
