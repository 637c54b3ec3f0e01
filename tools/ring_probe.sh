#!/bin/bash
# small-n latency of the rolling-window kernels against the ring size: prime widths run with every load in flight
echo "n mvec us/update PA_us solve_us PB_us"
for n in 1e4 1e5 1e6; do for m in 16 17 19 20 23 24; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --flavor c --vlen $n --mvec $m --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(1e3*d['ms_per_step'],1), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))"
done; done
