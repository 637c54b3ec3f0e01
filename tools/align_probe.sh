#!/bin/bash
# does a slot pitch that is a multiple of the 4 KiB tile matter?  n = k*512 against its ragged neighbours
echo "n mvec us/update PA_us solve_us PB_us"
for n in 9999872 10000000 10000384 12499968 12500000 12500480 99999744 100000000 100000256; do for m in 20; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --flavor c --vlen $n --mvec $m --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(1e3*d['ms_per_step'],1), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))"
done; done
