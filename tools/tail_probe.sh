#!/bin/bash
# what does the ragged tail (n mod 512 elements, done by the last block after its tiles) cost at small n?
echo "n mvec us/update PA_us solve_us PB_us"
for n in 99840 100000 999936 1000000 9999872 10000000; do for m in 10 20; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --flavor c --vlen $n --mvec $m --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(1e3*d['ms_per_step'],1), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))"
done; done
