#!/bin/bash
# updates/s and roofline fraction over vector length and subspace size (one GPU)
echo "n mvec updates/s ms/update frac(dominant kernel) frac(whole update) PA_ms solve_ms PB_ms"
for m in 5 10 20; do for n in 1e4 1e5 1e6 1e7 1e8; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --vlen $n --mvec $m --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(d['value'],1), round(d['ms_per_step'],4), round(r['frac'],3), round(r["whole_update"]["frac"],3), round(k['PA_k_dots']['mean_ms'],4), round(k['k_solve']['mean_ms'],4), round(k['PB_k_combine']['mean_ms'],4))"
done; done
