#!/bin/bash
# updates/s and roofline fractions (bytes moved / time / 8 TB/s) over vector length and subspace size (one GPU)
FL=${1:-c}
echo "# flavor=$FL"
echo "n mvec updates/s us/update frac(dominant_kernel) frac(whole_update) PA_us solve_us PB_us"
for m in 5 10 20; do for n in 1e4 1e5 1e6 1e7 1e8; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --flavor $FL --vlen $n --mvec $m --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(d['value'],1), round(1e3*d['ms_per_step'],1), round(r['frac'],3), round(r['whole_update']['frac'],3), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))"
done; done
