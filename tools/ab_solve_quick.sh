#!/bin/bash
# solve-kernel check after a change: phases by s_memtime stamps (diagnostic build) and the small-n lines of the sweep
mkdir -p gpurun_out
for m in 10 20; do NKA_HIP_LIB=$PWD/nka_amd/libnka_hip_stamps.so python tools/solve_phases.py --mvec $m; done > gpurun_out/solve_phases.txt 2>&1 || exit 1
echo "n mvec updates/s us/update PA_us solve_us PB_us" > gpurun_out/sweep_small.txt
for m in 5 10 20; do for n in 1e4 1e5 1e6; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --flavor c --vlen $n --mvec $m --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(d['value'],1), round(1e3*d['ms_per_step'],1), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))" >> gpurun_out/sweep_small.txt || exit 1
done; done
cat gpurun_out/solve_phases.txt | grep -v "^$" | tail -24; cat gpurun_out/sweep_small.txt
