# round-4: scalar-step phases (stamps build) and the small-n sweep after the address block went in
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_round4.py tests/test_hip_parity.py -x -q --tb=short > gpurun_out/pytest_r4.log 2>&1 || { tail -30 gpurun_out/pytest_r4.log; exit 1; }
tail -2 gpurun_out/pytest_r4.log
export NKA_HIP_DIAG_LIB=$PWD/nka_amd/libnka_hip_stamps.so
for nm in "1e5 20" "1e5 10" "1.25e7 20"; do set -- $nm; echo "## n = $1, mvec = $2"; timeout -k 10 200 python tools/solve_phases.py --vlen $1 --mvec $2 2>&1 | grep -v amdgpu; done
unset NKA_HIP_DIAG_LIB
echo "n mvec updates/s us/update frac frac PA solve PB"
for m in 10 20; do for n in 1e4 1e5 1e6; do for rep in 1 2; do
  NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --vlen $n --mvec $m --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(d['value'],1), round(1e3*d['ms_per_step'],1), round(r['frac'],3), round(r['whole_update']['frac'],3), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))"
done; done; done
