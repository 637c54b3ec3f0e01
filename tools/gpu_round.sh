mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_gpu.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'roofline.frac',d['roofline']['frac'],'whole',d['roofline']['whole_update']['frac'], 'PA', d['roofline']['kernels']['PA_k_dots']['mean_ms'], 'PB', d['roofline']['kernels']['PB_k_combine']['mean_ms'])
a=d['also_f08_rounding']; print('f08',a['value'],a['roofline']['frac'],a['roofline']['whole_update']['frac'])
print({k:(v['value'],v['frac']) for k,v in d['config5_abstract_vector'].items() if isinstance(v,dict)})"
bash tools/sweep.sh > gpurun_out/sweep_n_mvec.txt 2>&1; cat gpurun_out/sweep_n_mvec.txt
