mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_example_dev_gpu.py -m gpu -q --tb=short 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
./tools/hbm_probe 1e8 0 w 2>&1 | tee gpurun_out/hbm_probe_write.txt | sort -k7 -n -r | head -12
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'roofline.frac',d['roofline']['frac'],'whole',d['roofline']['whole_update']['frac'])
print('f08',d['also_f08_rounding']['value'],d['also_f08_rounding']['roofline']['frac'],d['also_f08_rounding']['roofline']['whole_update']['frac'])
print(json.dumps(d.get('config5_abstract_vector'),indent=1))
print(d['cpu_baseline'])"
