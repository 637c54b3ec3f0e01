# round-4: reference-order sums -- the whole GPU suite (the default at n <= 64 changed), the cost table, a soak over the three sum modes
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/pytest_gpu.log 2>&1
rc=$?; grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" gpurun_out/pytest_gpu.log | grep -E "passed|failed|Error|error|assert|FAILED" | tail -15
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/sum_order_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sum_order_cost.txt
[ ${PIPESTATUS[0]} -ge 124 ] && exit 124
timeout -k 10 300 python tools/fuzz_gpu.py --seconds 200 --first-seed 9000 --out gpurun_out/fuzz_array_sums.txt > gpurun_out/fuzz_array_sums.log 2>&1
rc=$?; tail -1 gpurun_out/fuzz_array_sums.log | cut -c1-700; grep -A12 "^FAIL" gpurun_out/fuzz_array_sums.txt | head -40
exit $rc
