mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"
grep -n "passed\|failed\|error" gpurun_out/pytest_gpu.log | tail -5
