mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_gpu.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
