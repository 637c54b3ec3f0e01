#!/bin/bash
# One GPU-box round trip: full GPU test suite + smoke + default bench.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
