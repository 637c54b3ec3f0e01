#!/bin/bash
# One GPU-box round trip: full GPU test suite, then smoke, then the default bench
# (each step only if the one before it succeeded).
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -25 gpurun_out/pytest_gpu.log | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)"
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 &&
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1
rc=$?
tail -1 gpurun_out/bench_default.log
exit $rc
