mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py -m gpu -q --tb=short -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
NKA_HIP_LIB=$PWD/nka_amd/libnka_hip_stamps.so python tools/solve_phases.py --mvec 20 --variant 0
