mkdir -p gpurun_out
for p in 2 4; do NKA_HIP_PA_PIPE=$p timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -3; done
python tools/ab_inproc.py --flavor c --key pa_pipe --values 0 2 4 --rounds 8 --steps 10
python tools/ab_inproc.py --flavor c --vlen 1.25e7 --key pa_pipe --values 0 2 4 --rounds 8 --steps 20
