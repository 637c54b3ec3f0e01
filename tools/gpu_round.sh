mkdir -p gpurun_out
for p in 101 102; do NKA_HIP_PA_PIPE=$p NKA_HIP_PB_PIPE=$p timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -3; done
python tools/ab_inproc.py --flavor c --key pa_pipe --values 0 4 102 101 --rounds 6 --steps 10
python tools/ab_inproc.py --flavor c --key pb_pipe --values 0 4 102 101 --rounds 6 --steps 10
python tools/ab_inproc.py --flavor f08 --key pb_pipe --values 0 4 102 101 --rounds 6 --steps 10
