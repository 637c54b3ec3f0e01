timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py tests/test_example_dev_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -6
for m in 5 10; do for n in 1e7 1e8; do
python tools/ab_inproc.py --flavor c --vlen $n --mvec $m --key pb_pipe --values 0 -1 --rounds 6 --steps 10 | tail -2
python tools/ab_inproc.py --flavor c --vlen $n --mvec $m --key pa_pipe --values 0 -1 --rounds 6 --steps 10 | tail -2
done; done
