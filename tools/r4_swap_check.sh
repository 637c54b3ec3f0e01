# round-4: tests of the out-of-place entry + tools/ab_swap.py (profiles/r04/ab_swap_entry.txt)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_round4.py -x -q --tb=short > gpurun_out/pytest_r4.log 2>&1 || { tail -40 gpurun_out/pytest_r4.log; exit 1; }
tail -3 gpurun_out/pytest_r4.log
for fl in c f08; do
timeout -k 10 300 python tools/ab_swap.py --flavor $fl --vlen 1e8 --mvec 20 --rounds 4 --steps 8 > gpurun_out/ab_swap_${fl}_1e8.txt 2>&1 || { tail -20 gpurun_out/ab_swap_${fl}_1e8.txt; exit 1; }
tail -4 gpurun_out/ab_swap_${fl}_1e8.txt
done
timeout -k 10 300 python tools/ab_swap.py --flavor c --vlen 1.25e7 --mvec 20 --rounds 6 --steps 10 > gpurun_out/ab_swap_c_1.25e7.txt 2>&1; tail -4 gpurun_out/ab_swap_c_1.25e7.txt
timeout -k 10 300 python tools/ab_swap.py --flavor c --vlen 1e7 --mvec 10 --rounds 6 --steps 10 > gpurun_out/ab_swap_c_1e7_m10.txt 2>&1; tail -4 gpurun_out/ab_swap_c_1e7_m10.txt
timeout -k 10 300 python tools/ab_swap.py --flavor c --vlen 1e5 --mvec 20 --rounds 6 --steps 20 > gpurun_out/ab_swap_c_1e5.txt 2>&1; tail -4 gpurun_out/ab_swap_c_1e5.txt
