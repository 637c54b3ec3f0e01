# round-4: tests of the list word + its in-process A/B (profiles/r04/ab_list_word.txt); needs libnka_hip_diag.so (built by build())
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_round4.py -x -q --tb=short > gpurun_out/pytest_r4.log 2>&1 || { tail -40 gpurun_out/pytest_r4.log; exit 1; }
tail -3 gpurun_out/pytest_r4.log
for fl in c f08; do
timeout -k 10 300 python tools/ab_inproc.py --key list_word --values 0 1 --flavor $fl --vlen 1e8 --mvec 20 --span-dim 12 --rounds 4 --steps 8 > gpurun_out/ab_list_word_${fl}_1e8.txt 2>&1; cat gpurun_out/ab_list_word_${fl}_1e8.txt | tail -4
done
timeout -k 10 300 python tools/ab_inproc.py --key list_word --values 0 1 --flavor c --vlen 1e8 --mvec 20 --span-dim 5 --rounds 4 --steps 8 > gpurun_out/ab_list_word_c_1e8_d5.txt 2>&1; tail -4 gpurun_out/ab_list_word_c_1e8_d5.txt
timeout -k 10 300 python tools/ab_inproc.py --key list_word --values 0 1 --flavor c --vlen 1.25e7 --mvec 20 --span-dim 12 --rounds 6 --steps 10 > gpurun_out/ab_list_word_c_1.25e7.txt 2>&1; tail -4 gpurun_out/ab_list_word_c_1.25e7.txt
NKA_BENCH_SECONDARY=0 timeout -k 10 300 python bench.py --workload drops --no-cpu-baseline > gpurun_out/bench_drops.log 2>&1; tail -1 gpurun_out/bench_drops.log | cut -c1-1500
