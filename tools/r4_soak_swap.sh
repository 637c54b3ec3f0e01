# round-4: soak runs with out-of-place updates mixed in (odd seeds), array and sharded (profiles/r04/fuzz_soak.txt).
# Per-call stops are recorded and the sequence runs on (tests/parity_util.py check(stop=False)); a run killed by its
# timeout starts nothing further.
mkdir -p gpurun_out
timeout -k 10 500 python tools/fuzz_gpu.py --seconds 400 --first-seed 3000 --out gpurun_out/fuzz_array_swap.txt > gpurun_out/fuzz_array_swap.log 2>&1
rc=$?; tail -1 gpurun_out/fuzz_array_swap.log | cut -c1-600
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 300 --first-seed 3000 --sharded 3 --out gpurun_out/fuzz_sharded_swap.txt > gpurun_out/fuzz_sharded_swap.log 2>&1
rc=$?; grep -E "^# seeds|^FAIL" gpurun_out/fuzz_sharded_swap.log; grep -h "^stop" gpurun_out/fuzz_sharded_swap.txt.rank* | cut -c1-400
exit $rc
