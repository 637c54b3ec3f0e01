# round-4: long soak over the three sum modes (array path), then the sharded path once more on the final tree
mkdir -p gpurun_out
timeout -k 10 700 python tools/fuzz_gpu.py --seconds 600 --first-seed 12000 --out gpurun_out/fuzz_array_sums_long.txt > gpurun_out/fuzz_array_sums_long.log 2>&1
rc=$?; tail -1 gpurun_out/fuzz_array_sums_long.log | cut -c1-900
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python tools/fuzz_gpu.py --seconds 200 --first-seed 12000 --sharded 3 --out gpurun_out/fuzz_sharded_final.txt > gpurun_out/fuzz_sharded_final.log 2>&1
rc2=$?; grep -E "^# seeds|^FAIL" gpurun_out/fuzz_sharded_final.log | cut -c1-400
echo "rc array $rc, rc sharded $rc2"
