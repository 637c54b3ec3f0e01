# round-4 final pass.  (1) The seeds the soak with out-of-place updates flagged, in place against out of place bit for bit
# (array: 3067 3069 3219; sharded over 3 ranks: 3001 3319).  (2) The sharded soak with out-of-place updates continued
# behind seed 3319 (stopped there by the per-call stop: profiles/r04/sharded_seed_3319_replay.txt).  (3) The whole GPU
# suite, smoke, default bench (tools/gpu_check.sh).
mkdir -p gpurun_out
timeout -k 10 200 python tools/swap_vs_inplace_seed.py 3067 3069 3219 > gpurun_out/swap_vs_inplace_seed.txt 2>&1 &&
timeout -k 10 200 python tools/swap_vs_inplace_seed.py --sharded 3 3001 3319 2>&1 | grep -E "^sharded seed|Error|assert" >> gpurun_out/swap_vs_inplace_seed.txt
rc=$?
cat gpurun_out/swap_vs_inplace_seed.txt | cut -c1-250
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 240 --first-seed 3320 --sharded 3 --out gpurun_out/fuzz_sharded_swap2.txt > gpurun_out/fuzz_sharded_swap2.log 2>&1
echo "sharded soak from 3320: rc $?"; grep -c "^ok" gpurun_out/fuzz_sharded_swap2.txt.rank0; grep "^FAIL" gpurun_out/fuzz_sharded_swap2.txt.rank*
bash tools/gpu_check.sh
