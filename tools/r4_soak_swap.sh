# round-4: soak runs with out-of-place updates mixed in (odd seeds), array and sharded (profiles/r04/fuzz_soak.txt)
mkdir -p gpurun_out
timeout -k 10 500 python tools/fuzz_gpu.py --seconds 420 --first-seed 3000 --out gpurun_out/fuzz_array_swap.txt | tail -1
timeout -k 10 300 python tools/fuzz_gpu.py --seconds 180 --first-seed 3000 --sharded 3 --out gpurun_out/fuzz_sharded_swap.txt | tail -2
