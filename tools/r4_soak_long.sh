# round-4: second soak run with fresh seeds (profiles/r04/fuzz_soak.txt)
mkdir -p gpurun_out
timeout -k 10 700 python tools/fuzz_gpu.py --seconds 600 --first-seed 1000 --out gpurun_out/fuzz_array_long.txt | tail -1
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 300 --first-seed 1000 --vector --out gpurun_out/fuzz_vector_long.txt | tail -1
