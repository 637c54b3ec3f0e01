# round-4 evidence, part C: soak runs under the truth rule, the plain launch form of bench.py, the one-GPU rehearsal
# of every shard size, the size sweep
mkdir -p gpurun_out
timeout -k 10 330 python tools/fuzz_gpu.py --seconds 240 --out gpurun_out/fuzz_array.txt | tail -2
timeout -k 10 260 python tools/fuzz_gpu.py --seconds 150 --vector --out gpurun_out/fuzz_vector.txt | tail -2
timeout -k 10 160 python tools/fuzz_gpu.py --seconds 80 --hostdot --out gpurun_out/fuzz_hostdot.txt | tail -2
timeout -k 10 260 python tools/fuzz_gpu.py --seconds 120 --sharded 3 --out gpurun_out/fuzz_sharded.txt | tail -2
echo "== plain form, two ranks sharing the GPU (rehearsal: gloo + staged hook)"
NKA_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --backend gloo --allreduce staged --vlen 3000001 --mvec 6 --steps 6 --no-cpu-baseline > gpurun_out/plain_form_rehearsal.json 2> gpurun_out/plain_form_rehearsal.err; echo "rc=$?"; cut -c1-400 gpurun_out/plain_form_rehearsal.json
echo "== plain form, two ranks on ONE GPU with the default --allreduce rccl: RCCL cannot work, the line must still come"
NKA_BENCH_SHARE_GPU=1 NKA_BENCH_WATCHDOG_S=90 timeout -k 10 400 python bench.py --gpus 2 --vlen 2000001 --mvec 5 --steps 5 --no-cpu-baseline --launch-timeout 150 > gpurun_out/plain_form_fallback.json 2> gpurun_out/plain_form_fallback.err; echo "rc=$?"; cut -c1-300 gpurun_out/plain_form_fallback.json; grep -E "nka_amd.dist|\[bench\]" gpurun_out/plain_form_fallback.err | head -12
echo "== one-GPU rehearsal of every shard size"
bash tools/scale_rehearsal.sh > /dev/null 2>&1; cat gpurun_out/scale_rehearsal.txt
echo "== sweep"
bash tools/sweep.sh c > gpurun_out/sweep_n_mvec.txt 2>&1; cat gpurun_out/sweep_n_mvec.txt
