# round-4: the remaining soak modes on the final tree -- user dot product (bit for bit), abstract-vector flavour sharded over two ranks
mkdir -p gpurun_out
timeout -k 10 300 python tools/fuzz_gpu.py --hostdot --seconds 200 --first-seed 30000 --out gpurun_out/fuzz_hostdot_final.txt > gpurun_out/fuzz_hostdot_final.log 2>&1
rc=$?; tail -1 gpurun_out/fuzz_hostdot_final.log | cut -c1-500
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/fuzz_gpu.py --vector-sharded 2 --seconds 240 --first-seed 30000 --out gpurun_out/fuzz_vector_sharded_final.txt > gpurun_out/fuzz_vector_sharded_final.log 2>&1
rc2=$?; tail -1 gpurun_out/fuzz_vector_sharded_final.log | cut -c1-700
echo "rc hostdot $rc, rc vector-sharded $rc2"
