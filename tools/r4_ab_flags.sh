# (historical: libnka_hip_noflags.so was a build with PB's two store flags compiled out, -DNKA_PB_FLAGS_OFF, a macro that existed
#  for this measurement only -- profiles/r04/ab_pb_flags.txt)
mkdir -p gpurun_out
export NKA_BENCH_SECONDARY=0
echo "== n = 1e8, m = 20: libnka_hip.so (PB with the two uniform store flags of the out-of-place entry) vs libnka_hip_noflags.so (stores unconditional)" > gpurun_out/ab_pb_flags.txt
bash tools/ab_bench.sh 4 "" libnka_hip.so libnka_hip_noflags.so >> gpurun_out/ab_pb_flags.txt 2>&1
echo "== n = 1.25e7, m = 20" >> gpurun_out/ab_pb_flags.txt
bash tools/ab_bench.sh 4 "--vlen 1.25e7 --steps 50" libnka_hip.so libnka_hip_noflags.so >> gpurun_out/ab_pb_flags.txt 2>&1
echo "== n = 1e8, f08" >> gpurun_out/ab_pb_flags.txt
bash tools/ab_bench.sh 3 "--flavor f08" libnka_hip.so libnka_hip_noflags.so >> gpurun_out/ab_pb_flags.txt 2>&1
cat gpurun_out/ab_pb_flags.txt
