# (libnka_hip_dead0.so = the library built with -DNKA_DEAD_SLOT_TILE0=0; libnka_hip_stamps.so = make -C nka_amd/csrc stamps)
# round-4 evidence, part A (run through gpurun): tests of the round, dead ring slots A/B, phase stamps incl. PB and the kernel boundaries
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_round4.py tests/test_hip_round2.py -x -q --tb=short > gpurun_out/pytest_r4.log 2>&1 || { tail -40 gpurun_out/pytest_r4.log; exit 1; }
tail -2 gpurun_out/pytest_r4.log
export NKA_BENCH_SECONDARY=0
echo "== full subspace (no dead slot): libnka_hip.so = dead slots re-read f's FIRST tile; libnka_hip_dead0.so = the tile at hand (before)" > gpurun_out/ab_dead_slot.txt
bash tools/ab_bench.sh 3 "" libnka_hip.so libnka_hip_dead0.so >> gpurun_out/ab_dead_slot.txt 2>&1
echo "== drops workload (one dead PB slot per update)" >> gpurun_out/ab_dead_slot.txt
bash tools/ab_bench.sh 3 "--workload drops" libnka_hip.so libnka_hip_dead0.so >> gpurun_out/ab_dead_slot.txt 2>&1
echo "== n = 1.25e7 (the 8-GPU shard), full subspace" >> gpurun_out/ab_dead_slot.txt
bash tools/ab_bench.sh 3 "--vlen 1.25e7 --steps 50" libnka_hip.so libnka_hip_dead0.so >> gpurun_out/ab_dead_slot.txt 2>&1
echo "== n = 1e7, m = 10" >> gpurun_out/ab_dead_slot.txt
bash tools/ab_bench.sh 3 "--vlen 1e7 --mvec 10 --steps 50" libnka_hip.so libnka_hip_dead0.so >> gpurun_out/ab_dead_slot.txt 2>&1
cat gpurun_out/ab_dead_slot.txt
export NKA_HIP_DIAG_LIB=$PWD/nka_amd/libnka_hip_stamps.so
: > gpurun_out/solve_phases_r04.txt
for nm in "1e7 10" "1.25e7 20" "1e5 20" "1e8 20"; do
  set -- $nm
  echo "## n = $1, mvec = $2" >> gpurun_out/solve_phases_r04.txt
  timeout -k 10 300 python tools/solve_phases.py --vlen $1 --mvec $2 >> gpurun_out/solve_phases_r04.txt 2>&1
done
cat gpurun_out/solve_phases_r04.txt
