# round-4 evidence, part B: the whole GPU suite + smoke + default bench, then rocprofv3 kernel stats and PMC traffic
bash tools/gpu_check.sh || exit 1
bash tools/rocprof_bench.sh r04 c > gpurun_out/rocprof_c.log 2>&1; tail -5 gpurun_out/rocprof_c.log
bash tools/rocprof_bench.sh r04 f08 > gpurun_out/rocprof_f08.log 2>&1; tail -5 gpurun_out/rocprof_f08.log
bash tools/rocprof_bench.sh r04drops c --workload drops > gpurun_out/rocprof_drops.log 2>&1; tail -5 gpurun_out/rocprof_drops.log
ls gpurun_out/profiles_r04 gpurun_out/profiles_r04drops
