bash tools/rocprof_bench.sh r02 c > gpurun_out/rocprof_c.log 2>&1; tail -4 gpurun_out/rocprof_c.log
bash tools/rocprof_bench.sh r02 f08 > gpurun_out/rocprof_f08.log 2>&1; tail -4 gpurun_out/rocprof_f08.log
bash tools/pmc_memsys.sh r02 f08 > gpurun_out/memsys_f08.log 2>&1; tail -5 gpurun_out/memsys_f08.log
bash tools/pmc_memsys.sh r02 c > gpurun_out/memsys_c.log 2>&1; tail -5 gpurun_out/memsys_c.log
