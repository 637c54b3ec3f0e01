mkdir -p gpurun_out
bash tools/sweep.sh > gpurun_out/sweep_n_mvec.txt 2>&1; cat gpurun_out/sweep_n_mvec.txt
bash tools/sweep.sh f08 > gpurun_out/sweep_n_mvec_f08.txt 2>&1; tail -6 gpurun_out/sweep_n_mvec_f08.txt
