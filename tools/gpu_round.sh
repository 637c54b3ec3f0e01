mkdir -p gpurun_out
bash tools/sweep.sh 2>&1 | tee gpurun_out/sweep_n_mvec.txt
