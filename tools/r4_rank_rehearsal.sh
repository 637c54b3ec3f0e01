# round-4: the plain-form launcher with FOUR ranks sharing the one GPU of the box (the pool allows six processes on a card: four ranks, their torchrun agent, one spare):
# (1) host-staged hook asked for; (2) RCCL asked for -- it must refuse ranks that share a device and the ladder must end on the
# staged hook on every rank, collectively, inside the watchdog.  The numbers mean nothing (ranks share a GPU); the lines and their
# config.parallelism / ranks[*] fields are the evidence (profiles/r04/four_rank_rehearsal.txt).
mkdir -p gpurun_out
export NKA_BENCH_SHARE_GPU=1
( time timeout -k 10 280 python3 bench.py --gpus 4 --backend gloo --allreduce staged --vlen 6000001 --mvec 20 --steps 6 --warmup 24 --no-cpu-baseline ) > gpurun_out/four_rank_staged.txt 2>&1
rc1=$?; tail -4 gpurun_out/four_rank_staged.txt | cut -c1-1500
[ $rc1 -ge 124 ] && exit $rc1
( time timeout -k 10 290 python3 bench.py --gpus 4 --vlen 6000001 --mvec 20 --steps 6 --warmup 24 --no-cpu-baseline ) > gpurun_out/four_rank_rccl_asked.txt 2>&1
rc2=$?; tail -4 gpurun_out/four_rank_rccl_asked.txt | cut -c1-1500
echo "rc staged $rc1, rc rccl-asked $rc2"
