#!/bin/bash
# tools/evidence.sh STEP [args] -- the GPU-box round trips behind profiles/rNN/, one parameterised script (round 5: the fifteen
# one-off tools/r4_*.sh and the r5 scripts folded into it).  Run through gpurun from the repository root, e.g.
#     gpurun --timeout 900 -- 'bash tools/evidence.sh suite'
# Every step writes under gpurun_out/ (scratch); copy what is to be judged into profiles/rNN/.  A step that is killed by its
# own timeout starts nothing further (exit code >= 124 is passed on).
#
#   suite                       pytest -m gpu, smoke(), default bench.py                       (tools/gpu_check.sh)
#   rocprof TAG                 rocprofv3 kernel stats + PMC traffic of bench.py: compact, src-F08 rounding, drops workload
#   soak MODE SECONDS SEED [W]  tools/fuzz_gpu.py: MODE = array | vector | hostdot | sharded | vector-sharded, W ranks
#   soak-paired MODE "F:N ..." [W]  the same seeds once per fast sum mode (NKA_FUZZ_FORCE_SUMS)   (tools/soak_compare.py)
#   soak-seeds SEEDS...         flagged seeds in place against out of place, bit for bit       (tools/swap_vs_inplace_seed.py)
#   list-word                   tests of the list word + its in-process A/B, drops workload
#   swap                        tests of the out-of-place entry + tools/ab_swap.py
#   small-n                     scalar-step phases (stamps build) + sweep n = 1e4..1e6
#   reforder                    whole GPU suite, cost table of the reference-order sums, soak over the three sum modes
#   rank-rehearsal              bench.py --gpus 4 in the plain form, ranks sharing the GPU: staged asked / rccl asked
#   scale                       one-GPU rehearsal of every shard size + sweep                   (tools/scale_rehearsal.sh, sweep.sh)
#   mall                        round 5: Infinity-Cache reuse between the passes (needs `make -C nka_amd/csrc ftemporal`)
#   p2p                         round 5: the peer-to-peer exchange, latency as far as one GPU can tell
#   lib-ab ROUNDS "ARGS" LIBS   interleaved bench.py A/B of several builds of libnka_hip.so     (tools/ab_bench.sh)
set -o pipefail
mkdir -p gpurun_out
step=$1; shift || true
pass_on() { [ "$1" -ge 124 ] && exit "$1"; return 0; }

case "$step" in
suite)
  bash tools/gpu_check.sh ;;
rocprof)
  tag=${1:-r05}
  bash tools/rocprof_bench.sh $tag c > gpurun_out/rocprof_c.log 2>&1; tail -5 gpurun_out/rocprof_c.log
  bash tools/rocprof_bench.sh $tag f08 > gpurun_out/rocprof_f08.log 2>&1; tail -5 gpurun_out/rocprof_f08.log
  bash tools/rocprof_bench.sh ${tag}drops c --workload drops > gpurun_out/rocprof_drops.log 2>&1; tail -5 gpurun_out/rocprof_drops.log
  ls gpurun_out/profiles_$tag gpurun_out/profiles_${tag}drops ;;
soak)
  mode=${1:-array}; secs=${2:-240}; seed=${3:-0}; world=${4:-3}
  case "$mode" in
    array) flags="" ;; vector) flags="--vector" ;; hostdot) flags="--hostdot" ;;
    sharded) flags="--sharded $world" ;; vector-sharded) flags="--vector-sharded $world" ;;
    *) echo "soak: unknown mode $mode"; exit 2 ;;
  esac
  out=gpurun_out/fuzz_${mode}_${seed}.txt
  timeout -k 10 $((secs + 120)) python tools/fuzz_gpu.py --seconds $secs --first-seed $seed $flags --out $out > ${out%.txt}.log 2>&1
  rc=$?; tail -2 ${out%.txt}.log | cut -c1-900; grep -h -A12 "^FAIL" $out* 2>/dev/null | head -40
  exit $rc ;;   # (2 = a sequence of more than 512 elements beyond the allowance that tests/golden/soak_cases.json does not list)
soak-paired)
  # round 6: the SAME seeds once per fast sum mode -- raw-sum Gram row (blocked) / Gram row on the rounded
  # w1' (rounded) -- for tools/soak_compare.py.  soak-paired MODE "FIRST:COUNT FIRST:COUNT ..." [W]; array and vector run the two
  # modes side by side (two processes), sharded one after the other (3 ranks each).
  mode=${1:-array}; spans=${2:-"0:100"}; world=${3:-3}
  case "$mode" in
    array) flags="" ;; vector) flags="--vector" ;; sharded) flags="--sharded $world" ;; vector-sharded) flags="--vector-sharded $world" ;;
    *) echo "soak-paired: unknown mode $mode"; exit 2 ;;
  esac
  mkdir -p gpurun_out/paired
  one() {   # one SUMS FIRST COUNT
    out=gpurun_out/paired/fuzz_${mode}_$2_$1.txt
    NKA_FUZZ_FORCE_SUMS=$1 timeout -k 10 1100 python tools/fuzz_gpu.py --seeds $3 --first-seed $2 $flags --out $out > ${out%.txt}.log 2>&1
    rc=$?; echo "$mode $1 seeds $2+$3: rc $rc; $(tail -1 ${out%.txt}.log | cut -c1-300)"; return $rc
  }
  for span in $spans; do
    first=${span%%:*}; count=${span##*:}
    if [ "$mode" = sharded ] || [ "$mode" = vector-sharded ]; then
      one blocked $first $count; pass_on $?
      one rounded $first $count; pass_on $?
    else
      one blocked $first $count & p1=$!
      one rounded $first $count & p2=$!
      wait $p1; r1=$?; wait $p2; r2=$?; pass_on $r1; pass_on $r2
    fi
  done ;;
soak-seeds)
  timeout -k 10 300 python tools/swap_vs_inplace_seed.py "$@" 2>&1 | tee gpurun_out/swap_vs_inplace_seed.txt | cut -c1-250 ;;
list-word)
  timeout -k 10 600 python -m pytest tests/test_hip_round4.py -x -q --tb=short > gpurun_out/pytest_r4.log 2>&1 || { tail -40 gpurun_out/pytest_r4.log; exit 1; }
  tail -3 gpurun_out/pytest_r4.log
  for spec in "c 1e8 12 4 8" "f08 1e8 12 4 8" "c 1e8 5 4 8" "c 1.25e7 12 6 10"; do
    set -- $spec
    timeout -k 10 300 python tools/ab_inproc.py --key list_word --values 0 1 --flavor $1 --vlen $2 --mvec 20 --span-dim $3 --rounds $4 --steps $5 \
      > gpurun_out/ab_list_word_$1_$2_d$3.txt 2>&1; pass_on $?; tail -4 gpurun_out/ab_list_word_$1_$2_d$3.txt
  done
  NKA_BENCH_SECONDARY=0 timeout -k 10 300 python bench.py --workload drops --no-cpu-baseline > gpurun_out/bench_drops.log 2>&1; tail -1 gpurun_out/bench_drops.log | cut -c1-1500 ;;
swap)
  timeout -k 10 900 python -m pytest tests/test_hip_round4.py tests/test_hip_round5.py -x -q --tb=short > gpurun_out/pytest_swap.log 2>&1 || { tail -40 gpurun_out/pytest_swap.log; exit 1; }
  tail -3 gpurun_out/pytest_swap.log
  for spec in "c 1e8 20 4 8" "f08 1e8 20 4 8" "c 1.25e7 20 6 10" "c 1e7 10 6 10" "c 1e5 20 6 20"; do
    set -- $spec
    timeout -k 10 300 python tools/ab_swap.py --flavor $1 --vlen $2 --mvec $3 --rounds $4 --steps $5 > gpurun_out/ab_swap_$1_$2.txt 2>&1; pass_on $?
    tail -4 gpurun_out/ab_swap_$1_$2.txt
  done ;;
small-n)
  export NKA_HIP_DIAG_LIB=$PWD/nka_amd/libnka_hip_stamps.so       # make -C nka_amd/csrc stamps
  for nm in "1e5 20" "1e5 10" "1.25e7 20"; do set -- $nm; echo "## n = $1, mvec = $2"; timeout -k 10 200 python tools/solve_phases.py --vlen $1 --mvec $2 2>&1 | grep -v amdgpu; done
  unset NKA_HIP_DIAG_LIB
  echo "n mvec updates/s us/update frac frac PA solve PB"
  for m in 10 20; do for n in 1e4 1e5 1e6; do for rep in 1 2; do
    NKA_BENCH_SECONDARY=0 python bench.py --no-cpu-baseline --vlen $n --mvec $m --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$n', $m, round(d['value'],1), round(1e3*d['ms_per_step'],1), round(r['frac'],3), round(r['whole_update']['frac'],3), round(1e3*k['PA_k_dots']['mean_ms'],1), round(1e3*k['k_solve']['mean_ms'],1), round(1e3*k['PB_k_combine']['mean_ms'],1))"
  done; done; done ;;
reforder)
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/pytest_gpu.log 2>&1
  rc=$?; grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" gpurun_out/pytest_gpu.log | grep -E "passed|failed|Error|error|assert|FAILED" | tail -15
  [ $rc -eq 0 ] || exit $rc
  timeout -k 10 300 python tools/sum_order_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sum_order_cost.txt; pass_on ${PIPESTATUS[0]}
  bash tools/evidence.sh soak array 200 9000 ;;
rank-rehearsal)
  export NKA_BENCH_SHARE_GPU=1
  ( time timeout -k 10 280 python3 bench.py --gpus 4 --backend gloo --allreduce staged --vlen 6000001 --mvec 20 --steps 6 --warmup 24 --no-cpu-baseline ) > gpurun_out/four_rank_staged.txt 2>&1
  rc1=$?; tail -4 gpurun_out/four_rank_staged.txt | cut -c1-1500; pass_on $rc1
  ( time timeout -k 10 290 python3 bench.py --gpus 4 --vlen 6000001 --mvec 20 --steps 6 --warmup 24 --no-cpu-baseline ) > gpurun_out/four_rank_rccl_asked.txt 2>&1
  rc2=$?; tail -4 gpurun_out/four_rank_rccl_asked.txt | cut -c1-1500
  echo "rc staged $rc1, rc rccl-asked $rc2" ;;
scale)
  bash tools/scale_rehearsal.sh > /dev/null 2>&1; cat gpurun_out/scale_rehearsal.txt
  bash tools/sweep.sh c > gpurun_out/sweep_n_mvec.txt 2>&1; cat gpurun_out/sweep_n_mvec.txt ;;
mall)
  # (a) temporal loads for the vectors both passes read (libnka_hip_diag_ft<bits>.so: 1 = in PA, 2 = in PB, 3 = both),
  # (b) PB in the reverse of PA's tile order (pb_reverse), (c) both: profiles/r05/ab_mall_reuse.txt
  D=nka_amd/libnka_hip_diag
  out=gpurun_out/r5_mall_ab.txt; : > $out
  for spec in "c 1.25e7 20" "f08 1.25e7 20" "c 1e7 10" "f08 1e7 10" "c 5e6 20" "c 1e8 20"; do
    set -- $spec
    echo "=== flavor $1 n $2 m $3: base build against temporal loads in both passes, each with pb_reverse 0 / 1" | tee -a $out
    python tools/ab_libs.py --libs $D.so ${D}_ft3.so --combos pb_reverse=0 pb_reverse=1 --flavor $1 --vlen $2 --mvec $3 \
        --rounds 10 --steps 16 --check-bits 2>&1 | grep -v amdgpu.ids | tee -a $out; pass_on ${PIPESTATUS[0]}
  done
  for spec in "c 1.25e7 20" "c 1e7 10" "f08 1.25e7 20" "c 1e8 20"; do
    set -- $spec
    echo "=== flavor $1 n $2 m $3: which pass's policy matters (ft1 = PA only, ft2 = PB only, ft3 = both)" | tee -a $out
    python tools/ab_libs.py --libs $D.so ${D}_ft1.so ${D}_ft2.so ${D}_ft3.so --combos pb_reverse=0 --flavor $1 --vlen $2 --mvec $3 \
        --rounds 10 --steps 16 --check-bits 2>&1 | grep -v amdgpu.ids | tee -a $out; pass_on ${PIPESTATUS[0]}
  done ;;
p2p)
  out=gpurun_out/r5_p2p_latency.txt; : > $out
  python tools/p2p_latency.py 12500000 2>&1 | grep -E "one rank|hook " | tee -a $out
  python tools/p2p_latency.py 100000 2>&1 | grep -E "one rank|hook " | tee -a $out
  for W in 2 4; do for N in 4096 1000000; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port 29577 \
        tools/p2p_latency.py share $N 2>&1 | grep -E "ranks sharing|hook " | tee -a $out
  done; done ;;
lib-ab)
  bash tools/ab_bench.sh "$@" ;;
*)
  sed -n 2,28p "$0"; exit 2 ;;
esac
