# round-4: two trees timed alternately on one box.  Needs the other tree beside this one, built:
#   git worktree add _r03 f3841c3 && make -C _r03/nka_amd/csrc -j6   (and `git worktree remove --force _r03` afterwards)
mkdir -p gpurun_out
for rep in 1 2 3; do
  for tree in . _r03; do
    timeout -k 10 120 python tools/tree_ab.py $tree 1.25e7 20 2>&1 | grep -v amdgpu.ids || exit 1
  done
done | tee gpurun_out/tree_ab_shard.txt
for rep in 1 2; do
  for tree in . _r03; do
    timeout -k 10 120 python tools/tree_ab.py $tree 1e7 10 2>&1 | grep -v amdgpu.ids || exit 1
  done
done | tee gpurun_out/tree_ab_cfg2.txt
for rep in 1 2; do
  for tree in . _r03; do
    timeout -k 10 200 python tools/tree_ab.py $tree 1e8 20 2>&1 | grep -v amdgpu.ids || exit 1
  done
done | tee gpurun_out/tree_ab_full.txt
