import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import nka_amd
acc = nka_amd.nka(diagnostic=True).init(64, 3)
rng = np.random.default_rng(13)
n = 60000
x = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-12, 12, n)
_ = rng.uniform(0.5, 2, n)
x2 = rng.choice([-1.0, 1.0], n) * 2.0 ** rng.integers(-1074, -1000, n).astype(np.float64)
y = rng.uniform(-1, 1, n)
_ = rng.uniform(-1, 1, n)
rare = np.where(rng.random(n) < 0.001, 1e12, 1.0) * rng.uniform(-1, 1, n)
ones = torch.ones(n, dtype=torch.float64, device='cuda')
tr = torch.from_numpy(rare).cuda()
cum = np.add.accumulate(np.concatenate([[0.0], rare]))
# per-block test: start from the true prefix, one block at a time
for b0 in range(0, n - 1024, 1024):
    start = float(cum[b0])
    got, _ = acc.debug_chain_sum(tr[b0:b0+1024].contiguous(), ones[:1024].contiguous(), start, False)
    want = float(np.add.accumulate(np.concatenate([[start], rare[b0:b0+1024]]))[-1])
    if got != want:
        print("block", b0 // 1024, "start", start.hex(), "got", got.hex(), "want", want.hex())
        np.save('/root/repo/gpurun_out/fail_block.npy', np.concatenate([[start], rare[b0:b0+1024]]))
        break
else:
    print("no single block fails from the true prefix")
