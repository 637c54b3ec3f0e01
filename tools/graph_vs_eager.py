"""Small-n update rate: eager launches vs one captured hipGraph replayed (steady state)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nka_amd
from nka_amd import synth

torch.cuda.set_device(0)
for n, m in ((10**4, 20), (10**5, 20), (10**6, 20), (10**5, 5)):
    acc = nka_amd.nka().init(n, m, flavor=nka_amd.FLAVOR_C)
    side = torch.cuda.Stream()
    K = 200
    pool = torch.empty((m + 3 + 2 * K, n), dtype=torch.float64, device="cuda")
    for t in range(pool.shape[0]):
        synth.fill_torch(pool[t], 7, t, 0, n)
    static = torch.empty(n, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(side):
        for t in range(m + 3):
            acc.accel_update(pool[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(m + 3, m + 3 + K):
            acc.accel_update(pool[t])
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / K
        g = torch.cuda.CUDAGraph()
        static.copy_(pool[m + 3 + K])
        with torch.cuda.graph(g, stream=side):
            acc.accel_update(static)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(K - 1):
            g.replay()                       # (inputs not refreshed: timing only)
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / (K - 1)
    print(f"n={n:>8} m={m:>2}: eager {1e6 * eager:7.1f} us/update   graph replay {1e6 * graph:7.1f} us/update")
