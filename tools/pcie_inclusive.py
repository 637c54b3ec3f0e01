"""Host-array entry point timed with the PCIe copies of f included (n=1e8, m=20)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nka_amd
from nka_amd import synth
torch.cuda.set_device(0)
n, m = 10**8, 20
acc = nka_amd.nka().init(n, m, flavor=nka_amd.FLAVOR_C)
buf = torch.empty(n, dtype=torch.float64, device='cuda')
for t in range(m + 2):            # fill the subspace on the device path
    synth.fill_torch(buf, 1, t, 0, n); acc.accel_update(buf)
torch.cuda.synchronize()
host = np.empty(n)
pinned = torch.empty(n, dtype=torch.float64).pin_memory().numpy()
for name, arr in (("pageable", host), ("pinned", pinned)):
    dt = []
    for t in range(4):
        synth.fill_torch(buf, 1, 100 + t, 0, n); arr[:] = buf.cpu().numpy()
        t0 = time.perf_counter(); acc.accel_update(arr); dt.append(time.perf_counter() - t0)
    print(name, "host-array accel_update: median %.1f ms -> %.2f updates/s" % (1e3 * np.median(dt), 1 / np.median(dt)))
