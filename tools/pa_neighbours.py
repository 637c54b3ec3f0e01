#!/usr/bin/env python3
"""Why is PA slower inside an update than launched back to back on its own?  One accelerator (diagnostic build), n = 1e8,
m = 20, steady state; PA's device time (HIP events of the library) under four neighbourhoods:
  V0  updates back to back, inputs rotating over 30 resident rows (what bench.py does)
  V1  a stream synchronisation and 2 ms of idle between updates (does PB's write drain reach into PA?)
  V2  the SAME input buffer for every update (does the address translation of f matter?)
  V3  a 4 GB device-to-device copy between updates (caches and translations of somebody else's pages)
and PA alone, launched ten times back to back (nka_hip_debug_time_pa)."""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nka_amd  # noqa: E402
from nka_amd import synth  # noqa: E402

n, m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8, 20
acc = nka_amd.nka(diagnostic=True).init(n, m, flavor=nka_amd.FLAVOR_C)
P = 30
pool = torch.empty((P, n), dtype=torch.float64, device="cuda")
for j in range(P):
    synth.fill_torch(pool[j], 12345, j, 0, n)
junk = torch.empty(2 * 10**9 // 8 * 2, dtype=torch.float64, device="cuda")      # 4 GB
t = 0
for _ in range(m + 3):
    acc.accel_update(pool[t % P]); t += 1
assert acc.num_vec() == m
K = 12


def run(label, between, same=False):
    global t
    acc.set_timing(K)
    for _ in range(K):
        acc.accel_update(pool[0] if same else pool[t % P]); t += 1
        between()
    torch.cuda.synchronize()
    ph = [acc.timing_ms(b) for b in range(K)]
    pa, pb = [p[0] for p in ph], [p[2] for p in ph]
    print(f"{label:<62s} PA {statistics.mean(pa):.3f} ms (min {min(pa):.3f})   PB {statistics.mean(pb):.3f} (min {min(pb):.3f})", flush=True)


def idle():
    torch.cuda.synchronize()
    time.sleep(0.002)


for rep in range(2):
    run("V0 back to back, rotating inputs", lambda: None)
    run("V1 synchronise + 2 ms idle between updates", idle)
    run("V2 the same input buffer every update", lambda: None, same=True)
    run("V3 a 4 GB device copy between updates", lambda: junk[:junk.numel() // 2].copy_(junk[junk.numel() // 2:]))
    ms = C.c_float()
    f = pool[t % P]
    for _ in range(2):
        assert acc._L.nka_hip_debug_time_pa(acc._handle(), C.c_void_p(f.data_ptr()), 10, C.byref(ms)) == 0
    print(f"{'PA alone, ten launches back to back':<62s} PA {ms.value:.3f} ms = {8e-6 * n * 22 / ms.value:.0f} GB/s", flush=True)
