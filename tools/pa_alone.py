#!/usr/bin/env python3
"""PA (the pure-read dot pass) launched back to back on its own, per variant --
to separate the kernel's own rate from what its neighbours in an update do to it."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nka_amd  # noqa: E402
from nka_amd import synth  # noqa: E402

n, m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8, 20
acc = nka_amd.nka(diagnostic=True).init(n, m, flavor=nka_amd.FLAVOR_C)
f = torch.empty(n, dtype=torch.float64, device="cuda")
for t in range(m + 3):
    synth.fill_torch(f, 12345, t, 0, n)
    acc.accel_update(f)
synth.fill_torch(f, 12345, 99, 0, n)
L = acc._L          # libnka_hip_diag.so (nka_hip_debug_time_pa: include/nka_hip_diag.h)
for v in [int(x) for x in (sys.argv[2:] or ["0", "201", "202"])]:
    acc.set_tuning("pa_pipe", v)
    ms = C.c_float()
    for _ in range(2):
        assert L.nka_hip_debug_time_pa(acc._handle(), C.c_void_p(f.data_ptr()), 10, C.byref(ms)) == 0
    print(f"pa_pipe={v:4d}: PA alone {ms.value:.3f} ms = {8e-6 * n * 22 / ms.value:.0f} GB/s", flush=True)
