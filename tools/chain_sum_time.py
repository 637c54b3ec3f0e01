#!/usr/bin/env python3
"""Device time of one reference-order sum of a long vector (k_chain_sums through nka_hip_debug_chain_sum): whole blocks
through the chain (walk=0) against the element-after-element walk (walk=1), on signed products (a random walk: binade
crossings) and on squares (monotone: every block accepted).  With NKA_HIP_DIAG_LIB pointing at a -DNKA_CHAIN_STAMPS build
the kernel's phases are printed too (10 ns ticks of wavefront 0).
  tools/chain_sum_time.py [n ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nka_amd  # noqa: E402

acc = nka_amd.nka(diagnostic=True).init(64, 3)
stamps = "chain_stamps" in os.environ.get("NKA_HIP_DIAG_LIB", "")
sizes = [int(float(a)) for a in sys.argv[1:]] or [10**6, 10**7, 10**8]
for n in sizes:
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    for walk, many in ((False, False), (False, True), (True, False)):
        if walk and n > 10**7:
            continue
        for what, (a, b) in (("signed", (x, y)), ("squares", (x, x))):
            acc.debug_chain_sum(a, b, 0.0, walk, many)
            s, ms = acc.debug_chain_sum(a, b, 0.0, walk, many)
            line = f"n={n:>10d} {what:8s} walk={int(walk)} many={int(many)} {ms:10.3f} ms  {1e6 * ms / n:7.3f} ns/element  sum={s!r}"
            if stamps:
                out = (C.c_double * 16)()
                acc._L.nka_hip_get_stamps(acc._handle(), out)
                names = ("load", "summary", "wait", "apply", "wait", "store", "wait")
                if many:
                    line += f"  walking us: {out[3] * 0.01:.0f}"
                else:
                    line += "  phases us: " + " ".join(f"{nm}={out[i] * 0.01:.0f}" for i, nm in enumerate(names))
                cn = ("blocks in runs", "singly", "summarised again", "walked") if many else \
                    ("blocks in runs", "on their own", "summarised again", "walked")
                line += "\n      " + ", ".join(f"{nm} {int(out[8 + i])}" for i, nm in enumerate(cn))
            print(line, flush=True)
