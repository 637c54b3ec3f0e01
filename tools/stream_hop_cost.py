#!/usr/bin/env python3
"""What do two cross-stream event dependencies per update cost?  (What a PB launched early on a second stream would have to pay:
PA -> PB's stream, PB -> the next update.)  Chains of three small kernels A -> B -> C per iteration, all on one stream against
B on a second stream behind an event, C behind an event back; wall time per iteration over 2000 iterations, kernels ~3 us each."""
import time

import torch

torch.cuda.set_device(0)
x = torch.zeros(1 << 16, dtype=torch.float64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 2000


def one_stream():
    with torch.cuda.stream(s1):
        for _ in range(N):
            x.add_(1.0); x.mul_(1.0); x.sub_(1.0)


def two_streams():
    for _ in range(N):
        with torch.cuda.stream(s1):
            x.add_(1.0)
            e1 = torch.cuda.Event(); e1.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(e1)
            x.mul_(1.0)
            e2 = torch.cuda.Event(); e2.record(s2)
        with torch.cuda.stream(s1):
            s1.wait_event(e2)
            x.sub_(1.0)


for name, fn in (("one stream", one_stream), ("two streams, two event dependencies per iteration", two_streams)) * 2:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{name:<52s} {1e6 * (time.perf_counter() - t0) / N:7.2f} us per iteration of three kernels", flush=True)
