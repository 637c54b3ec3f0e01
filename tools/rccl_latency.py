#!/usr/bin/env python3
"""Device time of the ONE all-reduce a sharded update makes: 2+2*mvec doubles (336 B at mvec = 20)
through the library's own RCCL communicator on the kernel stream (nka_hip_allreduce_now), measured
with HIP events over back-to-back calls and -- what an update really pays -- between two kernels
(PA ... all-reduce ... scalar step) as the difference of whole updates with and without the hook.

One rank (the boxes available here have one GPU): this is the FLOOR of the latency -- launch of the
RCCL kernel and its flag protocol without any xGMI hop.  tools/scale_rehearsal.sh adds it to the
measured shard times; the first multi-GPU record replaces it with the real figure.
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import nka_amd
    from nka_amd import dist as nd
    from nka_amd import synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    m = 20
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_500_000
    acc = nka_amd.nka().init(n, m)
    which = nd.attach_allreduce(acc, 0, 1, prefer="rccl")
    print(f"# hook={which} comm (nranks, rank)={acc.comm_info()} rccl={nka_amd.nka.rccl_library()}")
    buf = torch.zeros(2 + 2 * m, dtype=torch.float64, device="cuda")
    for _ in range(20):
        acc.allreduce_now(buf)
    torch.cuda.synchronize()
    reps = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            acc.allreduce_now(buf)
        e1.record()
        torch.cuda.synchronize()
        dts.append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"all-reduce of {buf.numel()} doubles, back to back on the kernel stream: {statistics.median(dts):.1f} us each (min {min(dts):.1f})")

    # inside an update: whole updates with the hook against whole updates without, same accelerator state
    B = 48
    pool0 = torch.empty((B, n), dtype=torch.float64, device="cuda")
    for j in range(B):
        synth.fill_torch(pool0[j], 12345, j, 0, n)
    pool = pool0.clone()
    for j in range(m + 3):
        acc.accel_update(pool[j % B])
    res = {"rccl": [], "none": []}
    for r in range(6):
        for kind in (("rccl", "none") if r % 2 == 0 else ("none", "rccl")):
            if kind == "none":
                acc.drop_rccl()
            else:
                nd.attach_rccl(acc, 0, 1)
            pool.copy_(pool0)
            acc.accel_update(pool[0]); pool[0].copy_(pool0[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(B):
                acc.accel_update(pool[j])
            torch.cuda.synchronize()
            res[kind].append((time.perf_counter() - t0) / B * 1e6)
    a, b = statistics.median(res["rccl"]), statistics.median(res["none"])
    print(f"whole update at n_local={n}, m={m}: {a:.1f} us with the RCCL hook, {b:.1f} us without -> {a - b:+.1f} us per update")
    acc.delete()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
