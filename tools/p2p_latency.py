#!/usr/bin/env python3
"""What the PEER-TO-PEER EXCHANGE (include/nka_hip.h: nka_hip_p2p_*) costs an update, as far as ONE GPU can tell.

  python tools/p2p_latency.py [n_local]                       one rank: whole updates without a hook, with the RCCL hook
                                                               (one-rank communicator) and with the exchange attached for
                                                               ONE rank (own mailbox only: the floor of the fused path --
                                                               the final-sums kernel sends, the scalar step gathers)
  python -m torch.distributed.run --nproc-per-node N ... tools/p2p_latency.py share [n_local]
                                                               N ranks SHARING cuda:0 (hipIpc between processes on one
                                                               device): per-update wall time with the exchange against the
                                                               host-staged hook, at a launch-bound size.  The ranks compete for
                                                               one GPU: only the DIFFERENCE between the hooks means anything.
Over xGMI (one GPU per rank) neither figure applies: the first multi-GPU record must measure it.
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed_updates(torch, acc, pool, pool0, B, rounds=5):
    out = []
    for _ in range(rounds):
        pool.copy_(pool0)
        acc.accel_update(pool[0]); pool[0].copy_(pool0[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(B):
            acc.accel_update(pool[j])
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / B * 1e6)
    return out


def single(n):
    import torch
    import torch.distributed as dist
    import nka_amd
    from nka_amd import dist as nd
    from nka_amd import synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    m, B = 20, 48
    pool0 = torch.empty((B, n), dtype=torch.float64, device="cuda")
    for j in range(B):
        synth.fill_torch(pool0[j], 12345, j, 0, n)
    pool = pool0.clone()
    acc = nka_amd.nka().init(n, m)
    for j in range(m + 3):
        acc.accel_update(pool[j % B])
    res = {"none": [], "rccl": [], "p2p": []}
    for r in range(4):
        order = ("none", "rccl", "p2p") if r % 2 == 0 else ("p2p", "rccl", "none")
        for kind in order:
            acc.set_dot_prod(None)
            if kind == "rccl":
                nd.attach_rccl(acc, 0, 1)
            elif kind == "p2p":
                nd.attach_p2p(acc, 0, 1)
            res[kind] += timed_updates(torch, acc, pool, pool0, B, rounds=2)
            if kind == "rccl":
                acc.drop_rccl()
            elif kind == "p2p":
                acc.p2p_detach()
    base = statistics.median(res["none"])
    print(f"one rank, n_local={n}, m={m}: whole update (wall clock, {B} back to back, median of 8 blocks)")
    for kind in ("none", "rccl", "p2p"):
        v = statistics.median(res[kind])
        print(f"  hook {kind:5s}: {v:8.1f} us  ({v - base:+.1f} us)")
    acc.delete()
    dist.destroy_process_group()


def share(n):
    import torch
    import torch.distributed as dist
    import nka_amd
    from nka_amd import dist as nd
    from nka_amd import synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    m, B = 20, 64
    pool0 = torch.empty((B, n), dtype=torch.float64, device="cuda")
    for j in range(B):
        synth.fill_torch(pool0[j], 12345, j, rank * n, world * n)
    pool = pool0.clone()
    res = {}
    for kind in ("staged", "p2p", "staged", "p2p"):
        acc = nka_amd.nka().init(n, m)
        got = nd.attach_allreduce(acc, rank, world, prefer=kind, ladder=(kind,))
        assert got == kind
        for j in range(m + 3):
            acc.accel_update(pool[j % B])
        dist.barrier()
        res.setdefault(kind, [])
        res[kind] += timed_updates(torch, acc, pool, pool0, B, rounds=3)
        dist.barrier()
        acc.delete()
    if rank == 0:
        print(f"{world} ranks sharing one GPU, n_local={n}, m={m}: whole update by wall clock, median of 6 blocks of {B}")
        for kind in ("staged", "p2p"):
            print(f"  hook {kind:6s}: {statistics.median(res[kind]):8.1f} us per update (min {min(res[kind]):.1f})")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "share":
        share(int(float(sys.argv[2])) if len(sys.argv) > 2 else 4096)
    else:
        single(int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_500_000)
