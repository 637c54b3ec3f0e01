"""Worker of tests/test_p2p_exchange.py: the PEER-TO-PEER EXCHANGE (include/nka_hip.h: nka_hip_p2p_export / _attach) with
N ranks sharing cuda:0 -- mailboxes in fine-grained device memory mapped through hipIpc between the processes, the final-sums
kernel of an update writing into every rank's mailbox, the scalar step gathering in rank order.

Every rank drives TWO accelerators on its slice: `a` with the peer-to-peer exchange, `b` with a host-staged hook that adds
the ranks' rows in the same (rank) order.  After every call: outputs equal bit for bit, state digests equal between a and
b and across the ranks, decisions equal the unsharded oracle's.  Then: the ladder falls through to the next hook on EVERY
rank when one rank cannot export its mailbox, and a rank whose peers never send gets NKA_HIP_ECOMM after a bounded wait
instead of a hang."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nka_amd  # noqa: E402
from nka_amd import dist as nd  # noqa: E402
from nka_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


class _Alias:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)

    def rank_ordered_staged(ptr, count, stream):
        dev = torch.as_tensor(_Alias(ptr, count), device="cuda")
        host = dev.cpu()
        rows = [torch.zeros_like(host) for _ in range(world)]
        dist.all_gather(rows, host)
        acc = rows[0].clone()
        for r in range(1, world):
            acc += rows[r]                       # rank order: the order of the peer-to-peer gather
        dev.copy_(acc)

    # sums: NKA_HIP_SUMS_BLOCKED = the single-pass fast mode, whose final sums go STRAIGHT into the mailboxes and whose scalar
    # step gathers them (the fused exchange); the default (round 6: the norm first, then the rows on the rounded w1') makes its
    # two exchanges per update through the same mailboxes with the send-and-gather kernel
    for (n, m, flavor, calls, sums) in ((200003, 6, 2, 18, nka_amd.SUMS_BLOCKED), (65536, 20, 0, 30, nka_amd.SUMS_BLOCKED),
                                         (1000, 40, 2, 12, nka_amd.SUMS_BLOCKED), (200003, 6, 2, 18, nka_amd.SUMS_AUTO),
                                         (65536, 20, 0, 24, nka_amd.SUMS_AUTO)):
        lo, hi = nd.slice_bounds(n, world, rank)
        a = nka_amd.nka().init(hi - lo, m, flavor=flavor).set_sum_order(sums)
        b = nka_amd.nka().init(hi - lo, m, flavor=flavor).set_sum_order(sums)
        hook = nd.attach_allreduce(a, rank, world, prefer="p2p", ladder=("p2p", "staged"))
        assert hook == "p2p", hook
        b.set_dot_prod(rank_ordered_staged)
        full = O.OracleNKA(n, m, flavor)
        basis = np.stack([synth.fill_numpy(3, 50 + j, 0, n, n) for j in range(3)])
        for t in range(calls):
            x = (synth.fill_numpy(5, t, 0, 3, 3) @ basis) if t % 5 == 3 else synth.fill_numpy(777, t, 0, n, n)
            if t == 7:
                x = prev.copy()                  # repeated input: s == 0 on every rank at once
            prev = x
            f = x.copy()
            full.accel_update(f)
            fa = torch.from_numpy(x[lo:hi].copy()).cuda()
            fb = fa.clone()
            a.accel_update(fa)
            b.accel_update(fb)
            torch.cuda.synchronize()
            assert torch.equal(fa, fb), (rank, n, m, t, float((fa - fb).abs().max()))
            assert a.state_digest() == b.state_digest(), (rank, n, m, t)
            digs = nd.replica_digests(a)
            assert all(d == digs[0] for d in digs), (rank, n, m, t)
            assert a.num_vec() == full.num_vec() and a.state().list_order() == full.state().list_order(), (rank, n, m, t)
            assert np.array_equal(a.reductions(), b.reductions()), (rank, n, m, t)
            if t == 11:
                a.relax(); b.relax(); full.relax()
        assert a.defined() and b.defined()
        dist.barrier()                           # (nobody frees a mailbox a peer may still write into)
        a.delete(); b.delete()

    # ---- a captured update replays the exchange: its number lives on the device, not in a kernel argument -----------
    n, m = 65536, 4
    lo, hi = nd.slice_bounds(n, world, rank)
    side = torch.cuda.Stream()
    static = torch.empty(hi - lo, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(side):
        a = nka_amd.nka().init(hi - lo, m).set_sum_order(nka_amd.SUMS_BLOCKED)      # (the fused exchange: the gather inside the scalar step)
        assert nd.attach_allreduce(a, rank, world, prefer="p2p", ladder=("p2p",)) == "p2p"
        b = nka_amd.nka().init(hi - lo, m).set_sum_order(nka_amd.SUMS_BLOCKED)
        b.set_dot_prod(rank_ordered_staged)
        for t in range(m + 3):
            x = synth.fill_numpy(31, t, 0, n, n)
            static.copy_(torch.from_numpy(x[lo:hi].copy()))
            fb = static.clone()
            a.accel_update(static)
            b.accel_update(fb)
            torch.cuda.synchronize()
            assert torch.equal(static, fb), (rank, t)
        assert a.capture_safe()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            a.accel_update(static)
        for t in range(m + 3, m + 9):
            x = synth.fill_numpy(31, t, 0, n, n)
            static.copy_(torch.from_numpy(x[lo:hi].copy()))
            fb = static.clone()
            g.replay()
            b.accel_update(fb)
            torch.cuda.synchronize()
            assert torch.equal(static, fb), (rank, "replay", t)
            assert a.state_digest() == b.state_digest(), (rank, "replay", t)
    dist.barrier()
    del g
    a.delete(); b.delete()

    # ---- the same mailboxes as the transport of the reference-order chain: the single-rank reference's bits ------
    n, m = 5003, 5
    lo, hi = nd.slice_bounds(n, world, rank)
    e = nka_amd.nka().init(hi - lo, m, flavor=0)
    assert nd.attach_allreduce(e, rank, world, prefer="p2p", ladder=("p2p",)) == "p2p"
    e.set_sum_order(nka_amd.SUMS_REFERENCE_ORDER)
    full = O.OracleNKA(n, m, 0)
    for t in range(10):
        x = synth.fill_numpy(99, t, 0, n, n)
        f = x.copy()
        full.accel_update(f)
        ft = torch.from_numpy(x[lo:hi].copy()).cuda()
        e.accel_update(ft)
        assert np.array_equal(ft.cpu().numpy(), f[lo:hi]), (rank, t)
    dist.barrier()
    e.delete()

    # ---- the ladder: one rank cannot export -> EVERY rank runs the next hook ---------------------------------
    c = nka_amd.nka().init(1000, 3)
    if rank == world - 1:
        def refused(nranks):
            raise nka_amd.NKAError("hipIpcGetMemHandle: refused (pretend)")
        c.p2p_export = refused
    hook = nd.attach_allreduce(c, rank, world, prefer="p2p", ladder=("p2p", "staged"))
    assert hook == "staged", hook
    t = torch.ones(1000, dtype=torch.float64, device="cuda")
    c.accel_update(t)
    c.delete()

    # ---- a peer that never sends: bounded wait, NKA_HIP_ECOMM at the next synchronising call, no hang ---------
    os.environ["NKA_HIP_P2P_TIMEOUT_MS"] = "300"
    d = nka_amd.nka().init(4096, 3)
    assert nd.attach_allreduce(d, rank, world, prefer="p2p", ladder=("p2p",)) == "p2p"
    x = torch.ones(4096, dtype=torch.float64, device="cuda")
    d.accel_update(x.clone())                    # (the first update has no sums to exchange)
    d.accel_update(2 * x)                        # everybody: a real exchange
    assert d.num_vec() == 1
    dist.barrier()
    if rank == 0:
        d.accel_update(3 * x)                    # alone: the peers' rows never arrive
        try:
            d.num_vec()
            raise AssertionError("expected NKA_HIP_ECOMM after the bounded wait")
        except nka_amd.NKAError as exc:
            assert "did not arrive in time" in str(exc), str(exc)
    dist.barrier()
    d.delete()
    print(f"rank {rank}/{world} p2p OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
