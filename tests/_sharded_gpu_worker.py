"""Worker of tests/test_hip_parity.py::test_sharded_hip_path_two_ranks_one_gpu.

Two ranks share cuda:0 (the boxes available to the tests have one GPU).  Each
rank runs the PRODUCT path -- libnka_hip.so through nka_amd.nka -- on its
contiguous slice; the distribution hook (nka_hip_set_allreduce, the device-side
set_dot_prod) sums the partial inner products of the two ranks through
torch.distributed/gloo, staging the 2+2*mvec doubles through the host.  RCCL
refuses two ranks on one device, so the built-in RCCL hook is exercised
single-rank elsewhere; everything else of the N > 1 path is what runs here."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import nka_amd  # noqa: E402
from nka_amd import dist as nd  # noqa: E402
from nka_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as P  # noqa: E402


class _Alias:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def main():
    # NKA_TEST_RCCL=1 (two-GPU boxes): one GPU per rank and the library's own RCCL
    # all-reduce on the kernel stream, installed and proven by nd.attach_allreduce;
    # default: both ranks share cuda:0 and the hook stages through gloo (see above)
    use_rccl = os.environ.get("NKA_TEST_RCCL") == "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank if use_rccl else 0)
    n, m, calls = 200003, 6, 16
    lo, hi = nd.slice_bounds(n, world, rank)
    counts = []

    def hook(ptr, count, stream):
        dev = torch.as_tensor(_Alias(ptr, count), device="cuda")
        host = dev.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        dev.copy_(host)
        counts.append(count)

    for flavor in (nka_amd.FLAVOR_F08, nka_amd.FLAVOR_C):
        acc = nka_amd.nka().init(hi - lo, m, flavor=flavor)
        if use_rccl:
            which = nd.attach_allreduce(acc, rank, world, prefer="rccl")
            assert which == "rccl", which
        else:
            acc.set_dot_prod(hook)
        full = O.OracleNKA(n, m, flavor)
        spread = P.Spread(O, n, m)           # the reference's own inter-flavour spread: tolerance rule of parity_util
        basis = np.stack([synth.fill_numpy(3, 50 + j, 0, n, n) for j in range(3)])
        for t in range(calls):
            x = (synth.fill_numpy(5, t, 0, 3, 3) @ basis) if t % 5 == 3 else synth.fill_numpy(12345, t, 0, n, n)
            f_full = x.copy()
            full.accel_update(f_full)
            spread.update(x)
            ft = torch.from_numpy(x[lo:hi].copy()).cuda()
            if flavor == nka_amd.FLAVOR_C and t % 2 == 1:
                # every other update of the second pass OUT OF PLACE (nka_hip_accel_update_swap): the same collective in
                # between, the same bits out, on every rank (an empty slice hands over an empty buffer)
                _, acc_f = acc.accel_update_swap(ft)
                out = acc_f.cpu().numpy()
            else:
                acc.accel_update(ft)
                out = ft.cpu().numpy()
            assert acc.num_vec() == full.num_vec(), (rank, flavor, t)
            st = acc.state()
            assert st.list_order() == full.state().list_order(), (rank, flavor, t)
            # replicated scalars: bitwise identical on both ranks
            c = torch.from_numpy(st.c.copy())
            cmax, cmin = c.clone(), c.clone()
            dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
            assert torch.equal(cmax, cmin), (rank, flavor, t)
            digs = nd.replica_digests(acc)
            assert all(d == digs[0] for d in digs), (rank, flavor, t, digs)
            err = np.linalg.norm(out - f_full[lo:hi]) / np.linalg.norm(x)
            P.check(err, st, f"sharded rank {rank} flavor {flavor}", where=t, spread=spread.value,
                    truth=spread.truth(out, x, sl=slice(lo, hi)))
            if t == 9:
                acc.relax(); full.relax(); spread.relax()
        assert acc.defined()
        acc.delete()
    if not use_rccl:
        assert set(counts) <= {1, 1 + 2 * m, 2 + 2 * m} and counts      # (default sums: the norm, then the rows; NKA_HIP_SUMS_BLOCKED: one exchange)
    print(f"rank {rank}/{world} slice [{lo},{hi}) hook={'rccl' if use_rccl else 'gloo-staged'} OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
