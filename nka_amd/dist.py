"""Sharding of the accel_update path over ranks (one process per GPU).

The reference's parallel contract (src-F08/nka_type.F90:58-64): every rank calls
the same sequence collectively on ITS slice of the vectors and supplies a global
dot product.  Here the slice is contiguous, the per-rank partial dot products
stay on the GPU, and the only exchange is the SUM of those partials: ONE
all-reduce of 2+2*mvec doubles per update (the norm and both Gram rows, after the
pure-read pass PA), by RCCL over xGMI on the accelerator's own stream.  H, c and the lists are replicated; they stay
bitwise identical because every rank receives the same all-reduced bits.
"""
from __future__ import annotations


def slice_bounds(n_global: int, world_size: int, rank: int):
    """Contiguous slice [lo, hi) of rank `rank`: r*n/P .. (r+1)*n/P.
    Sizes differ by at most one; lo is kept even so that 16-byte vector loads
    stay aligned on a slice of a 16-byte aligned global array."""
    if not (0 <= rank < world_size) or n_global < 0:
        raise ValueError("bad slice request")
    lo = (rank * n_global) // world_size
    hi = ((rank + 1) * n_global) // world_size
    if rank > 0:
        lo -= lo % 2
    if rank + 1 < world_size:
        hi -= hi % 2
    return lo, hi


def broadcast_unique_id(make_id, rank: int, group=None) -> bytes:
    """Rank 0 creates the 128-byte RCCL unique id; everyone receives it through
    torch.distributed (any backend)."""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


def attach_rccl(acc, rank: int, world_size: int, group=None):
    """Give accelerator `acc` (nka_amd.nka) its own RCCL communicator."""
    from .nka import nka
    uid = broadcast_unique_id(nka.rccl_unique_id, rank, group)
    acc.use_rccl(uid, world_size, rank)
    return acc


def attach_torch_allreduce(acc, group=None):
    """Alternative hook: route the all-reduce through torch.distributed (RCCL
    under the nccl backend) on a tensor aliasing the library's device buffer."""
    import torch
    import torch.distributed as dist

    class _Alias:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def hook(ptr, count, stream):
        t = torch.as_tensor(_Alias(ptr, count), device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    acc.set_dot_prod(hook)
    return acc
