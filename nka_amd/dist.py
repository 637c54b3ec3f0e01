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


def broadcast_unique_id(make_id, rank: int, group=None):
    """Rank 0 creates the 128-byte RCCL unique id; everyone receives it through
    torch.distributed (any backend).  If rank 0 cannot create it, every rank
    receives the error text instead and raises the same exception."""
    import torch.distributed as dist
    box = [None]
    if rank == 0:
        try:
            box[0] = ("ok", make_id())
        except Exception as exc:      # noqa: BLE001 -- shipped to every rank below
            box[0] = ("error", repr(exc))
    dist.broadcast_object_list(box, src=0, group=group)
    kind, payload = box[0]
    if kind != "ok":
        raise RuntimeError(f"rank 0 could not create the RCCL unique id: {payload}")
    return payload


def all_agree(ok: bool, group=None, device=None) -> bool:
    """True iff `ok` holds on EVERY rank (a collective MIN through
    torch.distributed), so that a fallback decision is the same everywhere.
    `device`: CUDA device index of this rank (nccl backend; default: torch's current)."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        dev = f"cuda:{torch.cuda.current_device() if device is None else device}"
    else:
        dev = "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()) == 1)


def check_allreduce(acc, rank: int, world_size: int) -> bool:
    """One all-reduce through the hook installed on `acc`, on known values:
    rank r contributes (r+1)*[1, 2, 3]; every rank must read N(N+1)/2*[1, 2, 3]."""
    import torch
    t = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=f"cuda:{acc._device}") * (rank + 1)
    try:
        acc.allreduce_now(t)
        torch.cuda.synchronize(acc._device)
        want = world_size * (world_size + 1) / 2.0
        return bool((t.cpu() == torch.tensor([want, 2 * want, 3 * want], dtype=torch.float64)).all())
    except Exception:     # noqa: BLE001 -- reported through all_agree by the caller
        return False


def rccl_preflight(acc) -> bool:
    """Everything nka_hip_comm_init_rank can fail on BEFORE its blocking rendezvous
    (ncclCommInitRank waits for all ranks): RCCL can be bound in this process and the
    accelerator's device can be selected.  Local, never blocks."""
    import torch
    from .nka import nka
    try:
        nka.rccl_library()
        torch.cuda.set_device(acc._device)
        return True
    except Exception:     # noqa: BLE001 -- reported through all_agree by the caller
        return False


def attach_rccl(acc, rank: int, world_size: int, group=None):
    """Give accelerator `acc` (nka_amd.nka) its own RCCL communicator.  Callers that want a
    collective fallback go through attach_allreduce, which runs rccl_preflight on every rank
    first: a rank that cannot even reach ncclCommInitRank would leave the others blocked in it."""
    from .nka import nka
    uid = broadcast_unique_id(nka.rccl_unique_id, rank, group)
    acc.use_rccl(uid, world_size, rank)
    return acc


def attach_p2p(acc, rank: int, world_size: int, group=None):
    """The peer-to-peer exchange (include/nka_hip.h: nka_hip_p2p_export / _attach): every rank exports the hipIpc handle
    of its mailbox, the handles are gathered over `group` (any backend), every rank maps its peers.  Raises on the rank
    where a step fails; attach_allreduce turns that into a collective decision."""
    import torch.distributed as dist
    box = [None] * world_size
    try:
        mine = ("ok", acc.p2p_export(world_size))
    except Exception as exc:          # noqa: BLE001 -- every rank must still take part in the gather below
        mine = ("error", repr(exc))
    dist.all_gather_object(box, mine, group=group)
    bad = [(r, p) for r, (k, p) in enumerate(box) if k != "ok"]
    if bad:
        raise RuntimeError(f"peer-to-peer exchange: export failed on rank(s) {bad}")
    acc.p2p_attach([p for _, p in box], world_size, rank)
    return acc


def attach_allreduce(acc, rank: int, world_size: int, prefer: str = "rccl", group=None, data_group=None,
                     ladder=("rccl", "torch", "staged")) -> str:
    """Install the per-update all-reduce on `acc` and PROVE it before first use.
    Ladder, entered at `prefer`: "rccl" = the library's own communicator on the
    accelerator's stream (lowest latency); "torch" = torch.distributed's all-reduce
    on `data_group` (a callable is called -- collectively -- to create it, e.g.
    lambda: dist.new_group(backend="nccl")); "staged" = the 336 bytes staged through
    the host over `group` (any backend; slow but needs no RCCL at all).  A step that
    cannot be set up or fails its test all-reduce on ANY rank is dropped by EVERY
    rank and the next one is tried: every decision is a collective MIN over `group`
    (all_agree), so ranks can never disagree about who reduces with whom.  Keep
    `group` on gloo when the data path may be the thing that is broken (bench.py
    does).  The order inside the first step matters: ncclCommInitRank is a BLOCKING
    rendezvous, so what a rank can fail on before reaching it (binding RCCL,
    selecting its device) is checked locally and agreed on first (rccl_preflight) --
    an asymmetric early failure then moves every rank down the ladder instead of
    leaving the healthy ones waiting in the rendezvous.  Returns the hook in use.
    Raises when the last step of the ladder fails as well."""
    import sys
    steps = list(ladder)
    if prefer not in steps:
        raise ValueError(f"unknown all-reduce hook {prefer!r}")
    steps = steps[steps.index(prefer):]
    dev = acc._device
    if hasattr(acc, "set_shard"):
        acc.set_shard(rank, world_size)      # slices in rank order (slice_bounds): the sharded reference-order sums walk them so

    def say(msg):
        if rank == 0:
            print(f"[nka_amd.dist] {msg}", file=sys.stderr, flush=True)

    for i, hook in enumerate(steps):
        ok = True
        try:
            if hook == "rccl":
                ok = all_agree(rccl_preflight(acc), group, dev)
                if not ok:
                    say("RCCL pre-flight failed on at least one rank")
                else:
                    try:
                        attach_rccl(acc, rank, world_size, group)
                    except Exception as exc:      # noqa: BLE001
                        print(f"[nka_amd.dist] rank {rank}: RCCL communicator failed: {exc!r}", file=sys.stderr, flush=True)
                        ok = False
                    ok = all_agree(ok, group, dev)
            elif hook == "p2p":
                # opt-in step ahead of "rccl" (ladder=("p2p", "rccl", ...)): mailboxes mapped through hipIpc; where IPC is
                # refused on ANY rank every rank drops to the next step
                try:
                    attach_p2p(acc, rank, world_size, group)
                except Exception as exc:          # noqa: BLE001
                    print(f"[nka_amd.dist] rank {rank}: peer-to-peer exchange failed: {exc!r}", file=sys.stderr, flush=True)
                    ok = False
                ok = all_agree(ok, group, dev)
            elif hook == "torch":
                dg = data_group
                try:
                    if callable(dg):
                        dg = dg()
                    attach_torch_allreduce(acc, dg)
                except Exception as exc:          # noqa: BLE001
                    print(f"[nka_amd.dist] rank {rank}: torch.distributed data group failed: {exc!r}", file=sys.stderr, flush=True)
                    ok = False
                ok = all_agree(ok, group, dev)
            else:
                attach_staged_allreduce(acc, group)
            if ok:
                ok = all_agree(check_allreduce(acc, rank, world_size), group, dev)
        except Exception as exc:                  # noqa: BLE001 -- e.g. the control group itself failed: nothing left to agree with
            raise RuntimeError(f"all-reduce hook '{hook}': the collective decision failed: {exc!r}") from exc
        if ok:
            return hook
        try:
            if hook == "rccl":
                acc.drop_rccl()
            if hook == "p2p":
                acc.p2p_detach()
            acc.set_dot_prod(None)
        except Exception:                         # noqa: BLE001
            pass
        if i + 1 < len(steps):
            say(f"the '{hook}' all-reduce hook is unusable on at least one rank; every rank switches to '{steps[i + 1]}'")
    raise RuntimeError(f"the '{steps[-1]}' all-reduce hook failed its self-test (no hook left)")


def attach_torch_allreduce(acc, group=None):
    """Alternative hook: route the all-reduce through torch.distributed (RCCL
    under the nccl backend) on a tensor aliasing the library's device buffer.
    The collective is issued under the stream the library passes in, so it is
    ordered between the final sums of PA and the scalar step whatever stream the
    accelerator was bound to."""
    import torch
    import torch.distributed as dist

    class _Alias:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def hook(ptr, count, stream):
        if stream:
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=acc._device))
        else:       # 0 = HIP's default (null) stream
            ctx = torch.cuda.stream(torch.cuda.default_stream(acc._device))
        with ctx:
            t = torch.as_tensor(_Alias(ptr, count), device=f"cuda:{acc._device}")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    acc.set_dot_prod(hook)
    return acc


def attach_staged_allreduce(acc, group=None):
    """REHEARSAL hook: the all-reduce staged through the host over any torch.distributed
    backend (gloo): device -> host, all_reduce, host -> device, synchronous.  Slow (two PCIe hops
    and a host collective per update) -- it exists so that the multi-rank logic around the
    library (slicing, collective decisions, replica checks, bench.py --gpus N) can be executed
    with several ranks SHARING one GPU, which RCCL refuses."""
    import torch
    import torch.distributed as dist

    class _Alias:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def hook(ptr, count, stream):
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=acc._device)) if stream else \
            torch.cuda.stream(torch.cuda.default_stream(acc._device))
        with ctx:
            dev = torch.as_tensor(_Alias(ptr, count), device=f"cuda:{acc._device}")
            host = dev.cpu()                       # synchronises the stream
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            dev.copy_(host)

    acc.set_dot_prod(hook)
    return acc


def replica_digests(acc, group=None):
    """Digest of the replicated scalar state of `acc` from every rank (list of
    ints, same on all ranks).  All entries equal <=> the ranks took the same
    s == 0 / drop / slot decisions on the same all-reduced sums."""
    import torch
    import torch.distributed as dist
    d = acc.state_digest()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [d]
    dev = f"cuda:{acc._device}" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([d - (1 << 64) if d >= (1 << 63) else d], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [int(x.item()) & ((1 << 64) - 1) for x in out]
