#!/usr/bin/env python3
"""One soak seed (tools/fuzz_ops.py) twice through the C ABI: every update in place, and with the out-of-place updates the soak
tool mixes into that seed -- every output and the state digest after every operation compared BIT FOR BIT.
  tools/swap_vs_inplace_seed.py SEED [SEED ...]
  tools/swap_vs_inplace_seed.py --sharded W SEED [SEED ...]     the sharded soak's sequence of that seed (tools/fuzz_gpu.py
        one_seed_sharded) on W ranks sharing cuda:0, the partial sums staged through gloo: per rank, the slice's output of the
        in-place accelerator against the one that takes two of three updates out of place, as the soak does on odd seeds"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import nka_amd  # noqa: E402
from fuzz_ops import array_ops, array_shape  # noqa: E402

class _Alias:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def sharded_rank(seeds, steps=60):
    import torch.distributed as dist
    from nka_amd import dist as nd
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)

    def hook(ptr, count, stream):
        dev = torch.as_tensor(_Alias(ptr, count), device="cuda")
        host = dev.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        dev.copy_(host)

    for seed in seeds:
        rng = np.random.default_rng(90_000 + seed)       # the draws of fuzz_gpu.one_seed_sharded, in its order
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 255, 512, 513, 1025, 2049, 4099])) if rng.random() < 0.7 else int(rng.integers(1, 9000))
        m = int(rng.integers(1, 25))
        flavor = int(rng.integers(0, 3))
        lo, hi = nd.slice_bounds(n, world, rank)
        a, b = nka_amd.nka().init(hi - lo, m, flavor=flavor), nka_amd.nka().init(hi - lo, m, flavor=flavor)
        a.set_dot_prod(hook); b.set_dot_prod(hook)
        basis = rng.standard_normal((3, n))
        prev = rng.standard_normal(n)
        nswap = 0
        for step in range(steps):
            r = rng.random()
            if r < 0.80:
                kind = rng.random()
                x = rng.standard_normal(n) if kind < 0.55 else rng.standard_normal(3) @ basis if kind < 0.85 else prev.copy() if kind < 0.95 else np.zeros(n)
                prev = x
                fa, fb = torch.from_numpy(x[lo:hi].copy()).cuda(), torch.from_numpy(x[lo:hi].copy()).cuda()
                a.accel_update(fa)
                if step % 3 != 0:
                    _, fb = b.accel_update_swap(fb)
                    nswap += 1
                else:
                    b.accel_update(fb)
                assert torch.equal(fa, fb), (seed, rank, step, float((fa - fb).abs().max()))
            elif r < 0.87:
                a.relax(); b.relax()
            elif r < 0.91:
                a.restart(); b.restart()
            elif r < 0.96:
                vt = float(10.0 ** rng.uniform(-3, -0.3))
                a.set_vec_tol(vt); b.set_vec_tol(vt)
            else:
                a, b = a.copy(), b.copy()
            assert a.state_digest() == b.state_digest(), (seed, rank, step)
        print(f"sharded seed {seed} rank {rank}/{world}: n={n} m={m} flavor {flavor} slice [{lo},{hi}), {steps} operations, {nswap} updates out of "
              f"place: every output and every state digest bit-identical to the in-place run", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if "--sharded-rank" in sys.argv:
    sharded_rank([int(s) for s in sys.argv[sys.argv.index("--sharded-rank") + 1:]])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == "--sharded":
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    sys.exit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(sys.argv[2])}",
                             "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--sharded-rank"] + sys.argv[3:],
                            env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")).returncode)


for seed in [int(s) for s in sys.argv[1:]]:
    rng, n, m, flavor = array_shape(seed)
    ops = list(array_ops(rng, n, 120))
    rng_swap = np.random.default_rng(777_000 + seed)
    share = float(rng_swap.choice([0.3, 0.7, 1.0]))
    a, b = nka_amd.nka().init(n, m, flavor=flavor), nka_amd.nka().init(n, m, flavor=flavor)
    nswap = 0
    for step, op in enumerate(ops):
        if op[0] == "update":
            fa, fb = torch.from_numpy(op[1].copy()).cuda(), torch.from_numpy(op[1].copy()).cuda()
            a.accel_update(fa)
            if rng_swap.random() < share:
                _, fb = b.accel_update_swap(fb)
                nswap += 1
                if rng_swap.random() < 0.5:
                    torch.cuda.synchronize()
            else:
                b.accel_update(fb)
            assert torch.equal(fa, fb), (seed, step, float((fa - fb).abs().max()))
        elif op[0] == "relax":
            a.relax(); b.relax()
        elif op[0] == "restart":
            a.restart(); b.restart()
        elif op[0] == "set_vec_tol":
            a.set_vec_tol(op[1]); b.set_vec_tol(op[1])
        else:
            a, b = a.copy(), b.copy()
        assert a.state_digest() == b.state_digest(), (seed, step, op[0])
    print(f"seed {seed}: n={n} m={m} flavor {flavor}, {len(ops)} operations, {nswap} updates out of place (share {share}): every output and every "
          f"state digest bit-identical to the in-place run", flush=True)
