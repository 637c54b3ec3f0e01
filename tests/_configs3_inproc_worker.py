"""Worker of tests/test_sharded_ngpu.py::test_configs3_partition_8_ranks_one_process: BASELINE configs[3] -- n_global = 1e8,
m = 20, EIGHT contiguous slices (the reference's parallel contract, /root/reference/src-F08/nka_type.F90:58-64) -- at full
size on ONE GPU: eight handles of the product library in one process, one per slice (nka_amd.dist.slice_bounds), each on its
own stream and driven by its own host thread (ctypes releases the GIL inside the library), exactly as eight ranks would
drive them.  Nothing is re-exec'ed and one process holds the card.

The exchange of the 2 + 2 mvec sums (NKA_C3_TRANSPORT):
  hook   nka_hip_set_allreduce with an in-process hook: every thread parks its row, the threads meet at a barrier, and
         every one adds the eight rows IN RANK ORDER (0, 1, ... 7) into its own buffer, on its own stream;
  p2p    the peer-to-peer mailboxes (struct P2P with n = 8) attached in-process: nka_hip_p2p_export on every handle,
         nka_hip_p2p_attach_local with the eight mailbox addresses.  NKA_C3_SUMS=blocked (the single-pass fast mode): the
         final sums of slice p go straight into every slice's mailbox and the scalar step of slice q waits ON THE DEVICE for
         the eight rows and adds them in rank order; default sums: the same mailboxes through the send-and-gather kernel,
         twice per update.  Needs one hardware queue per stream: GPU_MAX_HW_QUEUES is raised by the test.

Inputs and truth as tests/_sharded_ngpu_worker.py (the tiled-oracle construction of tests/test_hip_fullsize.py): the global
vector is F = tile(x, R), R = 4^k, so the oracle runs the n0-element problem.  After EVERY call: num_vec and the list order
of all eight handles equal the oracle's; the eight nka_hip_state_digest values are one value; the GLOBAL error
||F_out - tile(f_oracle)|| / ||F_in|| (sums of squares of the eight slices) passes the truth rule of tests/parity_util.py at
base 1e-10.  Before the first call the transport is proven on an all-reduce whose result depends on the ORDER of the
additions.  Each slice's whole-update device time (HIP events inside the library) and the host's time for one global update
are written to $NKA_C3_REPORT."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nka_amd  # noqa: E402
from nka_amd import dist as nd  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
import parity_util as P  # noqa: E402
from _sharded_ngpu_worker import small_inputs  # noqa: E402


class RankOrderedHook:
    """The all-reduce of eight slices living in one process: rows[parity][rank] <- mine; barrier; mine <- rows[0] + rows[1]
    + ... in rank order.  Two row sets: a thread can reach exchange x + 2 only after every thread has passed the barrier of
    x + 1, which each does after synchronising the stream that read the rows of x."""

    def __init__(self, world, cap, device):
        self.world, self.cap = world, cap
        self.rows = torch.zeros(2, world, cap, dtype=torch.float64, device=device)
        self.barrier = threading.Barrier(world)
        self.parity = [0] * world
        self.device = device

    def hook_of(self, rank):
        class _Alias:
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        def hook(ptr, count, stream):
            assert count <= self.cap
            s = torch.cuda.ExternalStream(stream, device=self.device) if stream else torch.cuda.default_stream(self.device)
            par = self.parity[rank]
            self.parity[rank] = par ^ 1
            with torch.cuda.stream(s):
                mine = torch.as_tensor(_Alias(ptr, count), device=self.device)
                self.rows[par, rank, :count].copy_(mine)
            s.synchronize()
            self.barrier.wait(timeout=120)
            with torch.cuda.stream(s):
                acc = self.rows[par, 0, :count].clone()
                for r in range(1, self.world):                 # rank order
                    acc += self.rows[par, r, :count]
                mine.copy_(acc)
            s.synchronize()
        return hook


def in_threads(world, fn):
    """fn(rank) on `world` threads; the first exception is re-raised here."""
    errs = [None] * world

    def run(r):
        try:
            torch.cuda.set_device(0)
            fn(r)
        except BaseException as exc:      # noqa: BLE001
            errs[r] = exc
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r, e in enumerate(errs):
        if e is not None:
            raise RuntimeError(f"slice {r}: {e!r}") from e


def main():
    transport = os.environ.get("NKA_C3_TRANSPORT", "hook")
    world = int(os.environ.get("NKA_C3_WORLD", "8"))
    n0 = int(os.environ.get("NKA_C3_N0", "97656"))
    R = int(os.environ.get("NKA_C3_R", "1024"))
    m = int(os.environ.get("NKA_C3_MVEC", "20"))
    flavors = [int(v) for v in os.environ.get("NKA_C3_FLAVORS", "2,0").split(",")]
    sum_mode = {"auto": nka_amd.SUMS_AUTO, "blocked": nka_amd.SUMS_BLOCKED}[os.environ.get("NKA_C3_SUMS", "auto")]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n = n0 * R
    bounds = [nd.slice_bounds(n, world, r) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
    assert len({s.cuda_stream for s in streams}) == world
    idx = [torch.arange(lo, hi, device=dev, dtype=torch.int64) % n0 for lo, hi in bounds]          # F[i] = x[i mod n0]
    calls = m + 8
    X = small_inputs(n0, calls, seed=321)
    report = {"transport": transport, "sums": os.environ.get("NKA_C3_SUMS", "auto"), "world": world, "n_global": n, "mvec": m, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
              "flavors": {}}
    for flavor in flavors:
        tag = f"configs[3] in one process x{world} ({transport}, sums {os.environ.get('NKA_C3_SUMS', 'auto')}) n={n} m={m} flavor {flavor}"
        accs = [nka_amd.nka().init(hi - lo, m, flavor=flavor, device=0, stream=streams[r].cuda_stream).set_sum_order(sum_mode)
                for r, (lo, hi) in enumerate(bounds)]
        for r, a in enumerate(accs):
            a.set_shard(r, world)
            a.set_timing(calls, 1)
        if transport == "hook":
            ring = RankOrderedHook(world, 2 + 2 * m + 64, dev)
            for r, a in enumerate(accs):
                a.set_dot_prod(ring.hook_of(r))
        else:
            for a in accs:
                a.p2p_export(world)
            boxes = [a.p2p_mailbox() for a in accs]
            for r, a in enumerate(accs):
                a.p2p_attach_local(boxes, r)
        torch.cuda.synchronize()

        # the transport, proven on sums whose value depends on the ORDER of the additions: slice r contributes column r
        probe = np.array([[1e16, 1.0, -1e16, 1.0, 3.0, 1e-3, -1e16, 1e16][:world] if world <= 8 else [1.0] * world,
                          [(-1.0) ** r * 10.0 ** (r % 5) / 3.0 for r in range(world)],
                          [float(r + 1) for r in range(world)]])
        want = np.zeros(3)
        for k in range(3):
            acc = probe[k, 0]
            for r in range(1, world):
                acc = acc + probe[k, r]
            want[k] = acc
        got = [None] * world

        def self_test(r):
            with torch.cuda.stream(streams[r]):
                t = torch.from_numpy(probe[:, r].copy()).to(dev)
                accs[r].allreduce_now(t)
                streams[r].synchronize()
                got[r] = t.cpu().numpy()
        in_threads(world, self_test)
        for r in range(world):
            assert np.array_equal(got[r], want), (tag, "rank-ordered all-reduce", r, got[r], want)

        ora = O.OracleNKA(n0, m, flavor)
        spread = P.Spread(O, n0, m)
        worst, walls, dev_ms = 0.0, [], []
        start = threading.Barrier(world)
        for t, x in enumerate(X):
            f = x.copy()
            ora.accel_update(f)
            spread.update(x)
            xd = torch.from_numpy(x).to(dev)
            ref = torch.from_numpy(f).to(dev)
            ex = torch.from_numpy(spread.exact).to(dev)
            torch.cuda.synchronize()
            sums = [None] * world
            t0s, t1s = [0.0] * world, [0.0] * world

            def step(r):
                s = streams[r]
                with torch.cuda.stream(s):
                    big = xd[idx[r]].contiguous()
                    s.synchronize()
                    start.wait(timeout=120)
                    t0s[r] = time.perf_counter()
                    accs[r].accel_update(big)
                    s.synchronize()
                    t1s[r] = time.perf_counter()
                    sums[r] = torch.stack([((big - ref[idx[r]]) ** 2).sum(), ((big - ex[idx[r]]) ** 2).sum(),
                                           (xd[idx[r]] ** 2).sum()]).cpu().numpy()
            in_threads(world, step)
            walls.append((max(t1s) - min(t0s)) * 1e3)
            # decisions: exact, on every slice; one digest
            for r, a in enumerate(accs):
                assert a.num_vec() == ora.num_vec(), (tag, r, t, a.num_vec(), ora.num_vec())
                assert a.state().list_order() == ora.state().list_order(), (tag, r, t)
            digs = [a.state_digest() for a in accs]
            assert all(d == digs[0] for d in digs), (tag, t, [f"{d:016x}" for d in digs])
            tot = np.sum(np.stack(sums), axis=0)
            nx = max(float(tot[2]) ** 0.5, 1e-300)
            err, err_ex = float(tot[0]) ** 0.5 / nx, float(tot[1]) ** 0.5 / nx
            P.check(err, ora.state(), tag, base=1e-10, where=t, spread=spread.value, truth=(err_ex, spread.err_ref, n, m))
            worst = max(worst, err)
            dev_ms.append([a.timing_ms(0)[3] for a in accs])
        P.finish()
        assert all(a.defined() and a.num_vec() == m for a in accs)
        steady = slice(m + 2, None)
        per_rank = np.median(np.array(dev_ms)[steady], axis=0)
        report["flavors"][str(flavor)] = {
            "worst_rel_err_vs_tiled_oracle": worst,
            "steady_whole_update_device_ms_per_slice_median": [round(float(v), 4) for v in per_rank],
            "steady_global_update_host_ms_median": round(float(np.median(walls[steady])), 4),
            "note": "eight slices SHARE one GPU: a slice's time is about the whole job's, not an eighth of it"}
        print(f"{tag}: worst rel err vs tiled oracle {worst:.2e}; steady state: slices' whole-update device ms "
              f"{per_rank.min():.3f}..{per_rank.max():.3f}, one global update (host, all eight done) "
              f"{np.median(walls[steady]):.3f} ms", flush=True)
        torch.cuda.synchronize()
        for a in accs:
            if transport == "p2p":
                a.p2p_detach()
        for a in accs:
            a.delete()
        del accs
        torch.cuda.empty_cache()
    out = os.environ.get("NKA_C3_REPORT")
    if out:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as fh:
            json.dump(report, fh, indent=1)
    print("configs3 in one process OK", flush=True)


if __name__ == "__main__":
    main()
