"""BASELINE configs[3] -- n = 1e8, m = 20 sharded by contiguous slices over the GPUs of one node, ONE RCCL all-reduce of
2 + 2 mvec doubles per update -- as a parity test (the reference's parallel contract:
/root/reference/src-F08/nka_type.F90:58-64), and its rehearsal on the one-GPU boxes.

The pool's test boxes have one GPU, so the first test SKIPS there; on a box with N >= 2 GPUs it runs N ranks, one per GPU,
at the full size against the tiled oracle.  The second test runs the SAME worker today with ranks sharing the one GPU
(all-reduce staged through the host over gloo, n_global reduced): slicing, the collective set-up, decisions, digests and
the global error norm are executed for real; only the transport differs."""
import os
import socket
import subprocess

from launch_util import run_ranks  # noqa: E402
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_sharded_ngpu_worker.py")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(world, env_extra, timeout, worker=WORKER):
    # child processes only: this process never hands its GPU context to another program
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), worker]
    p = run_ranks(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    assert p.stdout.count(" OK") == world, p.stdout[-3000:]
    return p.stdout


def test_n_gpu_rccl_sharded_run_against_the_tiled_oracle():
    """configs[3] itself: N = every GPU of the box (8 on the scaling node), n_global = 1024 * 97 656 = 99 999 744, m = 20,
    both storage flavours, the library's own RCCL communicator (no fallback: the ladder is `rccl` only, so a broken
    communicator FAILS this test instead of quietly staging through the host).  Per call and per rank: decisions equal the
    oracle's on the 97 656-element problem, identical state digests, comm_info() == (N, rank), global error against the
    tiled oracle <= 1e-10 under the truth rule.  (torch.cuda.device_count() does not initialise the GPU in this process.)"""
    import torch
    ngpu = min(torch.cuda.device_count(), 8)          # (one node: BASELINE configs[3] names 8)
    if ngpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device); rehearsed below with ranks sharing the GPU")
    out = _run(ngpu, {"NKA_NGPU_MODE": "rccl", "NKA_NGPU_N0": "97656", "NKA_NGPU_R": "1024", "NKA_NGPU_MVEC": "20"}, 800)
    assert out.count("hook=rccl") == 2 and f"comm=({ngpu}, 0)" in out, out[-2000:]


@pytest.mark.parametrize("world,r_tile", [(4, 64), (3, 256)])
def test_sharded_worker_rehearsed_with_ranks_sharing_the_gpu(world, r_tile):
    """The same worker, today: `world` ranks on cuda:0 -- four is what the box's process guard leaves (at most six processes
    may have the card open: this test process, the torch.distributed.run launcher, four ranks; five ranks were killed by
    the guard) -- the sums staged through the host over gloo, the 97 656-element problem tiled 64 resp. 256 times
    (n_global = 6.25e6 / 2.5e7, m = 20)."""
    out = _run(world, {"NKA_NGPU_MODE": "share", "NKA_NGPU_N0": "97656", "NKA_NGPU_R": str(r_tile), "NKA_NGPU_MVEC": "20"}, 900)
    assert out.count("hook=staged") == 2, out[-2000:]


INPROC_WORKER = os.path.join(ROOT, "tests", "_configs3_inproc_worker.py")


@pytest.mark.parametrize("transport,sums", [("hook", "auto"), ("p2p", "auto"), ("p2p", "blocked"), ("hook", "blocked")])
def test_configs3_partition_8_ranks_one_process(transport, sums):
    """configs[3] on its own workload, on the one GPU of this pool: n_global = 1024 * 97 656 =
    99 999 744, m = 20 in EIGHT slices -- eight handles of libnka_hip.so in one child process (8 x 4.2 GB of slots), each on
    its own stream and driven by its own host thread.  `hook`: nka_hip_set_allreduce with an in-process hook that adds the
    eight rows in rank order; `p2p`: the peer-to-peer mailboxes with n = 8, attached in-process
    (nka_hip_p2p_attach_local; the scalar step of each slice waits on the device for the other seven, so the child gets
    one hardware queue per stream: GPU_MAX_HW_QUEUES).  `sums`: the default (the norm first, then the rows on the rounded w1': two
    exchanges per update) and NKA_HIP_SUMS_BLOCKED (one exchange; with the mailboxes fused into the final sums).  Per call: decisions of all eight = the oracle's on the 97 656-
    element problem, eight identical state digests, global error against the tiled oracle under the truth rule at 1e-10;
    both storage flavours; before that an all-reduce whose result depends on the order of the additions.  What this does
    NOT exercise: a transport between GPUs (RCCL over N > 1 ranks, the mailboxes over xGMI) -- the test above."""
    report = os.path.join(ROOT, "gpurun_out", f"configs3_inproc_{transport}_{sums}.json")
    env = dict(os.environ, OMP_NUM_THREADS="1", GPU_MAX_HW_QUEUES="16", NKA_C3_TRANSPORT=transport, NKA_C3_SUMS=sums, NKA_C3_WORLD="8",
               NKA_C3_N0="97656", NKA_C3_R="1024", NKA_C3_MVEC="20", NKA_C3_REPORT=report)
    p = subprocess.run([sys.executable, INPROC_WORKER], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    assert "configs3 in one process OK" in p.stdout and p.stdout.count("worst rel err vs tiled oracle") == 2, p.stdout[-3000:]
    print(p.stdout[-1500:])


def test_in_process_mailboxes_are_refused_when_the_streams_cannot_own_hardware_queues():
    """Eight slices on one device with the HIP runtime's default of 4 hardware queues: a device-side wait would sit in front of
    the kernel it waits for (round 6 measured exactly that: a timeout, NaNs, a failed self-test).  nka_hip_p2p_attach_local now
    refuses up front and says which variable to set."""
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env.update(OMP_NUM_THREADS="1", NKA_C3_TRANSPORT="p2p", NKA_C3_WORLD="8", NKA_C3_N0="97656", NKA_C3_R="4", NKA_C3_MVEC="5",
               NKA_C3_FLAVORS="2")
    p = subprocess.run([sys.executable, INPROC_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "GPU_MAX_HW_QUEUES >= 8" in p.stderr, (p.returncode, p.stdout[-1000:], p.stderr[-3000:])
    env["GPU_MAX_HW_QUEUES"] = "8"                      # ... and with it, the same small run goes through
    p = subprocess.run([sys.executable, INPROC_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "configs3 in one process OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


REFORDER_WORKER = os.path.join(ROOT, "tests", "_sharded_reforder_worker.py")


@pytest.mark.parametrize("world", [3, 2])
def test_sharded_reference_order_sums_return_the_single_rank_reference_bits(world):
    """with nka_hip_set_sum_order(REFERENCE_ORDER) a SHARDED accelerator continues the running sums
    from rank to rank (norm rounds first, then the rows: 2N exchanges through whatever hook is installed), so an N-rank
    run returns the bits of the single-rank compiled reference -- the last (mode x topology) cell without a bit-exact
    anchor.  Ranks share the GPU (staged hook); every golden scenario in all three flavours, and n = 100 003, m = 20
    against oracle/_ref/libnka_ref_f08.so, compared with np.array_equal."""
    out = _run(world, {"NKA_NGPU_MODE": "share"}, 900, worker=REFORDER_WORKER)
    assert "every bit equal" in out, out[-2000:]


def test_sharded_reference_order_sums_with_the_whole_device_form_of_the_longest_slices():
    """The same with every rank taking its slice through k_chain_blocks / k_chain_predict / k_chain_apply (automatic from 2^19
    elements per slice on; forced here wherever a slice has a full block of 1 024): each rank's sums START from the prefix
    the ranks before it left in red[], exactly as in the one-compute-unit form."""
    out = _run(3, {"NKA_NGPU_MODE": "share", "NKA_TEST_CHAIN_MANY": "1"}, 900, worker=REFORDER_WORKER)
    assert "every bit equal" in out, out[-2000:]


def test_sharded_reference_order_sums_over_rccl():
    """The same through the library's own RCCL communicator, one GPU per rank (skips on the one-GPU boxes)."""
    import torch
    ngpu = min(torch.cuda.device_count(), 4)
    if ngpu < 2:
        pytest.skip("needs >= 2 GPUs")
    out = _run(ngpu, {"NKA_NGPU_MODE": "rccl"}, 1200, worker=REFORDER_WORKER)
    assert "every bit equal" in out, out[-2000:]
