"""Stand-in for bench.py's rank side in the CPU tests of its launcher (tests/test_bench_logic_cpu.py): started by
torch.distributed.run like the real thing, no GPU, no torch import.  --mode ok: rank 0 prints a JSON line; hang: every
rank sleeps (the parent's watchdog must kill the whole group; each rank leaves its pid in --pidfile.<rank>);
fail-unless-staged: exit 7 unless `--allreduce staged` was appended by the parent's second attempt; noretry: fail and
say that a retry is pointless; fail-other-unless-staged: a failure that does not name the communication path (the parent
must NOT make its second attempt)."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="ok")
ap.add_argument("--pidfile", default="")
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--backend", default="nccl")
ap.add_argument("--allreduce", default="rccl")
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == a.gpus, (world, a.gpus)
if a.pidfile:
    with open(f"{a.pidfile}.{rank}", "w") as fh:
        fh.write(str(os.getpid()))
if a.mode == "hang" or (a.mode == "hang-unless-staged" and a.allreduce != "staged"):
    time.sleep(3600)
if a.mode == "fail-unless-staged" and a.allreduce != "staged":
    print(f"rank {rank}: ncclCommInitRank: unhandled system error (pretend)", file=sys.stderr)
    sys.exit(7)
if a.mode == "fail-other-unless-staged" and a.allreduce != "staged":
    print(f"rank {rank}: MemoryError: the host could not hold the inputs (pretend)", file=sys.stderr)
    sys.exit(9)
if a.mode == "noretry":
    print(f"rank {rank}: only 1 GPU(s) visible (NKA_BENCH_NO_RETRY)", file=sys.stderr)
    sys.exit(5)
if rank == 0:
    print("some chatter that is not the result")
    print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": world,
                      "config": {"parallelism": f"all-reduce={a.allreduce}", "backend": a.backend}}), flush=True)
