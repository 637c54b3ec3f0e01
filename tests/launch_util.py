"""Launching rank groups from tests: `python -m torch.distributed.run ... --master-port P ...`.  The port is found by binding
port 0 and letting go of it again -- between that and the launcher's own bind another process of the box can take it
(seen once on a GPU box: EADDRINUSE in the static rendezvous).  run_ranks repeats such a launch with a fresh port; any
other failure is returned as it is."""
import socket
import subprocess


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_ranks(cmd, attempts=3, **kw):
    """subprocess.run(cmd, **kw) for a command line holding `--master-port <P>`; a launch that died on an occupied port is
    repeated (LOUDLY) with another one."""
    cmd = list(cmd)
    for attempt in range(attempts):
        p = subprocess.run(cmd, **kw)
        words = (p.stderr or "") if isinstance(p.stderr, str) else ""
        taken = p.returncode != 0 and ("EADDRINUSE" in words or "address already in use" in words.lower())
        if not taken or attempt + 1 == attempts or "--master-port" not in cmd:
            return p
        i = cmd.index("--master-port") + 1
        new = str(free_port())
        print(f"LAUNCH REPEATED: port {cmd[i]} was taken between the probe and the launcher's bind; now {new}", flush=True)
        cmd[i] = new
    return p
