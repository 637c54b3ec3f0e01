#!/usr/bin/env python3
"""bench.py -- NKA accel_update throughput on MI355X (the BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--vlen 100000000] [--mvec 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for every N.  In the plain form with N > 1 this process touches no GPU: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process group (never an exec),
relays rank 0's JSON line, returns the child's exit code, and kills the whole group when a watchdog
expires (launch_ranks below).  If that attempt dies or hangs with the RCCL all-reduce, ONE second attempt
runs with the all-reduce staged through the host over gloo, and the line says so in config.parallelism.

A "step" is ONE nka accel_update on one synthetic correction vector: the hot
path of /root/reference/src-F08/nka_type.F90:249-419 in steady state (subspace
full: L = k = mvec), inputs already resident in HBM when the timed region
starts.  Workload at N=1: BASELINE.json configs[2], n=1e8, mvec=20, fp64.
With N>1 the SAME global vector is sharded by contiguous slices (strong
scaling, BASELINE configs[3]); the only exchanges are two small RCCL all-reduces
per update on the kernel stream -- the norm (1 double), then both Gram rows (1+2*mvec)
-- or ONE of 2+2*mvec doubles in the opt-in fast sum mode (--sums blocked).

Prints ONE JSON line (rank 0) with the driver's keys plus
  roofline     : PHYSICAL fractions only.  Top level = the dominant kernel (PB
                 k_combine): bytes that launch really moves (byte model confirmed by
                 the PMC passes under profiles/) / its mean duration over the timed
                 steps (HIP events on the kernel stream; every timed update at the
                 headline size, every 4th below 5e7 elements per GPU, where four
                 event records would widen a 0.8 ms update by 2 %) / 8 TB/s;
                 `whole_update` = PA + scalar step + PB the same way; `kernels`
                 per launch.  The
                 contract's B_alg = 8n(11+L+2k) (a three-pass
                 schedule that moves more bytes than this one) appears as
                 `contract_bytes_ratio`; a B_alg-based rate (`contract_GBps`,
                 `contract_frac_of_peak`) only where B_alg bounds the traffic from below:
                 under also_f08_rounding, never for the compact flavour.
  cpu_baseline : the compiled reference (oracle/_ref, kind "reference") or the
                 oracle port, serial (1 core), timed on this box's host in this run: at the
                 benchmark's own n when the host has the memory for it (22 fill calls + 3
                 timed updates at n = 1e8, m = 20: ~60 s of CPU work), else at n = 2e7 with
                 the reason in `sample`.
                 `traffic`: HBM bytes of that launch by the PMC counters, measured IN THIS RUN
                 (two rocprofv3 passes, FETCH_SIZE and WRITE_SIZE, over a short child process of
                 this script after the timed region; `traffic_source` says so, or names the
                 committed summary under profiles/ that was used instead).
                 `device_time_stats_ms`: mean / median / min / max of the per-update
                 device times; `probe_ceiling`: the repo's own streaming
                 probe (tools/hbm_probe: pure read, pure write, PB's read/write mixes)
                 run as a child process after the timed region of the SAME run.
  value        : the flavour the drop-in front ends run BY DEFAULT (`call a%init(vlen,
                 mvec)` of the Fortran module, nka_init of the F95 one, nka().init in
                 Python: NKA_HIP_FLAVOR_DEFAULT = compact storage, the src-C statement of
                 the combine; include/nka_hip.h).  `config.flavor_is_front_end_default`
                 says so; --flavor / NKA_HIP_FLAVOR select another one.
  also_f08_rounding : the same workload and protocol measured a second time in
                 the same run with the src-F08 statement bit for bit (flavor=F08: two
                 stored vectors per pair).  Both are checked against the compiled
                 src-F08 reference in tests/ (decisions exact, values within the stated
                 tolerance).
  with_drops (N = 1, headline run only): the same accelerator shape with a SHRUNK subspace -- every input in a 12-dimensional
                 span, a dependence drop and a device synchronisation per update (--workload drops): updates/s, launch
                 widths, physical roofline at the actual list length with PMC traffic.
  out_of_place_entry (N = 1, headline run only): the headline workload through the opt-in nka_hip_accel_update_swap
                 (two store streams less in PB; bit-identical results).
  sum_mode_blocked (N = 1, headline run only): the headline workload in the OTHER fast sum mode.  Since round 6 `value` is
                 quoted in the DEFAULT sums (the norm first, then PA on the rounded w1' = the Gram row as the reference
                 defines it: re-decided from 23 000 paired soak records, profiles/r06/soak_paired.txt); this entry is the
                 opt-in single-pass mode NKA_HIP_SUMS_BLOCKED (raw-sum Gram row: 2 words per element and one exchange
                 less).  With --sums blocked the roles are swapped (sum_mode_rounded).
  config2_n1e7_m10 (N = 1, headline run only): BASELINE configs[1] (n = 1e7, m = 10) measured in the same
                 run on the first 1e7 elements of the resident inputs (updates/s, whole-update fraction).
  config5_abstract_vector (N = 1, headline size only): BASELINE configs[4], the
                 src-F08-vector abstract path through the Fortran vector flavour on
                 the device block vector, measured by nka_vector_driver in a child
                 process after the main measurement.
  replica_check (N > 1): digest of the replicated scalar state of every rank,
                 compared after warm-up and after the timed steps (outside the
                 timed region); a mismatch aborts the job.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 12345


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=None, help="default mvec+4 (fills the subspace)")
    # (--n clashes with torchrun's own option abbreviations when given after the script name)
    ap.add_argument("--vlen", "--n", dest="n", type=float, default=1e8, help="GLOBAL vector length")
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the abstract-vector (BASELINE configs[4]) extra")
    ap.add_argument("--cpu-n", type=float, default=None,
                    help="vector length of the CPU baseline; default: the benchmark's own n when the host has >= 48 GB "
                         "available (BASELINE.md section 4), else 2e7")
    ap.add_argument("--flavor", choices=["default", "c", "f08", "f08vec"], default=os.environ.get("NKA_BENCH_FLAVOR", "default"),
                    help="'default' = what `call a%%init(vlen, mvec)` of the drop-in Fortran module runs (compact storage "
                         "unless NKA_HIP_FLAVOR says otherwise); or name the reference rounding mirrored")
    ap.add_argument("--sums", choices=["default", "rounded", "blocked"], default=os.environ.get("NKA_BENCH_SUMS", "default"),
                    help="how the fast passes form the Gram row (nka_hip_set_sum_order).  default = what every front end runs "
                         "since round 6: the norm first, then PA on the ROUNDED w1' (NKA_HIP_SUMS_BLOCKED_ROUNDED: 2 words per "
                         "element and one exchange more); blocked = the opt-in single-pass fast mode (raw-sum Gram row)")
    ap.add_argument("--allreduce", choices=["p2p", "rccl", "torch", "staged"], default=os.environ.get("NKA_BENCH_ALLREDUCE", "rccl"),
                    help="the transport of the exchanges of a sharded update.  rccl (default): the library's RCCL communicator on the kernel "
                         "stream; p2p: the opt-in peer-to-peer exchange (mailboxes mapped through hipIpc, no communication "
                         "kernel; falls through to rccl where IPC is refused); torch / staged: fallbacks")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("NKA_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend.  gloo (+ --allreduce staged, NKA_BENCH_SHARE_GPU=1) is a REHEARSAL of the "
                         "multi-rank logic with all ranks on one GPU (RCCL refuses that); its numbers mean nothing")
    ap.add_argument("--workload", choices=["full", "drops"], default="full",
                    help="full: independent inputs, subspace full (L = k = mvec), the BASELINE workload.  drops: every "
                         "input lies in a --drop-dim dimensional span, so every update takes a dependence drop "
                         "(F08:326-345) and the subspace holds k = drop-dim < mvec vectors; the caller synchronises "
                         "after every update, like a solver that reads its residual norm")
    ap.add_argument("--drop-dim", type=int, default=12)
    ap.add_argument("--launch-timeout", type=int, default=int(os.environ.get("NKA_BENCH_LAUNCH_TIMEOUT_S", "280")),
                    help="plain form, N > 1: seconds the parent waits for the ranks before it kills their process group")
    ap.add_argument("--no-fallback", action="store_true",
                    help="plain form, N > 1: no second attempt with the host-staged all-reduce after a failed one")
    return ap.parse_args(argv)


def host_mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for ln in fh:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 1048576.0
    except OSError:
        pass
    return None


def cpu_baseline(mvec: int, n: int, timed: int = 6, gen=None, note: str = "", after_update=None):
    """Time the reference's own accel_update (compiled from /root/reference into
    oracle/_ref, if it travelled with the repo) or else the oracle port, serial,
    on the same workload: same generator (gen(t) -> the t-th input as a host array;
    default: the numpy twin of the device generator), same mvec, steady state.
    after_update(t, f, kind), if given, sees every output (the device's reference-order
    twin of the run compares itself with it there, bit for bit)."""
    import numpy as np
    from nka_amd import synth
    from oracle import oracle_py as O
    if gen is None:
        def gen(t):
            return synth.fill_numpy(SEED, t, 0, n, n)
    kind = "port"
    acc = None
    if O.have_ref():
        try:
            acc = O.RefF08(n, mvec)
            kind = "reference"
        except Exception:
            acc = None
    if acc is None:
        acc = O.OracleNKA(n, mvec)
    t_all = time.perf_counter()
    t_cpu = 0.0
    for t in range(mvec + 2):
        f = gen(t)
        t0 = time.perf_counter()
        acc.accel_update(f)
        t_cpu += time.perf_counter() - t0
        if after_update is not None:
            after_update(t, f, kind)
    assert acc.num_vec() == mvec
    dt = []
    for t in range(mvec + 2, mvec + 2 + timed):
        f = gen(t)
        t0 = time.perf_counter()
        acc.accel_update(f)
        dt.append(time.perf_counter() - t0)
        if after_update is not None:
            after_update(t, f, kind)
    per = float(np.median(dt))
    return {
        "value": 1.0 / per, "unit": "updates/s", "cores": 1, "kind": kind,
        "sample": f"n={n:d}, mvec={mvec}, {timed} steady-state updates after {mvec + 2} fill calls "
                  f"({t_cpu + sum(dt):.1f} s of CPU work in accel_update, {time.perf_counter() - t_all:.1f} s with the "
                  f"inputs); serial like the reference" + note,
        "n": n, "s_per_update": per, "host_mem_available_gb": host_mem_available_gb(),
        "algorithmic_GBps": 8.0 * n * (11 + 3 * mvec) / per / 1e9,
    }


def config5_abstract_vector(steps: int = 20):
    """BASELINE configs[4]: the src-F08-vector abstract path (user dot/axpy hooks on a
    4-field block vector, 4 x 1e7 per field, m = 20) through the Fortran vector
    flavour and its device block vector (nka_amd/fortran/build/nka_vector_driver),
    run as a child process after the main measurement has released its memory.
    Reported as an extra key; never part of `value`."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "nka_amd", "fortran", "build", "nka_vector_driver")
    if not os.path.exists(exe):
        return {"value": None, "error": "nka_vector_driver not built"}
    out = {"workload": "BASELINE configs[4]: abstract vector hooks, 4 fields x 1e7, mvec=20, fp64, 1 GPU",
           "unit": "updates/s", "steps": steps}
    n, m = 4 * 10**7, 20
    for key, compact, words in (("reference_rounding", "0", 8 + 3 * m), ("compact_option", "1", 9 + 2 * m)):
        try:
            p = subprocess.run([exe, "bench", "4", "10000000", str(m), str(steps), compact], capture_output=True,
                               text=True, timeout=600)
            mt = re.search(r"updates/s\s+([0-9.]+)\s+ms/update\s+([0-9.]+)", p.stdout)
            if p.returncode != 0 or not mt:
                out[key] = {"value": None, "error": (p.stdout + p.stderr)[-300:]}
                continue
            ups, ms = float(mt.group(1)), float(mt.group(2))
            moved = 8.0 * n * words
            out[key] = {"value": ups, "ms_per_step": ms, "bytes_moved_per_update": moved,
                        "achieved_GBps": moved / (ms * 1e-3) / 1e9, "frac": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "byte_model": ("8n(8+3m)" if compact == "0" else "8n(9+2m)")
                                      + ": update_norm2_dots 2+m (ONE pure-read pass for the norm and both inner-product "
                                      "rows; the normalisation of the new pair deferred to the combine), "
                                      "update/axpy_many_keep " + ("6+2m" if compact == "0" else "7+m")}
            pmv = pmc_traffic_vector(compact, n, m)       # counter evidence: committed PMC summary of the same workload
            if pmv:
                out[key]["traffic"] = pmv[0]["hbm_bytes_per_update"]
                out[key]["traffic_over_bytes_moved"] = pmv[0]["hbm_bytes_per_update"] / moved
                out[key]["traffic_per_kernel_over_model"] = {k: round(v["traffic_over_model"], 4) for k, v in pmv[0]["kernels"].items()}
                out[key]["traffic_source"] = "committed file, not this run: " + pmv[1]
            if compact == "0":      # B_alg bounds the traffic of the reference rounding only (ratio 1.04): the BASELINE.md-style rate
                out[key]["contract_bytes_ratio"] = 8.0 * n * (11 + 3 * m) / moved
                out[key]["contract_GBps"] = 8.0 * n * (11 + 3 * m) / (ms * 1e-3) / 1e9
                out[key]["contract_frac_of_peak"] = out[key]["contract_GBps"] / HBM_PEAK_GBPS
        except Exception as exc:   # an extra, never the measured path
            out[key] = {"value": None, "error": repr(exc)}
    return out


def probe_ceilings(n: int = 10**8, timeout: int = 240, device: int | None = None):
    """The repo's own streaming probe (tools/hbm_probe mode `cr`: no arithmetic, random
    data, one block per CU) run as a CHILD process after the timed region, on the same
    box in the same run: what this memory system gives a pure read of 22 streams (PA's
    shape), a pure write, and the 22-read / 5-write and 42-read / 5-write mixes of PB
    with tile tickets.  These -- not the 8 TB/s spec figure -- are the ceilings the
    kernels' `achieved` figures can be read against."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "hbm_probe")
    if not os.path.exists(exe):
        return {"error": "tools/hbm_probe not built"}
    env = dict(os.environ)
    if device is not None:        # the child sees only this rank's GPU (N > 1 lines)
        vis = env.get("HIP_VISIBLE_DEVICES")
        env["HIP_VISIBLE_DEVICES"] = str(device) if not vis else vis.split(",")[device]
    try:
        p = subprocess.run([exe, str(n), "0", "cr"], capture_output=True, text=True, timeout=timeout, env=env)
    except Exception as exc:      # an extra, never the measured path
        return {"error": repr(exc)}
    if p.returncode != 0:
        return {"error": (p.stdout + p.stderr)[-300:]}
    names = {"window S=22 W=0": "pure_read_22_streams_GBps", "S=0 W=4": "pure_write_4_streams_GBps",
             "tickets S=22 W=5": "mix_22R_5W_GBps", "tickets S=42 W=5": "mix_42R_5W_GBps"}
    out = {}
    for line in p.stdout.splitlines():
        mt = re.search(r"([0-9.]+) GB/s", line)
        if not mt:
            continue
        for prefix, key in names.items():
            if line.startswith(prefix):
                out[key] = max(out.get(key, 0.0), float(mt.group(1)))
    out["source"] = f"tools/hbm_probe {n} 0 cr (child process of this run, after the timed region; best of 2)"
    return out


FLAVOR_TEXT = {"c": "src-C rounding f += c*(v-w), compact storage (v slot keeps v'-w'); held to the src-F08 reference",
               "f08": "src-F08 rounding (f - c*w) + c*v, two stored vectors per pair",
               "f08vec": "src-F08-vector rounding ((-c)*w + c*v) + f, normalise by reciprocal"}


def words_moved(flavor: str, L: int, k: int, out_of_place: bool = False, norm_pass: bool = False):
    """8-byte words per element each launch of the two-pass schedule really moves
    (confirmed by the PMC passes under profiles/): PA reads w1, f and the L stored
    w; PB reads f and the k pairs (one vector per pair with compact storage, the
    pending pair always as two) and writes w1', v1', w_new, v_new, f -- out of place
    (nka_hip_accel_update_swap) w1', v1', v_new only.  norm_pass (the default sums
    since round 6): the norm of d = w1 - f in a short pass of its own before PA
    (k_norm_diff: w1 and f once more) -- counted with the PA phase, whose events span it."""
    # lists longer than 32 take the multi-pass kernels (k_dots / k_combine in passes of 32): every further pass of PA reads
    # f and w1 again, every further pass of PB reads the running f and stores it once more
    npa, npb = max(1, -(-L // 32)), max(1, -(-k // 32))
    pb_reads = (1 + k) if flavor == "c" else (2 * k)
    return {"PA_k_dots": L + 2 * npa + (2 if norm_pass else 0), "PB_k_combine": pb_reads + npb + (2 if out_of_place else 4) + npb}


def pmc_traffic(flavor: str, n_local: int, m: int, norm_pass: bool = False):
    """HBM bytes per launch from the newest rocprofv3 PMC summary of THIS workload
    (tools/rocprof_bench.sh + tools/pmc_summary.py, committed under profiles/)."""
    import glob
    cands = glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{flavor}.json")) + \
        glob.glob(os.path.join(ROOT, "profiles", "r*", "*", f"pmc_traffic_{flavor}.json"))      # (shard sizes: profiles/rNN/shard_<n>/)
    for cand in sorted(cands, reverse=True):
        try:
            with open(cand) as fh:
                pm = json.load(fh)
            if pm.get("n") == n_local and pm.get("mvec") == m and pm.get("hbm_bytes_per_update") and \
                    ("k_norm_diff" in pm.get("kernels", {})) == norm_pass:         # (a summary of the same sum mode)
                return pm, os.path.relpath(cand, ROOT)
        except Exception:
            pass
    return None, None


def pmc_traffic_vector(compact: str, n: int, m: int):
    """HBM bytes per update of the abstract-vector kernels from the newest committed PMC summary of config 5
    (tools/rocprof_vector.sh + tools/pmc_vector_summary.py -> profiles/rNN/pmc_traffic_vector_compact<c>.json)."""
    import glob
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_vector_compact{compact}.json")), reverse=True):
        try:
            with open(cand) as fh:
                pm = json.load(fh)
            if pm.get("n") == n and pm.get("mvec") == m and pm.get("hbm_bytes_per_update"):
                return pm, os.path.relpath(cand, ROOT)
        except Exception:
            pass
    return None


def pmc_same_run(flavor: str, n_local: int, m: int, timeout: int = 170, extra_args=()):
    """HBM bytes per launch measured IN THIS RUN: two rocprofv3 counter passes (FETCH_SIZE, then WRITE_SIZE: separate
    passes with --kernel-trace only, units and the gfx950 FETCH_SIZE x 2 correction as MI355X_MICROARCH.md prescribes,
    tools/pmc_summary.py) over a short CHILD process of this very script -- same box, same build, same workload, a few
    steady-state updates -- after the timed region.  Returns the dict of tools/pmc_summary.py or None (no rocprofv3,
    a failed pass, a timeout: the committed summary under profiles/ is then used and labelled as such)."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or os.environ.get("NKA_BENCH_PMC", "1") == "0":
        return None
    try:
        spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
        ps = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ps)
        out = tempfile.mkdtemp(prefix="nka_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp", NKA_BENCH_SECONDARY="0", NKA_BENCH_PMC="0")
        for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(out, sub), "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--flavor", flavor, "--steps", "4",
                   "--vlen", str(n_local), "--mvec", str(m)] + list(extra_args)
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            if p.returncode != 0:
                return None
        fetch = ps.load_counter(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
        write = ps.load_counter(os.path.join(out, "pmc_write"), "WRITE_SIZE")
        res = {"n": n_local, "mvec": m, "kernels": {}}
        total = 0.0
        def plain(agg, stem):           # (not a template: the norm pass of the default sums)
            ks = [k for k in agg if k.startswith("nka::" + stem)]
            return max(ks, key=lambda k: len(agg[k])) if ks else None
        for stem in ("k_norm_diff", "k_dots", "k_combine"):
            kf, kw = (plain(fetch, stem), plain(write, stem)) if stem == "k_norm_diff" else (ps.widest(fetch, stem), ps.widest(write, stem))
            if stem == "k_norm_diff" and (not kf or not kw):
                continue                  # (--sums blocked: there is no norm pass)
            if not kf or not kw:
                return None
            fv = sorted(fetch[kf])[-max(1, len(fetch[kf]) // 2):]      # steady state: the launches with the most traffic
            wv = sorted(write[kw])[-max(1, len(write[kw]) // 2):]
            rb, wb = sum(fv) / len(fv) * 1024 * 2, sum(wv) / len(wv) * 1024
            res["kernels"][stem] = {"kernel": kf, "read_bytes": rb, "write_bytes": wb, "launches_averaged": len(fv),
                                    "words_per_element": (rb + wb) / 8.0 / max(n_local, 1)}
            total += rb + wb
        res["hbm_bytes_per_update"] = total
        shutil.rmtree(out, ignore_errors=True)
        return res
    except Exception:      # an extra, never the measured path
        return None


def roofline_block(flavor: str, n_local: int, m: int, mean, probe=None, stats=None, pm_live=None, L=None, k=None,
                   out_of_place=False, norm_pass=False):
    """`roofline` object of the JSON line.  Every `achieved`/`frac` in it is
    PHYSICAL: bytes the launch really moves (byte model above, confirmed by the
    PMC counters) / mean launch duration (HIP events on the kernel stream during
    the timed steps) / 8 TB/s, hence <= 1.  Top level = the dominant kernel (PB
    k_combine); `whole_update` = both passes and the scalar step together.  The
    contract's figure B_alg = 8n(11+L+2k) of SURVEY.md 8(d) assumes a three-pass
    schedule that moves MORE bytes than this one; its ratio to the bytes moved
    is `contract_bytes_ratio`, and `contract_GBps` = B_alg / time is a rate of
    useful work, not a bandwidth (it may exceed the peak and is never `frac`)."""
    L = m if L is None else L          # stored vectors PA reads / pairs PB combines: mvec with the subspace full,
    k = m if k is None else k          # fewer after dependence drops (--workload drops)
    words = words_moved(flavor, L, k, out_of_place, norm_pass)
    ms = {"PA_k_dots": mean[0], "PB_k_combine": mean[2]}      # (PA phase: k_norm_diff + k_dots with the default sums)
    if pm_live:
        pm, pm_src = pm_live, ("same run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over a child process of "
                               "this script after the timed region; FETCH_SIZE x 2 on gfx950")
    else:
        pm, pm_src = pmc_traffic(flavor, n_local, m, norm_pass) if (L, k) == (m, m) and not out_of_place else (None, None)
    pmk = {"PA_k_dots": "k_dots", "PB_k_combine": "k_combine"}
    kernels = {}
    for name, w in words.items():
        t = ms[name] * 1e-3
        b = 8.0 * n_local * w
        tr = None
        if pm and pmk[name] in pm.get("kernels", {}):
            kk = pm["kernels"][pmk[name]]
            tr = kk["read_bytes"] + kk["write_bytes"]
            if name == "PA_k_dots" and norm_pass and "k_norm_diff" in pm["kernels"]:
                tr += pm["kernels"]["k_norm_diff"]["read_bytes"] + pm["kernels"]["k_norm_diff"]["write_bytes"]
        kernels[name] = {"bytes_moved": b, "words_per_element": w, "mean_ms": ms[name],
                         "achieved": (b / t / 1e9) if t > 0 else None,
                         "frac": (b / t / 1e9 / HBM_PEAK_GBPS) if t > 0 else None,
                         "traffic": tr}
    kernels["k_solve"] = {"mean_ms": mean[1]}
    dom = max(("PA_k_dots", "PB_k_combine"), key=lambda nm: ms[nm])
    upd_s = mean[3] * 1e-3
    moved = 8.0 * n_local * sum(words.values())
    b_alg = 8.0 * n_local * (11 + L + 2 * k)
    out = {
        "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
        "kernel": dom, "achieved": kernels[dom]["achieved"], "frac": kernels[dom]["frac"],
        "traffic": kernels[dom]["traffic"], "traffic_source": pm_src,
        "what": "dominant kernel: bytes the launch moves (8*n_local*words_per_element; PMC-confirmed) / mean launch "
                "duration over the timed steps (HIP events on the kernel stream) / 8 TB/s",
        "whole_update": {"bytes_moved": moved, "mean_ms": mean[3],
                         "achieved": (moved / upd_s / 1e9) if upd_s > 0 else None,
                         "frac": (moved / upd_s / 1e9 / HBM_PEAK_GBPS) if upd_s > 0 else None,
                         "traffic": pm["hbm_bytes_per_update"] if pm else None,
                         "what": "PA + scalar step + PB, first kernel start .. last kernel end"},
        "kernels": kernels,
        "sum_mode": "default: norm pass (k_norm_diff) + PA on the rounded w1' (NKA_HIP_SUMS_BLOCKED_ROUNDED)" if norm_pass
                    else "NKA_HIP_SUMS_BLOCKED: one pure-read pass, raw-sum Gram row",
        "bytes_moved_per_update": moved,
        "probe_ceiling": probe,
        "device_time_stats_ms": stats,
        # the same figures once more as FLAT scalars: a record that keeps only the scalar entries of this object (the
        # driver's `parsed`) still shows both passes, the whole update and the same-run ceiling of the dominant mix
        "whole_update_frac": (moved / upd_s / 1e9 / HBM_PEAK_GBPS) if upd_s > 0 else None,
        "whole_update_ms": mean[3],
        "PA_k_dots_frac": kernels["PA_k_dots"]["frac"], "PA_k_dots_ms": ms["PA_k_dots"],
        "PB_k_combine_frac": kernels["PB_k_combine"]["frac"], "PB_k_combine_ms": ms["PB_k_combine"],
        "k_solve_ms": mean[1],
    }
    if probe and not probe.get("error"):
        key = "mix_22R_5W_GBps" if flavor == "c" else "mix_42R_5W_GBps"
        out["probe_ceiling_dominant_mix_GBps"] = probe.get(key)
        out["probe_ceiling_pure_read_GBps"] = probe.get("pure_read_22_streams_GBps")
    if flavor != "c":
        # SURVEY 8(d)'s B_alg is a lower bound on the traffic only for the flavours that stream BOTH stored vectors of
        # a pair (ratio 1.04): there B_alg / time is the BASELINE.md-style fraction.  Compact storage moves FEWER bytes
        # than B_alg (ratio 1.45): NO contract figure of any kind is printed for it (B_alg / time would read as > 100 % of
        # the peak); the contract-comparable number of a compact line is roofline.reference_rounding, measured in the same run.
        out["contract_bytes_per_update"] = b_alg
        out["contract_bytes_ratio"] = b_alg / moved
        out["contract_GBps"] = (b_alg / upd_s / 1e9) if upd_s > 0 else None
        out["contract_frac_of_peak"] = (b_alg / upd_s / 1e9 / HBM_PEAK_GBPS) if upd_s > 0 else None
    return out


def reference_rounding_entry(also):
    """The src-F08-rounding measurement of the same run as the entry `roofline.reference_rounding` (+ flat scalars) of a
    compact-flavour line: the only figures of the line that can be held against SURVEY.md 8(d)'s B_alg and BASELINE.md."""
    r = also["roofline"]
    nested = {"what": "the same workload and protocol in the same run with the src-F08 statement bit for bit (flavor F08: two "
                      "stored vectors per pair): B_alg = 8n(11+L+2k) bounds ITS traffic from below (ratio 1.04), so "
                      "contract_frac_of_peak = B_alg / wall time per update / 8 TB/s is the BASELINE.md-comparable fraction",
              "value": also["value"], "unit": "updates/s", "ms_per_step": also["ms_per_step"],
              "steady_state": also["steady_state"],
              "contract_GBps": r.get("contract_GBps"), "contract_frac_of_peak": r.get("contract_frac_of_peak"),
              "physical_frac": r["whole_update"]["frac"],
              "kernel_fracs": {k: v.get("frac") for k, v in r["kernels"].items() if "frac" in v},
              "bytes_moved_per_update": r["bytes_moved_per_update"], "contract_bytes_per_update": r.get("contract_bytes_per_update")}
    flat = {"reference_rounding_updates_per_s": also["value"], "reference_rounding_ms_per_step": also["ms_per_step"],
            "reference_rounding_contract_frac_of_peak": r.get("contract_frac_of_peak"),
            "reference_rounding_physical_frac": r["whole_update"]["frac"],
            "reference_rounding_PA_frac": r["kernels"]["PA_k_dots"]["frac"],
            "reference_rounding_PB_frac": r["kernels"]["PB_k_combine"]["frac"]}
    return nested, flat


# What a record that keeps only the FIRST scalar entries of `roofline` and the first ~120 characters of a string must still
# show (the driver's BENCH_rNN.parsed kept 19 scalar keys of round 5's line and lost every reference_rounding_* one): the
# contract-comparable src-F08-rounding figures and the same-run ceiling come right after the contract's own keys.
ROOFLINE_FIRST = ("bound", "unit", "peak", "kernel", "achieved", "frac", "traffic",
                  "whole_update_frac", "PA_k_dots_frac", "PB_k_combine_frac", "frac_of_probe_ceiling",
                  "reference_rounding_updates_per_s", "reference_rounding_contract_frac_of_peak", "reference_rounding_PB_frac",
                  "sum_mode_blocked_updates_per_s", "probe_ceiling_dominant_mix_GBps", "whole_update_ms", "traffic_source", "reference_rounding_physical_frac")
SHORT = 100          # characters of a string that a truncating record keeps for certain


def shorten(d: dict, key: str, short: str | None = None):
    """Keep d[key] within SHORT characters; the full text moves to d[key + "_detail"]."""
    full = d.get(key)
    if isinstance(full, str) and len(full) > SHORT:
        d[key + "_detail"] = full
        d[key] = short if short is not None and len(short) <= SHORT else full[:SHORT - 3] + "..."
    return d


def finish_line(out: dict) -> dict:
    """Last step before the line is printed: `roofline` re-ordered so that ROOFLINE_FIRST leads (everything else keeps its
    order behind), `frac_of_probe_ceiling` = dominant kernel's achieved rate / the same-run probe ceiling of its mix, and the
    strings a reader needs first kept short (full texts under *_detail)."""
    rl = out.get("roofline")
    if isinstance(rl, dict):
        ceil = rl.get("probe_ceiling_dominant_mix_GBps")
        if ceil and rl.get("achieved"):
            rl["frac_of_probe_ceiling"] = rl["achieved"] / ceil
        src = rl.get("traffic_source")
        if isinstance(src, str):
            if src.startswith("same run"):
                shorten(rl, "traffic_source", "same run: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes, child process")
            else:             # a committed summary of another box and day (N > 1 lines, or the same-run passes failed)
                rl["traffic_source_detail"] = src
                rl["traffic_source"] = "committed file, not this run: " + src
                shorten(rl, "traffic_source")
        shorten(rl, "what", "dominant kernel: bytes moved / mean launch time (HIP events, kernel stream) / 8 TB/s")
        out["roofline"] = {**{k: rl[k] for k in ROOFLINE_FIRST if k in rl}, **{k: v for k, v in rl.items() if k not in ROOFLINE_FIRST}}
    cfg = out.get("config")
    if isinstance(cfg, dict):
        shorten(cfg, "workload")
        shorten(cfg, "flavor")
        shorten(cfg, "flavor_note")
        shorten(cfg, "parallelism")
    return out


NO_RETRY_MARK = "NKA_BENCH_NO_RETRY"      # a rank prints this on stderr when a second attempt could not help


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_rank_group(nproc, script, script_args, timeout_s, env=None):
    """One attempt (run_rank_group_once) -- repeated, at once and at most twice, if the launcher found the port it was given
    taken (EADDRINUSE in the static rendezvous: the port is probed by binding port 0 and released again before the launcher
    binds it; another process of the box can get in between).  That is no failure of the communication path and must not
    push the run onto the staged fallback."""
    t0 = time.perf_counter()
    for attempt in range(3):
        rc, line, err, timed_out = run_rank_group_once(nproc, script, script_args, max(30, timeout_s - (time.perf_counter() - t0)), env)
        if line is not None or timed_out or rc == 0 or "EADDRINUSE" not in err or attempt == 2:
            return rc, line, err, timed_out
        sys.stderr.write("[bench] the rendezvous port was taken between the probe and the launcher's bind: launching again\n")
    return rc, line, err, timed_out


def run_rank_group_once(nproc, script, script_args, timeout_s, env=None):
    """ONE attempt: `python -m torch.distributed.run --nproc-per-node nproc script args` as a child in a process group
    of its own (start_new_session; a subprocess, never an exec: this process has touched no GPU and stays alive to
    supervise).  stdout is read line by line (the last line that parses as a JSON object is the result), stderr is
    passed through and its tail kept.  When `timeout_s` expires the WHOLE group gets SIGTERM, then SIGKILL.
    Returns (rc, json_line or None, stderr_tail, timed_out)."""
    import signal
    import subprocess
    import threading
    env = dict(os.environ if env is None else env)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + list(script_args)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1,
                         start_new_session=True)
    found, tail = [None], []

    def pump_out():
        for ln in p.stdout:
            t = ln.strip()
            if t.startswith("{") and t.endswith("}"):
                try:
                    json.loads(t)
                    found[0] = t
                    continue
                except ValueError:
                    pass
            sys.stderr.write("[rank stdout] " + ln)

    def pump_err():
        for ln in p.stderr:
            sys.stderr.write(ln)
            tail.append(ln)
            del tail[:-200]

    th = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for t in th:
        t.start()

    def kill_group(sig):
        try:
            os.killpg(p.pid, sig)           # p.pid is the group's id (start_new_session)
        except (ProcessLookupError, PermissionError):
            pass

    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):       # a driver that gives up on us takes the ranks down too
        try:
            old[sg] = signal.signal(sg, lambda n, f: (kill_group(signal.SIGKILL), os._exit(128 + n)))
        except ValueError:                            # not the main thread (tests)
            pass
    timed_out = False
    try:
        try:
            rc = p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            sys.stderr.write(f"[bench] watchdog: no result after {timeout_s} s; killing the rank group {p.pid}\n")
            kill_group(signal.SIGTERM)
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
            kill_group(signal.SIGKILL)        # the group, not just torchrun: ranks that ignored SIGTERM too
            p.wait()
            rc = 124
    finally:
        if timed_out or p.poll() is None:     # (never after a normal exit: the group id of a reaped child is not ours any more)
            kill_group(signal.SIGKILL)        # nothing of a group that was given up on outlives the attempt
        for sg, h in old.items():
            signal.signal(sg, h)
    for t in th:
        t.join(timeout=5)
    return rc, found[0], "".join(tail), timed_out


def launch_ranks(args, argv, script=None):
    """The plain form `python bench.py --gpus N`, N > 1, WORLD_SIZE unset (VERDICT r3 task 1).  At most two attempts,
    both bounded: (1) the arguments as given; (2) only if (1) produced no line, was not a rehearsal already and did
    not say a retry is pointless: `--backend gloo --allreduce staged` -- no RCCL anywhere, one rank per GPU, the
    336-byte all-reduce staged through the host; the line then names that hook in config.parallelism and carries
    `launch.first_attempt`.  Prints the ONE JSON line and returns the exit code."""
    script = script or os.path.abspath(__file__)
    t0 = time.perf_counter()
    # ONE deadline for both attempts: launch_timeout for the first, and what is left of launch_timeout + 150 s for
    # the second -- a driver that allows ~7 minutes for a bench line is never outlasted.
    deadline = t0 + args.launch_timeout + 150
    rc, line, err, timed_out = run_rank_group(args.gpus, script, argv, args.launch_timeout)
    note = None
    # A second attempt can only help where the COMMUNICATION path is what failed: the first attempt hung (watchdog: a
    # collective that never returned) or its stderr names RCCL / NCCL / the communicator / the self-test all-reduce.  An
    # out-of-memory rank, a Python exception elsewhere or a build failure would only fail the same way again, slowly.
    comm_words = ("rccl", "nccl", "communicator", "all-reduce", "allreduce", "hipipc", "watchdog", "timed out", "timeout",
                  "rendezvous", "sigalrm", "collective")
    comm_failure = timed_out or any(w in err.lower() for w in comm_words)
    if line is None and not comm_failure and not args.no_fallback and args.allreduce != "staged":
        sys.stderr.write("[bench] the first attempt failed without a sign of the communication path: no second attempt\n")
    if line is None and comm_failure and not args.no_fallback and args.allreduce != "staged" and NO_RETRY_MARK not in err:
        why = "watchdog expired" if timed_out else f"exit code {rc}"
        lines = err.splitlines()                   # the ranks' own last words, not torchrun's failure report behind them
        cut = max([i for i, ln in enumerate(lines) if "_run_module_as_main" in ln] or [len(lines) + 1]) - 1
        own = [ln.strip() for ln in lines[:cut] if ln.strip() and not re.match(r"[WEI]\d{4} |\[[WE]", ln)]
        last = own[-1:] or [ln.strip() for ln in err.splitlines() if ln.strip()][-1:] or [""]
        note = {"first_attempt": f"{why} with --allreduce {args.allreduce}", "last_stderr_line": last[0][-300:],
                "first_attempt_s": round(time.perf_counter() - t0, 1)}
        sys.stderr.write(f"[bench] first attempt failed ({why}); second attempt with the host-staged all-reduce over gloo\n")
        left = int(max(45, min(args.launch_timeout, deadline - time.perf_counter())))
        rc, line, err, timed_out = run_rank_group(args.gpus, script, list(argv) + ["--backend", "gloo", "--allreduce", "staged"],
                                                  left)
    if line is not None:
        if note:
            d = json.loads(line)
            d["launch"] = note
            line = json.dumps(d)
        print(line, flush=True)
    return rc if (rc != 0 or line is not None) else 1


def main(argv=None):
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain form: supervise a child rank group; nothing here may import torch or touch the GPU
        raise SystemExit(launch_ranks(args, sys.argv[1:] if argv is None else list(argv)))
    import torch
    import torch.distributed as dist

    import nka_amd
    from nka_amd import dist as nd
    from nka_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        # a sharded run can only hang in a collective: never wait for ever (SIGALRM ends the
        # rank with a traceback and a non-zero code; torchrun then takes the others down, and the
        # supervising parent of the plain form makes its second attempt).  Armed AFTER the imports
        # (the first `import torch` on a fresh box takes a minute or two).
        import faulthandler
        import signal
        limit = int(os.environ.get("NKA_BENCH_WATCHDOG_S", "240"))
        faulthandler.register(signal.SIGALRM, all_threads=True, chain=True)
        signal.alarm(limit)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("NKA_BENCH_SHARE_GPU") == "1"      # rehearsal: every rank on cuda:0
    if share_gpu:
        local_rank = 0
    if args.backend == "gloo" and args.allreduce not in ("staged", "p2p"):
        raise SystemExit(f"--backend gloo needs --allreduce staged or p2p ({NO_RETRY_MARK})")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible ({NO_RETRY_MARK})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # CONTROL plane on gloo, always: the collective decisions about the all-reduce hook, the barriers around the
        # timed region, the max-over-ranks time and the replica digests must work when RCCL is the thing that is
        # broken.  The DATA path (the all-reduces of an update: 1 + (1+2*mvec) doubles, or 2+2*mvec at once in the fast mode) is RCCL: the library's own
        # communicator, else torch.distributed's nccl group created on demand, else staged through the host.
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=min(limit, 200)))

    def nccl_data_group():
        import datetime
        return dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)

    n_global, m = int(args.n), args.mvec
    lo, hi = nd.slice_bounds(n_global, world, rank)
    n_local = hi - lo
    K = args.steps
    W = args.warmup if args.warmup is not None else m + 4
    # The metric is STEADY-STATE accel_update (subspace full).  If the caller's W
    # is too short to fill the subspace, untimed priming calls are added in front
    # of the W warm-up steps (state set-up, like generating the inputs).
    prime = max(0, (m + 2) - W)
    W_all = prime + W

    FLAVORS = {"default": nka_amd.FLAVOR_DEFAULT, "f08": nka_amd.FLAVOR_F08, "f08vec": nka_amd.FLAVOR_F08_VECTOR,
               "c": nka_amd.FLAVOR_C}
    FLAVOR_NAMES = {nka_amd.FLAVOR_F08: "f08", nka_amd.FLAVOR_F08_VECTOR: "f08vec", nka_amd.FLAVOR_C: "c"}
    hook_box = ["none"]

    SUMS = {"default": nka_amd.SUMS_AUTO, "rounded": nka_amd.SUMS_BLOCKED_ROUNDED, "blocked": nka_amd.SUMS_BLOCKED}
    norm_pass = args.sums != "blocked"        # (the default resolves to the rounded passes beyond 64 elements: include/nka_hip.h)

    def make_acc(flavor_name):
        acc = nka_amd.nka().init(n_local, m, flavor=FLAVORS[flavor_name]).set_sum_order(SUMS[args.sums])
        if world > 1 or os.environ.get("NKA_BENCH_FORCE_HOOK") == "1":
            if not dist.is_initialized():          # single-process rehearsal of the N > 1 plumbing
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
                dist.init_process_group("gloo", rank=0, world_size=1)
            # collective decision over the control plane (all ranks end up with the same hook) + a proven test
            # all-reduce; ladder rccl -> torch (nccl group) -> staged (host, gloo), entered at --allreduce
            prefer = args.allreduce if hook_box[0] == "none" else hook_box[0]
            ladder = ("staged",) if args.backend == "gloo" and args.allreduce != "p2p" else \
                ("p2p", "staged") if args.backend == "gloo" else ("p2p", "rccl", "torch", "staged")
            # ("p2p" is entered only when asked for: attach_allreduce starts the ladder at `prefer`)
            hook_box[0] = nd.attach_allreduce(acc, rank, world, prefer=prefer, data_group=nccl_data_group, ladder=ladder)
            if hook_box[0] != args.allreduce and rank == 0:
                print(f"[bench] all-reduce hook: asked for '{args.allreduce}', running '{hook_box[0]}'", file=sys.stderr, flush=True)
        return acc

    def check_replicas(acc, where):
        """The scalar state (lists, h, c, all-reduced sums) is replicated: every rank
        must hold the same bits, or the ranks have diverged.  Outside
        the timed region.  Aborts the whole job loudly on a mismatch."""
        digs = nd.replica_digests(acc)
        same = all(d == digs[0] for d in digs)
        if not same:
            msg = f"[bench] FATAL: replicated NKA state differs across ranks {where}: " + \
                  ", ".join(f"rank{r}={d:016x}" for r, d in enumerate(digs))
            print(msg, file=sys.stderr, flush=True)
            raise SystemExit(3)
        return {"where": where, "ranks": len(digs), "digest": f"{digs[0]:016x}", "identical": True}

    acc = make_acc(args.flavor)
    flavor = FLAVOR_NAMES[acc.flavor()]      # "default" resolved by the library, like every front end's init
    is_default = FLAVOR_NAMES[nka_amd.nka.default_flavor()] == flavor

    # ---- inputs: resident in HBM before the timed region ----------------------
    free_b, _ = torch.cuda.mem_get_info(dev)
    pool_cap = max(2, int((free_b * 0.85) // (8 * max(n_local + 1, 2))))
    P = min(W_all + K, pool_cap)
    n_pad = n_local + (n_local % 2)            # keep every row 16-byte aligned
    pool_store = torch.empty((P, max(n_pad, 2)), dtype=torch.float64, device=dev)
    pool = [pool_store[j, :n_local] for j in range(P)]
    refill_in_timed_region = (W_all + K) > P

    basis_box = [None]

    def fill(j, t, workload="full"):
        if workload == "full":
            synth.fill_torch(pool[j], SEED, t, lo, n_global)
            return
        # drops: f_t = sum_j c_tj B_j with D fixed random vectors B_j (this rank's slices of them): every difference of
        # two inputs lies in span(B), so the subspace can hold D vectors and every further update drops one as dependent
        D = args.drop_dim
        if basis_box[0] is None:
            B = torch.empty((D, max(n_pad, 2)), dtype=torch.float64, device=dev)
            for q in range(D):
                synth.fill_torch(B[q, :n_local], SEED + 777, q, lo, n_global)
            basis_box[0] = B
        coef = torch.from_numpy(synth.fill_numpy(SEED + 60, t, 0, D, D)).to(dev)
        torch.mv(basis_box[0][:, :n_local].t(), coef, out=pool[j])

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    ev_stride = 1 if n_local >= 50_000_000 else 4

    def measure(acc, workload="full", out_of_place=False):
        """W_all untimed calls (priming + warm-up), then EXACTLY K timed updates
        bracketed by barrier + synchronize; returns wall time (max over ranks),
        mean per-phase device times (HIP events on the kernel stream), num_vec.
        workload "drops": inputs confined to a low-dimensional span and ONE device
        synchronisation per update, inside the timed region, where a solver reads its
        residual norm -- that is what lets the host see the list word (nka_hip_list_bound)."""
        sync_each = workload == "drops"
        # out of place (nka_hip_accel_update_swap): every input row is HANDED to the accelerator (it becomes w of the new
        # pair) and never written again -- W_all + K distinct rows; the buffers it returns are not needed here
        update = (lambda row: acc.accel_update_swap(row, views=False)) if out_of_place else acc.accel_update
        for t in range(min(P, W_all + K)):     # inputs (re)generated outside the timed region
            fill(t, t, workload)
        for t in range(W_all):
            if t >= P:
                fill(t % P, t, workload)
            update(pool[t % P])
            if sync_each:
                torch.cuda.synchronize(dev)
        sync_all()
        nv0 = acc.num_vec()
        checks = [check_replicas(acc, "after warm-up")] if world > 1 or hook_box[0] != "none" else []
        # per-phase HIP events on the kernel stream: every update at the headline size, every 4th
        # below 5e7 elements per GPU (four event records widen an update's kernel boundaries by
        # ~15 us: 0.2 % of a 6 ms update, but 2 % of a 0.9 ms shard update)
        acc.set_timing(min(-(-K // ev_stride), 4096), stride=ev_stride)
        sync_all()
        t0 = time.perf_counter()
        for s in range(K):
            t = W_all + s
            if t >= P:
                fill(t % P, t, workload)       # only when HBM cannot hold W+K inputs (reported below)
            update(pool[t % P])
            if sync_each:
                torch.cuda.synchronize(dev)
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64)        # control plane (gloo)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        nrec = min(-(-K // ev_stride), 4096)
        ph = [acc.timing_ms(b) for b in range(nrec)]
        mean = [sum(p[i] for p in ph) / nrec for i in range(4)]
        # device events around each update; median and min next to the mean
        stats = {}
        for i, name in enumerate(("PA_k_dots", "k_solve", "PB_k_combine", "whole_update")):
            col = sorted(p[i] for p in ph)
            stats[name] = {"mean": mean[i], "median": (col[(nrec - 1) // 2] + col[nrec // 2]) / 2.0, "min": col[0],
                           "max": col[-1]}
        stats["samples"] = nrec
        if checks:
            checks.append(check_replicas(acc, "after the timed steps"))
        return elapsed, mean, nv0, acc.num_vec(), checks, stats

    elapsed, mean, nv, nv_end, replica_check, stats = measure(acc, args.workload)
    k_steady = m if args.workload == "full" else min(m, args.drop_dim)     # vectors the subspace holds in steady state
    steady = (nv == k_steady)

    # an exchange of a sharded update by itself: 2 + 2 mvec doubles through the installed hook (the larger of the default's two), back to back on the kernel
    # stream (device events; max over ranks) -- the latency figure that decides between RCCL and the peer-to-peer exchange
    exchange = None
    if world > 1 or hook_box[0] != "none":
        # (every rank passes the same two barriers whatever happens locally)
        err, us_local = None, 0.0
        reps = 200 if hook_box[0] != "staged" else 20
        xb = torch.zeros(2 + 2 * m, dtype=torch.float64, device=dev)
        try:
            for _ in range(20):
                acc.allreduce_now(xb)
        except Exception as exc:      # an extra, never the measured path
            err = repr(exc)
        sync_all()
        try:
            if err is None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    acc.allreduce_now(xb)
                e1.record()
                torch.cuda.synchronize(dev)
                us_local = e0.elapsed_time(e1) / reps * 1e3
        except Exception as exc:
            err = repr(exc)
        sync_all()
        us = torch.tensor([us_local if err is None else float("inf")], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(us, op=dist.ReduceOp.MAX)
        if float(us.item()) != float("inf"):
            exchange = {"hook": hook_box[0], "doubles": 2 + 2 * m, "us_back_to_back": float(us.item()), "reps": reps,
                        "what": "nka_hip_allreduce_now through the installed hook, back to back on the kernel stream, device events, "
                                "max over ranks; inside an update the exchange sits between the final sums and the scalar step "
                                "(the peer-to-peer exchange is then fused into those two kernels: no kernel of its own)"}
        else:
            exchange = {"error": err or "failed on another rank"}

    # Opt-in (NKA_BENCH_P2P_PROBE=1, N > 1): the same 2 + 2 mvec doubles through the PEER-TO-PEER EXCHANGE on a small accelerator
    # of its own, back to back -- the figure to hold against `us_back_to_back` of the RCCL hook on the day a multi-GPU node runs
    # this (include/nka_hip.h: "needs a measured win over RCCL before it is preferred").  Bounded: every wait gives up after
    # 200 ms, a failed set-up or self-test is recorded and nothing else is touched.
    if world > 1 and exchange and os.environ.get("NKA_BENCH_P2P_PROBE") == "1" and hook_box[0] != "p2p":
        os.environ.setdefault("NKA_HIP_P2P_TIMEOUT_MS", "200")
        probe_acc, perr, pus = None, None, 0.0
        try:
            probe_acc = nka_amd.nka().init(4096, m)
            nd.attach_allreduce(probe_acc, rank, world, prefer="p2p", ladder=("p2p",))
            xb = torch.zeros(2 + 2 * m, dtype=torch.float64, device=dev)
            for _ in range(20):
                probe_acc.allreduce_now(xb)
            torch.cuda.synchronize(dev)
            probe_acc.num_vec()                        # (raises if a wait timed out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                probe_acc.allreduce_now(xb)
            e1.record()
            torch.cuda.synchronize(dev)
            probe_acc.num_vec()
            pus = e0.elapsed_time(e1) / 200 * 1e3
        except Exception as exc:      # an extra, never the measured path
            perr = repr(exc)
        sync_all()
        pt = torch.tensor([pus if perr is None else float("inf")], dtype=torch.float64)
        dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        exchange["p2p_probe"] = ({"us_back_to_back": float(pt.item()), "reps": 200,
                                  "what": "k_p2p_allreduce (one send-and-gather kernel per call) on mailboxes mapped through hipIpc"}
                                 if float(pt.item()) != float("inf") else {"error": perr or "failed on another rank"})
        if probe_acc is not None:
            sync_all()
            probe_acc.delete()

    # what every rank saw: which GPU, which RCCL, how many ranks ITS communicator connected, its digest of the
    # replicated state -- so that a multi-GPU record proves by itself that RCCL reduced over N ranks
    rank_info = None
    if world > 1 or hook_box[0] != "none":
        name, ncu = acc.device_info()
        mine = {"rank": rank, "local_rank": local_rank, "device": f"{name} ({ncu} CUs)", "hook": hook_box[0],
                "comm_nranks_rank": list(acc.comm_info()), "state_digest": f"{acc.state_digest():016x}",
                "n_local": n_local, "slice": [lo, hi]}
        if hook_box[0] == "rccl":
            mine["rccl_library"] = nka_amd.nka.rccl_library()
        if world > 1:
            rank_info = [None] * world
            dist.all_gather_object(rank_info, mine)
        else:
            rank_info = [mine]

    def config2_line(steps=50):
        """BASELINE configs[1] (n = 1e7, m = 10, one GPU: "single-GPU plumbing") in the same run, on the first 1e7
        elements of the resident inputs: default flavour, warm-up m+4, `steps` timed updates by wall clock with the
        per-phase events on every 4th (fixed costs are 8 % of an update here; four event records would widen it)."""
        n2, m2 = 10**7, 10
        if n_local < n2 or P < m2 + 4 + 8:
            return None
        a2 = nka_amd.nka().init(n2, m2, flavor=FLAVORS[args.flavor]).set_sum_order(SUMS[args.sums])
        try:
            views = [pool[j][:n2] for j in range(P)]
            for j in range(P):
                synth.fill_torch(views[j], SEED, 1000 + j, 0, n2)
            t = 0
            for _ in range(m2 + 4):
                a2.accel_update(views[t % P]); t += 1
            torch.cuda.synchronize(dev)
            if a2.num_vec() != m2:
                return {"error": "subspace not full"}
            a2.set_timing(-(-steps // 4), stride=4)
            # (inputs are re-used round robin: an accelerated f is as good an input as a fresh one for the traffic)
            t0 = time.perf_counter()
            for _ in range(steps):
                a2.accel_update(views[t % P]); t += 1
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / steps
            nrec = -(-steps // 4)
            ph = [a2.timing_ms(b) for b in range(nrec)]
            mean2 = [sum(p[i] for p in ph) / nrec for i in range(4)]
            fl2 = FLAVOR_NAMES[a2.flavor()]
            w2 = words_moved(fl2, m2, m2, norm_pass=norm_pass)
            moved = 8.0 * n2 * sum(w2.values())
            return {"workload": "BASELINE configs[1]: n=1e7, mvec=10, fp64, 1 GPU, subspace full", "flavor": FLAVOR_TEXT[fl2],
                    "value": 1.0 / dt, "unit": "updates/s", "us_per_update": 1e6 * dt, "steps": steps,
                    "steady_state": bool(a2.num_vec() == m2),
                    "whole_update": {"bytes_moved": moved, "achieved": moved / dt / 1e9, "frac": moved / dt / 1e9 / HBM_PEAK_GBPS,
                                     "what": "bytes moved / wall time per update / 8 TB/s"},
                    "phase_us": {"PA_k_dots": 1e3 * mean2[0], "k_solve": 1e3 * mean2[1], "PB_k_combine": 1e3 * mean2[2]}}
        finally:
            a2.delete()

    def host_array_extra(steps=6):
        """What an UNCHANGED caller of the reference gets: accel_update(f) on a HOST array (the reference's signature takes
        host memory, F08:252), i.e. nka_hip_accel_update_host -- H2D copy of f, the update, D2H copy, synchronised -- at
        BASELINE configs[1]'s size (n = 1e7, m = 10), pageable and pinned host memory.  PCIe-inclusive: never `value`."""
        import numpy as np
        n2, m2 = 10**7, 10
        a5 = nka_amd.nka().init(n2, m2, flavor=FLAVORS[args.flavor])
        try:
            dbuf = torch.empty(n2, dtype=torch.float64, device=dev)
            for t in range(m2 + 2):                       # fill the subspace through the device entry
                synth.fill_torch(dbuf, SEED, 2000 + t, 0, n2)
                a5.accel_update(dbuf)
            torch.cuda.synchronize(dev)
            res = {"entry": "nka_hip_accel_update_host (include/nka_hip.h): H2D copy of f, update, D2H copy, stream synchronised",
                   "workload": "n=1e7, mvec=10, fp64, subspace full; 2 x 80 MB cross PCIe per update", "unit": "updates/s"}
            for name, arr in (("pageable", np.empty(n2)), ("pinned", torch.empty(n2, dtype=torch.float64).pin_memory().numpy())):
                dt = []
                for t in range(steps):
                    synth.fill_torch(dbuf, SEED, 3000 + t, 0, n2)
                    arr[:] = dbuf.cpu().numpy()
                    t0 = time.perf_counter()
                    a5.accel_update(arr)
                    dt.append(time.perf_counter() - t0)
                med = float(np.median(dt[1:]))
                res[name] = {"value": 1.0 / med, "ms_per_update": 1e3 * med, "pcie_GBps": 2 * 8.0 * n2 / med / 1e9}
            res["steady_state"] = bool(a5.num_vec() == m2)
            return res
        finally:
            a5.delete()

    def with_drops_extra():
        """The same accelerator shape with a SHRUNK subspace (VERDICT r3 task 2): every input in a drop-dim dimensional
        span, so every update takes a dependence drop (F08:326-345) and the subspace holds k = drop-dim < mvec vectors;
        one synchronisation per update (the solver's residual norm), inside the timed time.  The host learns the list
        length from the list word and launches PA at exactly L = k; PB is launched one wider than it turns out to need
        (its width is fixed before the device decides the drop of this very update) -- one dead ring slot."""
        D = min(args.drop_dim, m)
        a3 = nka_amd.nka().init(n_local, m, flavor=FLAVORS[args.flavor]).set_sum_order(SUMS[args.sums])
        try:
            e3, mean3, nv3, nv3_end, _, stats3 = measure(a3, "drops")
            fl3 = FLAVOR_NAMES[a3.flavor()]
            lb = a3.list_bound()
            # ... and the same accelerator in the BIT-IDENTICAL mode on these correlated inputs (sums that drift: hardly a
            # block of a sum meets an end of its binade), beside the full workload's figure on uniform random vectors
            # (cpu_baseline.device_reference_order, where the outputs are compared with the compiled reference's)
            try:
                a3.set_sum_order(nka_amd.SUMS_REFERENCE_ORDER)
                t_in, R = W_all + K, 4
                for s_ in range(2):                                   # (the first such update allocates the block arrays)
                    fill(0, t_in, "drops")
                    a3.accel_update(pool[0])
                    t_in += 1
                for s_ in range(R):
                    fill(s_ % P, t_in + s_, "drops")
                torch.cuda.synchronize(dev)
                t_ref = time.perf_counter()
                for s_ in range(R):
                    a3.accel_update(pool[s_ % P])
                    torch.cuda.synchronize(dev)
                dt_ref = (time.perf_counter() - t_ref) / R
                ref3 = {"mode": "nka_hip_set_sum_order(NKA_HIP_SUMS_REFERENCE_ORDER) on the same inputs", "value": 1.0 / dt_ref,
                        "unit": "updates/s", "ms_per_step": 1e3 * dt_ref, "num_vec": a3.num_vec(), "updates_timed": R}
            except Exception as exc:                                  # an extra, never the measured path
                ref3 = {"value": None, "error": repr(exc)}
            return {"reference_order": ref3,
                    "workload": f"every input in a {D}-dimensional span: one dependence drop per update, num_vec = {nv3} "
                                f"of mvec = {m}; torch.cuda.synchronize() after every update (inside the timed region)",
                    "flavor": FLAVOR_TEXT[fl3], "value": K / e3, "unit": "updates/s", "ms_per_step": 1e3 * e3 / K,
                    "mean_k": float(nv3 + nv3_end) / 2.0, "steady_state": bool(nv3 == D and nv3_end == D),
                    "host_list_bound_after_the_run": lb, "list_length_on_device": nv3_end + 1,
                    "launch_widths": {"PA": lb - 1, "PB": min(lb, m)},
                    "roofline": roofline_block(fl3, n_local, m, mean3, None, stats3, None, L=D, k=D, norm_pass=norm_pass)}
        finally:
            a3.delete()
            basis_box[0] = None

    def out_of_place_extra():
        """The opt-in out-of-place entry on the headline workload: same inputs, same protocol, nka_hip_accel_update_swap in
        place of nka_hip_accel_update (bit-identical results: tests/test_hip_round4.py).  PB stores w1', v1', v_new only."""
        if refill_in_timed_region:
            return {"value": None, "error": "needs W + K distinct resident input rows"}
        a4 = nka_amd.nka().init(n_local, m, flavor=FLAVORS[args.flavor])
        try:
            e4, mean4, nv4, nv4_end, _, stats4 = measure(a4, "full", out_of_place=True)
            fl4 = FLAVOR_NAMES[a4.flavor()]
            return {"entry": "nka_hip_accel_update_swap (include/nka_hip.h): the caller's buffer becomes w_new, f_out is stored "
                             "once (as v_new) and lent to the caller",
                    "flavor": FLAVOR_TEXT[fl4], "value": K / e4, "unit": "updates/s", "ms_per_step": 1e3 * e4 / K,
                    "steady_state": bool(nv4 == m and nv4_end == m),
                    "roofline": roofline_block(fl4, n_local, m, mean4, None, stats4, None, out_of_place=True, norm_pass=norm_pass)}
        finally:
            a4.delete()                 # (before the input rows it holds are released)

    def other_sum_mode_extra():
        """The headline workload in the sum mode the line is NOT quoted in: with the default sums (the norm first, PA on the rounded
        w1': 51 words per element with compact storage) the opt-in single-pass fast mode NKA_HIP_SUMS_BLOCKED (49 words), and
        the other way round."""
        other = "rounded" if args.sums == "blocked" else "blocked"
        a6 = nka_amd.nka().init(n_local, m, flavor=FLAVORS[args.flavor]).set_sum_order(SUMS[other])
        try:
            e6, mean6, nv6, nv6_end, _, stats6 = measure(a6, "full")
            return {"mode": "nka_hip_set_sum_order(NKA_HIP_SUMS_BLOCKED): ONE pure-read pass forms every sum, the Gram row from raw sums "
                            "(include/nka_hip.h); opt-in" if other == "blocked" else
                            "nka_hip_set_sum_order(NKA_HIP_SUMS_BLOCKED_ROUNDED): norm pass + PA on the rounded w1' (the default)",
                    "sums": other, "value": K / e6, "unit": "updates/s", "ms_per_step": 1e3 * e6 / K,
                    "steady_state": bool(nv6 == m and nv6_end == m),
                    "phase_ms": {"PA_phase": mean6[0], "k_solve": mean6[1], "PB_k_combine": mean6[2]},
                    "words_per_element": sum(words_moved(FLAVOR_NAMES[a6.flavor()], m, m, norm_pass=(other != "blocked")).values())}
        finally:
            a6.delete()

    drops = None
    headline = world == 1 and (n_global, m) == (10**8, 20) and args.workload == "full" and not args.no_cpu_baseline
    if headline and os.environ.get("NKA_BENCH_SECONDARY", "1") != "0":
        try:
            drops = with_drops_extra()
        except Exception as exc:           # an extra, never the measured path
            drops = {"value": None, "error": repr(exc)}
    rounded = None                          # (the name of round 5: the measurement in the OTHER sum mode)
    if headline and os.environ.get("NKA_BENCH_SECONDARY", "1") != "0":
        try:
            rounded = other_sum_mode_extra()
        except Exception as exc:           # an extra, never the measured path
            rounded = {"value": None, "error": repr(exc)}
    oop = None
    if headline and os.environ.get("NKA_BENCH_SECONDARY", "1") != "0":
        try:
            oop = out_of_place_extra()
        except Exception as exc:           # an extra, never the measured path
            oop = {"value": None, "error": repr(exc)}

    # Secondary figure in the same run: the src-F08 rounding (two stored vectors
    # per pair, bit-faithful to F08:397), same workload, same protocol.
    also = None
    if flavor == "c" and args.workload == "full" and os.environ.get("NKA_BENCH_SECONDARY", "1") != "0":
        acc.delete()
        acc = make_acc("f08")
        e2, mean2, nv2, nv2_end, _, stats2 = measure(acc)
        also = {"flavor": FLAVOR_TEXT["f08"],
                "value": K / e2, "unit": "updates/s", "ms_per_step": 1e3 * e2 / K,
                "steady_state": bool(nv2 == m and nv2_end == m),
                "roofline": roofline_block("f08", n_local, m, mean2, None, stats2, norm_pass=norm_pass)}

    if rank == 0:
        Lk = k_steady
        rl = roofline_block(flavor, n_local, m, mean, None, stats, None, L=Lk, k=Lk, norm_pass=norm_pass)
        wl = (f"BASELINE configs[{2 if world == 1 else 3}]: synthetic n={n_global} (global), mvec={m}, fp64, subspace full "
              f"(num_vec={nv})") if args.workload == "full" else \
             (f"NOT a BASELINE config: inputs in a {args.drop_dim}-dim span, n={n_global}, mvec={m}, one drop + one sync per "
              f"update (num_vec={nv})")
        out = {
            "metric": ("NKA updates/sec + achieved HBM GB/s at n=1e8, m=20 fp64; 1/2/4/8 GPUs"   # BASELINE.json
                       if (n_global, m) == (10**8, 20) else
                       "NKA updates/sec + achieved HBM GB/s at n=%.3g, m=%d fp64" % (n_global, m)),
            "value": K / elapsed, "unit": "updates/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl, "inputs": "uniform(-1,1) fp64, counter-based generator keyed on (seed, call, global index)",
                       "n_global": n_global, "n_local": n_local, "mvec": m,
                       "flavor": FLAVOR_TEXT[flavor],
                       "sums": ("default: norm pass + PA on the rounded w1' (NKA_HIP_SUMS_BLOCKED_ROUNDED)" if norm_pass
                                else "NKA_HIP_SUMS_BLOCKED (opt-in): one pure-read pass, raw-sum Gram row"),
                       "flavor_is_front_end_default": bool(is_default),
                       "flavor_note": "the flavour `call a%init(vlen, mvec)` (Fortran), nka_init (F95) and nka().init "
                                      "(Python) run when the caller names none (include/nka_hip.h: NKA_HIP_FLAVOR_DEFAULT)"
                                      if is_default else "NOT the front ends' default: selected on the command line",
                       "parallelism": ("REHEARSAL (ranks share one GPU); " if share_gpu else "")
                                      + f"contiguous n-slices over {world} GPU(s); all-reduce={hook_box[0]}"
                                      + ("" if hook_box[0] == args.allreduce or hook_box[0] == "none" else
                                         f" (FALLBACK: '{args.allreduce}' was asked for and failed its set-up or self-test)")
                                      + ("; the sums are staged through the host over gloo (two PCIe hops and a host "
                                         "collective per exchange: slower than RCCL)" if hook_box[0] == "staged" else ""),
                       "control_plane": "torch.distributed gloo (decisions, barriers, timing, digests)" if world > 1 else None,
                       "steady_state": bool(steady and nv_end == k_steady), "prime_steps": prime,
                       "phase_events": f"HIP events recorded on every {ev_stride}{'st' if ev_stride == 1 else 'th'} timed update",
                       "inputs_resident": not refill_in_timed_region},
            "roofline": rl,
        }
        if replica_check:
            out["replica_check"] = replica_check
        if rank_info:
            out["ranks"] = rank_info
        if exchange:
            out["exchange"] = exchange
            if "us_back_to_back" in exchange:
                out["roofline"]["exchange_us_back_to_back"] = exchange["us_back_to_back"]      # (flat: survives a scalars-only record)
        if hook_box[0] == "rccl":
            out["config"]["rccl_library"] = nka_amd.nka.rccl_library()
        if also is not None:
            out["also_f08_rounding"] = also
            nested, flat = reference_rounding_entry(also)
            out["roofline"]["reference_rounding"] = nested
            out["roofline"].update(flat)
        if drops is not None:
            out["with_drops"] = drops
        if oop is not None:
            out["out_of_place_entry"] = oop
        if rounded is not None:
            out["sum_mode_" + rounded.get("sums", "other")] = rounded
            if rounded.get("value"):
                out["roofline"][f"sum_mode_{rounded.get('sums', 'other')}_updates_per_s"] = rounded["value"]      # (flat: survives a scalars-only record)
        if headline:
            try:
                c2 = config2_line()
                if c2:
                    out["config2_n1e7_m10"] = c2
            except Exception as exc:       # an extra, never the measured path
                out["config2_n1e7_m10"] = {"error": repr(exc)}
        lean = args.no_cpu_baseline or world > 1            # lean runs skip the CPU leg and the child-process extras
        host_arr = None
        if headline and os.environ.get("NKA_BENCH_SECONDARY", "1") != "0":
            try:
                host_arr = host_array_extra()
            except Exception as exc:       # an extra, never the measured path
                host_arr = {"error": repr(exc)}
        if world > 1 and not args.no_cpu_baseline and os.environ.get("NKA_BENCH_PROBE", "1") != "0":
            # N > 1: rank 0 adds the same-run streaming ceilings at ITS shard size (a child process on rank 0's GPU after
            # the timed region, while the other ranks wait in the final barrier: ~10 s), so that a scaling record carries
            # what this memory system gives the shard's mixes next to the event-time fractions
            probe = probe_ceilings(min(n_local, 10**8), timeout=90, device=local_rank)
            out["roofline"]["probe_ceiling"] = probe
            if probe and not probe.get("error"):
                out["roofline"]["probe_ceiling_dominant_mix_GBps"] = probe.get("mix_22R_5W_GBps" if flavor == "c" else "mix_42R_5W_GBps")
                out["roofline"]["probe_ceiling_pure_read_GBps"] = probe.get("pure_read_22_streams_GBps")
        if not lean:
            # release HBM (the child processes below need it) before the minute of CPU work
            del pool, pool_store
            acc.delete()
            torch.cuda.empty_cache()
            pool = pool_store = None
            try:
                # the CPU leg runs at the benchmark's own n unless the host cannot hold the
                # reference's 2(mvec+1) vectors (33.6 GB at n = 1e8, m = 20) with room to spare
                avail = host_mem_available_gb()
                need_gb = 8.0 * n_global * (2 * (m + 1) + 3) / 2**30
                if args.cpu_n is not None:
                    cpu_n, why = int(args.cpu_n), " (n chosen with --cpu-n)"
                elif avail is not None and avail >= need_gb + 12.0:
                    cpu_n, why = n_global, ""
                else:
                    cpu_n = min(n_global, 2 * 10**7)
                    why = (f" (n reduced: the reference needs {need_gb:.0f} GB of host memory at n = {n_global}, "
                           f"{avail if avail is None else round(avail)} GB available)")
                gen, timed = None, 6
                if cpu_n >= 5 * 10**7:            # inputs from the device generator (the numpy twin needs ~1.5 s per vector)
                    timed = 3
                    dbuf = torch.empty(cpu_n, dtype=torch.float64, device=dev)
                    hbuf = torch.empty(cpu_n, dtype=torch.float64, pin_memory=True)

                    def gen(t):
                        synth.fill_torch(dbuf, SEED, t, 0, n_global)
                        hbuf.copy_(dbuf)
                        return hbuf.numpy()
                # The device's bit-identical mode beside it (nka_hip_set_sum_order(NKA_HIP_SUMS_REFERENCE_ORDER), src-F08
                # flavour): the same inputs through a second handle, every output compared with the compiled reference's
                # -- torch.equal, at the full size -- and the steady-state updates timed.
                twin = {"acc": None, "equal": 0, "compared": 0, "ms": [], "error": None}
                if headline and cpu_n == n_local and os.environ.get("NKA_BENCH_SECONDARY", "1") != "0":
                    try:
                        twin["acc"] = nka_amd.nka().init(n_local, m, flavor=FLAVORS["f08"]).set_sum_order(nka_amd.SUMS_REFERENCE_ORDER)
                        twin["in"] = torch.empty(n_local, dtype=torch.float64, device=dev)
                        twin["ref"] = torch.empty(n_local, dtype=torch.float64, device=dev)
                    except Exception as exc:
                        twin["acc"], twin["error"] = None, repr(exc)

                def after_update(t, f_host, kind):
                    if twin["acc"] is None or kind != "reference":
                        return
                    try:
                        synth.fill_torch(twin["in"], SEED, t, 0, n_global)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        twin["acc"].accel_update(twin["in"])
                        e1.record()
                        twin["ref"].copy_(torch.from_numpy(f_host))
                        torch.cuda.synchronize(dev)
                        twin["compared"] += 1
                        twin["equal"] += int(torch.equal(twin["in"], twin["ref"]))
                        if t >= m + 2:
                            twin["ms"].append(e0.elapsed_time(e1))
                    except Exception as exc:
                        twin["error"] = repr(exc)
                        twin["acc"] = None

                out["cpu_baseline"] = cpu_baseline(m, cpu_n, timed, gen, why, after_update)
                out["cpu_baseline"]["updates_per_s_scaled_to_n_global"] = out["cpu_baseline"]["value"] * cpu_n / n_global
                if twin["compared"] or twin["error"]:
                    ms = sorted(twin["ms"])
                    med = ms[len(ms) // 2] if ms else None
                    out["cpu_baseline"]["device_reference_order"] = {
                        "what": "the same inputs through nka_hip_set_sum_order(NKA_HIP_SUMS_REFERENCE_ORDER), src-F08 flavour: every "
                                "sum formed in the reference's order (k_chain_sums); every output compared with the compiled "
                                "reference's on the host run beside it, torch.equal",
                        "outputs_compared": twin["compared"], "outputs_bit_identical": twin["equal"],
                        "ms_per_step": med, "value": (1e3 / med) if med else None, "unit": "updates/s",
                        "times_the_cpu_reference": (1e3 / med) / out["cpu_baseline"]["value"] if med else None,
                        "error": twin["error"]}
                if twin["acc"] is not None:
                    twin["acc"].delete()
                twin.clear()
                dbuf = hbuf = None
            except Exception as exc:  # the baseline is a reported extra, never the measured path
                out["cpu_baseline"] = {"value": None, "error": repr(exc)}
            torch.cuda.empty_cache()
            probe = probe_ceilings(min(n_local, 10**8))
            # HBM traffic of the dominant kernels by the PMC counters, measured in this run (child process under rocprofv3)
            pm_live = pmc_same_run(flavor, n_local, m, extra_args=("--sums", args.sums)) if headline else None
            out["roofline"] = roofline_block(flavor, n_local, m, mean, probe, stats, pm_live, L=Lk, k=Lk, norm_pass=norm_pass)
            if also is not None:
                also["roofline"]["probe_ceiling"] = probe
                nested, flat = reference_rounding_entry(also)
                out["roofline"]["reference_rounding"] = nested
                out["roofline"].update(flat)
            if rounded is not None and rounded.get("value"):
                out["roofline"][f"sum_mode_{rounded.get('sums', 'other')}_updates_per_s"] = rounded["value"]
            if drops is not None and (drops.get("reference_order") or {}).get("value"):
                out["roofline"]["reference_order_with_drops_updates_per_s"] = drops["reference_order"]["value"]
            dro = (out.get("cpu_baseline") or {}).get("device_reference_order")
            if dro and dro.get("value"):                        # (flat, like the others)
                out["roofline"]["reference_order_updates_per_s"] = dro["value"]
                out["roofline"]["reference_order_outputs_compared_with_cpu_reference"] = dro["outputs_compared"]
                out["roofline"]["reference_order_outputs_bit_identical"] = dro["outputs_bit_identical"]
            if host_arr is not None:
                out["host_array_entry"] = host_arr
                for kind in ("pageable", "pinned"):
                    if isinstance(host_arr.get(kind), dict):      # flat, so that a scalars-only record keeps it
                        out["roofline"][f"host_array_entry_n1e7_m10_{kind}_updates_per_s"] = host_arr[kind]["value"]
            if drops is not None and drops.get("roofline") and headline:
                pm_d = pmc_same_run(flavor, n_local, m, extra_args=("--workload", "drops", "--drop-dim", str(args.drop_dim), "--sums", args.sums))
                if pm_d:
                    D = min(args.drop_dim, m)
                    ph = drops["roofline"]["device_time_stats_ms"]
                    mean_d = [ph[nm]["mean"] for nm in ("PA_k_dots", "k_solve", "PB_k_combine", "whole_update")]
                    drops["roofline"] = roofline_block(flavor, n_local, m, mean_d, None, ph, pm_d, L=D, k=D, norm_pass=norm_pass)
            if not args.no_config5 and headline:
                out["config5_abstract_vector"] = config5_abstract_vector()
        print(json.dumps(finish_line(out)), flush=True)

    # tear down in a fixed order on every rank: the library's RCCL communicator first,
    # then torch's process group
    del pool, pool_store
    acc.delete()                           # (idempotent)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
