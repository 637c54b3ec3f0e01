#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref, built by
oracle/Makefile from /root/reference where it lies).  Run in the build
container only:   python oracle/make_golden.py

What is written (all small, all data -- inputs and expected outputs):
  reference_output_F08.txt / reference_output_C.txt
        the reference's own golden files (src-F08/reference_output,
        src-C/reference_output), verbatim data.
  example_tables.json
        stdout tables of the reference's example programs run here
        (src-F08 and src-F08-vector: 3 command lines each; src-C: 1).
  scenario_<name>.npz   (S1..S6 of SURVEY.md 8c, + a few more)
        op list, inputs, and per call: output f and num_vec from the F08
        reference; output f, list state (first,last,free,next,prev,h) from the C
        reference; output f from the F08-vector reference on a grid_vector.
  medium_n100000_m10.npz
        n=1e5, m=10, 25 calls: per-call ||f_out||_2, <f_out, probe>, num_vec and
        64 sampled entries from the F08 reference (inputs regenerated from the
        seed by tests).
TEST INFRASTRUCTURE ONLY.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

OPS = {"update": 0, "restart": 1, "relax": 2, "set_vec_tol": 3}


def scenario_defs():
    """name -> (n, mvec, ops, inputs).  ops rows: (opcode, input index, value)."""
    S = {}
    n = 64
    X = O.lcg_vectors(40, n, seed=1)

    def upd(i):
        return (OPS["update"], i, 0.0)

    # S1 independent inputs: capacity drops only
    S["S1_capacity"] = (n, 3, [upd(i) for i in range(10)], X[:10])

    # S2 rank-3 pool with m=5: dependence drops in mid-list, then fresh vectors
    pool = X[20:23]
    coef = O.lcg_vectors(9, 3, seed=7)
    dep = coef @ pool
    inp = np.vstack([dep, X[:6]])
    S["S2_dependence"] = (n, 5, [upd(i) for i in range(inp.shape[0])], inp)

    # S3 repeated input -> s == 0 -> relax inside accel_update
    inp = X[:6]
    S["S3_zero_difference"] = (n, 4, [upd(0), upd(1), upd(1), upd(2), upd(3), upd(3), upd(3), upd(4), upd(5)], inp)

    # S4 restart mid-stream
    S["S4_restart"] = (n, 4, [upd(0), upd(1), upd(2), upd(3), (OPS["restart"], 0, 0.0), upd(4), upd(5),
                              upd(6), upd(7), upd(8), upd(9)], X[:10])

    # S5 relax mid-stream (also relax twice and relax on an empty object)
    S["S5_relax"] = (n, 4, [(OPS["relax"], 0, 0.0), upd(0), upd(1), upd(2), (OPS["relax"], 0, 0.0),
                            (OPS["relax"], 0, 0.0), upd(3), upd(4), upd(5), (OPS["relax"], 0, 0.0), upd(6),
                            upd(7), upd(8)], X[:9])

    # S6 very strict tolerance: subspace pinned at one vector
    S["S6_vtol_0p9"] = (n, 4, [(OPS["set_vec_tol"], 0, 0.9)] + [upd(i) for i in range(8)], X[:8])

    # S7 mvec = 1 (every update is a capacity drop of the only older vector)
    S["S7_mvec1"] = (n, 1, [upd(i) for i in range(6)], X[:6])

    # S8 odd, tiny length and a wide subspace that never fills
    X8 = O.lcg_vectors(9, 7, seed=3)
    S["S8_n7_m8"] = (7, 8, [upd(i) for i in range(9)], X8)

    # S9 nearly dependent inputs with a looser tolerance change mid-stream
    base = X[30:34]
    coef = O.lcg_vectors(12, 4, seed=11)
    noise = 1e-3 * O.lcg_vectors(12, n, seed=13)
    inp = coef @ base + noise
    ops = [upd(i) for i in range(6)] + [(OPS["set_vec_tol"], 0, 1e-4)] + [upd(i) for i in range(6, 12)]
    S["S9_near_dependence"] = (n, 6, ops, inp)

    # S10 zero vector input on the first and later calls (f = 0 twice -> s == 0)
    Z = np.vstack([np.zeros((1, n)), X[:3], np.zeros((2, n)), X[3:5]])
    S["S10_zero_vectors"] = (n, 3, [upd(i) for i in range(Z.shape[0])], Z)
    return S


def run_scenario(name, n, mvec, ops, inputs):
    f08 = O.RefF08(n, mvec)
    cref = O.RefC(n, mvec, 0.01)
    # grid_vector flavour: put the n values in an (n x 1) interior with ghosts = 0
    vec = O.RefF08Vector(n, 1, mvec)
    nup = sum(1 for op in ops if op[0] == OPS["update"])
    m1 = mvec + 1
    out = dict(
        n=np.int64(n), mvec=np.int32(mvec), ops=np.array(ops, dtype=np.float64), inputs=inputs,
        f_out_f08=np.zeros((nup, n)), f_out_c=np.zeros((nup, n)), f_out_f08vec=np.zeros((nup, n)),
        num_vec=np.zeros(len(ops), np.int32),
        first=np.zeros(nup, np.int32), last=np.zeros(nup, np.int32), free=np.zeros(nup, np.int32),
        subspace=np.zeros(nup, np.int32), pending=np.zeros(nup, np.int32),
        next=np.zeros((nup, m1), np.int32), prev=np.zeros((nup, m1), np.int32), h=np.zeros((nup, m1, m1)),
    )
    u = 0
    for t, (op, idx, val) in enumerate(ops):
        idx = int(idx)
        if op == OPS["update"]:
            a = inputs[idx].copy()
            b = inputs[idx].copy()
            f08.accel_update(a)
            cref.accel_update(b)
            g = np.zeros((3, n + 2))  # (ny+2, nx+2) row-major == Fortran (nx+2, ny+2)
            g[1, 1:n + 1] = inputs[idx]
            gf = g.reshape(-1).copy()
            vec.accel_update(gf)
            out["f_out_f08"][u] = a
            out["f_out_c"][u] = b
            out["f_out_f08vec"][u] = gf.reshape(3, n + 2)[1, 1:n + 1]
            st = cref.state()
            for k in ("first", "last", "free", "subspace", "pending"):
                out[k][u] = int(getattr(st, k))
            out["next"][u] = st.next
            out["prev"][u] = st.prev
            out["h"][u] = st.h
            u += 1
        elif op == OPS["restart"]:
            f08.restart(); cref.restart(); vec.restart()
        elif op == OPS["relax"]:
            f08.relax(); cref.relax(); vec.relax()
        elif op == OPS["set_vec_tol"]:
            f08.set_vec_tol(val); vec.set_vec_tol(val)
            cref.set_vec_tol(val)     # field write: the C API takes vtol at construction only (.c:211)
        out["num_vec"][t] = f08.num_vec()
        assert f08.num_vec() == cref.num_vec() == vec.num_vec(), (name, t)
    np.savez_compressed(os.path.join(GOLD, f"scenario_{name}.npz"), **out)
    print(f"  {name}: num_vec trace {out['num_vec'].tolist()}")


def medium_case():
    n, m, calls, seed = 100000, 10, 25, 20240607
    rng = np.random.Generator(np.random.PCG64(seed))
    ref = O.RefF08(n, m)
    probe = np.cos(np.arange(n) * 0.001)
    idx = np.linspace(0, n - 1, 64).astype(np.int64)
    norms, probes, samples, nv, innorm = [], [], [], [], []
    for _ in range(calls):
        f = rng.random(n) * 2.0 - 1.0
        innorm.append(np.linalg.norm(f))
        ref.accel_update(f)
        norms.append(np.linalg.norm(f))
        probes.append(float(f @ probe))
        samples.append(f[idx].copy())
        nv.append(ref.num_vec())
    np.savez_compressed(os.path.join(GOLD, "medium_n100000_m10.npz"), n=n, mvec=m, calls=calls, seed=seed,
                        idx=idx, in_norm=np.array(innorm), out_norm=np.array(norms), out_probe=np.array(probes),
                        out_samples=np.array(samples), num_vec=np.array(nv, np.int32))
    print("  medium: num_vec", nv)


def example_tables():
    res = {}
    runs = {"f08": ("nka_example_f08", [[], ["--nka-vec", "5"], ["--sweeps", "4", "--nka-vec", "5"]]),
            "f08vec": ("nka_example_f08vec", [[], ["--nka-vec", "5"], ["--sweeps", "4", "--nka-vec", "5"]]),
            "c": ("nka_example_c", [[]])}
    for key, (exe, arglists) in runs.items():
        for args in arglists:
            with tempfile.TemporaryDirectory() as td:  # the programs write out.vtk in the cwd
                p = subprocess.run([os.path.join(O.REF_DIR, exe)] + args, cwd=td, capture_output=True,
                                   text=True, check=True)
            res[f"{key} {' '.join(args)}".strip()] = p.stdout.splitlines()
    with open(os.path.join(GOLD, "example_tables.json"), "w") as fh:
        json.dump(res, fh, indent=0)
    for k, v in res.items():
        print(f"  example [{k}]: {len(v)} lines, last: {v[-1]!r}")


def main():
    os.makedirs(GOLD, exist_ok=True)
    O.build(ref=True)
    shutil.copyfile(os.path.join(REF, "src-F08", "reference_output"), os.path.join(GOLD, "reference_output_F08.txt"))
    shutil.copyfile(os.path.join(REF, "src-C", "reference_output"), os.path.join(GOLD, "reference_output_C.txt"))
    example_tables()
    for name, (n, mvec, ops, inputs) in scenario_defs().items():
        run_scenario(name, n, mvec, ops, inputs)
    medium_case()


if __name__ == "__main__":
    main()
