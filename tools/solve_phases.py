#!/usr/bin/env python3
"""Where does the one-wavefront scalar step spend its time?  Needs a diagnostic
build of the library with s_memtime stamps:

  make -C nka_amd/csrc stamps
  NKA_HIP_DIAG_LIB=$PWD/nka_amd/libnka_hip_stamps.so python tools/solve_phases.py [--mvec 20] [--vlen 1e5]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["lst_load", "red/plan staging + list order", "gather rows", "factorisation loop", "scatter + drop replay",
         "new slot + ord + gather (phase 3 head)", "backward substitution", "comb plan stores", "lst_store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mvec", type=int, default=20)
    ap.add_argument("--vlen", type=float, default=1e5)
    a = ap.parse_args()
    import torch
    import nka_amd
    n, m = int(a.vlen), a.mvec
    # the stamps build IS a diagnostic build: NKA_HIP_DIAG_LIB=nka_amd/libnka_hip_stamps.so (make stamps)
    acc = nka_amd.nka(diagnostic=True).init(n, m)
    L = acc._L
    rng = np.random.default_rng(0)
    rows = []
    pa_rows = []
    pb_rows = []
    for t in range(m + 12):
        f = torch.from_numpy(rng.standard_normal(n)).cuda()
        acc.accel_update(f)
        st = np.zeros(16)
        L.nka_hip_get_stamps(acc._handle(), st.ctypes.data_as(C.POINTER(C.c_double)))
        if t >= m + 2:
            rows.append(np.diff(st[:10]))
            pa_rows.append(np.diff(st[10:14]))
            pb_rows.append((st[0] - st[13], st[14] - st[9], st[15] - st[14]))
    d = np.median(np.array(rows), axis=0)
    tot = d.sum()
    print(f"k_solve_rows phases, mvec={m} (median of {len(rows)} steady-state updates), s_memtime ticks (shader-clock cycles, ~2.1-2.4 GHz):")
    for nm, v in zip(NAMES, d):
        print(f"  {nm:<42s} {v:8.0f} cycles  {100 * v / tot:5.1f} %")
    print(f"  {'total inside the kernel':<42s} {tot:8.0f} cycles")
    pa = np.median(np.array(pa_rows), axis=0)
    print(f"k_dots_win block 0, n={n}: entry -> first loads issued {pa[0]:.0f}, tile loop {pa[1]:.0f}, block reduction + partial store {pa[2]:.0f} cycles")
    pb = np.median(np.array(pb_rows), axis=0)
    print(f"between the kernels (one clock for all: s_memtime), n={n}: end of PA's block 0 -> entry of the scalar step {pb[0]:.0f} cycles "
          f"(final sums + two kernel boundaries); end of the scalar step -> entry of PB's block 0 {pb[1]:.0f}; PB block 0: entry -> its "
          f"first tile done {pb[2]:.0f}")


if __name__ == "__main__":
    main()
