"""Shared tolerance rule and worst-error bookkeeping of the GPU parity tests.

The bar: ||f_hip - f_ref||_2 / ||f_in||_2 <= 1e-12 for n <= 1e5
(1e-10 at BASELINE sizes).  The coefficients solve (L L^T) z = W^T f, the stored
v's are scaled by 1/s, and last-bit differences of the inner products are amplified
by both -- the drop rule lets pivots get as small as vtol (F08:326), and a small
difference norm s makes the v's large against f.  On such inputs the REFERENCE ITSELF
is not within 1e-12 of the truth; how far it is, is MEASURED (round 4), not estimated:

  THE TRUTH RULE.  oracle/nka_oracle_exact.c is the pinned restatement compiled a second
  time in extended precision (same statements, same list logic, unit roundoff 5.4e-20).
  On the same calls it gives the "exact" trajectory f_exact, and
        err_ref = max over the reference's three flavours, cumulative over the calls so far, of
                  ||f_reference - f_exact|| / ||f_in||     (fixture outputs of the compiled
                  reference, or the oracle's flavours, each pinned to its reference bit for bit)
        err_dev = ||f_device - f_exact|| / ||f_in||
  and every test asserts, over its call sequence,
        max err_dev <= max(base, TRUTH_FACTOR * max err_ref),         TRUTH_FACTOR = 2 (4 within one tile, n <= 512: truth_factor):
  the device may be no further from the truth than twice the reference's own worst distance
  from it on the same calls, and within the stated figure wherever the reference is.  The
  comparison is per SEQUENCE, not per call: where one ill-conditioned event dominates, err_dev
  and err_ref at that event are two draws of the same rounding-error distribution (the device's
  inner products are NOT less accurate: tools/error_attribution.py), and which call a draw
  peaks at differs between them; per call, a hard stop at TRUTH_HARD (8; 32 within one tile) x the reference's distance
  so far catches a real defect where it happens (every test; the soak tool records a trip and lets the sequence run on:
  check(stop=False)).  The rule is an empirical bar with a counted exceedance rate, 25 of 18 916 soak records in round 4
  (truth_factor).  Decisions (num_vec, list order, slots, s == 0) are compared exactly, separately.

The rule of rounds 2-3 -- ||f_device - f_reference|| <= max(base, 4 x spread), spread = the
largest pairwise difference of the reference's three flavours -- is still COMPUTED and printed
per test ("K needed"), as a diagnostic only: it bounded the device by a quantity the builder
derived (and had folded four synthetic, perturbed dot products into); tools/error_attribution.py
shows what that distance consists of (at n = 61 698 it is the REFERENCE's sequential sums that
are 2.8e-9 from the truth, the device 1.7e-11).  A test that supplies neither truth nor spread
keeps the pivot rule (base above a smallest pivot of 0.5, base / pivot^2 below) and is labelled so.
Every check records what it saw; the worst per test is printed at the end of the run and written
to parity_worst.json under dump_dir() -- gpurun_out/ on a GPU box -- (err_dev_exact and err_ref_exact side by side).
"""
import json
import os

import numpy as np

WORST = {}     # key -> dict(err=..., tol=..., pivot=..., n=count)
K_SPREAD = 4.0       # diagnostic only since round 4
TRUTH_FACTOR = 2.0   # end of the sequence: max err_dev <= max(base, TRUTH_FACTOR * max err_ref)   (finish())
TRUTH_FACTOR_TINY = 4.0   # ... for vectors of at most one tile of the kernels (n <= 512), see truth_factor
TINY_N = 512
TRUTH_HARD_TINY = 32.0
TRUTH_HARD = 8.0     # every call, at once: err_dev <= max(base, TRUTH_HARD * err_ref so far)
TOUCHED = set()      # keys checked with `truth` since the last finish()
DIRECT_FACTOR = 2.0  # device vs reference, asserted directly wherever err_ref <= base / 2 (check()): err <= 2 x base


class Spread:
    """The reference's three flavours (the oracle's, each pinned to its reference flavour bit for bit) and the
    extended-precision restatement, driven in lock step on a call sequence without a fixture.
      .err_ref  = max over flavours and over the calls so far of ||f_flavour - f_exact|| / ||f_in||  (the rule)
      .exact    = f_exact of the most recent update (what the device output is measured against)
      .value    = largest pairwise ||f_a - f_b|| / ||f_in|| of the three flavours so far  (diagnostic)"""

    def __init__(self, oracle, n, m, vtol=None):
        self.n, self.m = int(n), int(m)
        self.accs = [oracle.OracleNKA(n, m, fl) for fl in (oracle.F08, oracle.F08_VECTOR, oracle.C_FLAVOR)]
        self.truth_acc = oracle.OracleExact(n, m, oracle.F08)
        if vtol is not None:
            self.set_vec_tol(vtol)
        self.value = 0.0
        self.err_ref = 0.0
        self.exact = None
        self.decisions_agree = True

    def update(self, x):
        outs = []
        for a in self.accs:
            f = np.array(x, dtype=np.float64, copy=True)
            a.accel_update(f)
            outs.append(f)
        fx = np.array(x, dtype=np.float64, copy=True)
        self.truth_acc.accel_update(fx)
        self.exact = fx
        # the truth is only the truth of THIS trajectory while it takes the reference's decisions
        if self.truth_acc.state().list_order() != self.accs[0].state().list_order():
            self.decisions_agree = False
        nx = float(np.linalg.norm(x))
        if nx > 0.0:                       # (a zero input is checked for an exactly zero result, not against a ratio)
            d = max(np.linalg.norm(outs[i] - outs[j]) for i in range(len(outs)) for j in range(i)) / nx
            self.value = max(self.value, float(d))
            self.err_ref = max(self.err_ref, max(float(np.linalg.norm(o - fx)) for o in outs) / nx)
        return self.value

    def truth(self, out_dev, x, sl=None, exact=None):
        """-> (err_dev, err_ref) for check(): the device output (optionally the slice `sl` of the vector) against the
        exact trajectory, relative to the norm of the whole input."""
        assert self.decisions_agree, "the extended-precision run took another drop decision than the reference"
        fx = self.exact if exact is None else exact
        if sl is not None:
            fx = fx[sl]
        nx = max(float(np.linalg.norm(x)), 1e-300)
        return float(np.linalg.norm(np.asarray(out_dev) - fx)) / nx, self.err_ref, self.n, self.m

    def relax(self):
        for a in self.accs + [self.truth_acc]:
            a.relax()

    def restart(self):
        for a in self.accs + [self.truth_acc]:
            a.restart()

    def set_vec_tol(self, v):
        for a in self.accs + [self.truth_acc]:
            a.set_vec_tol(v)


def fixture_spreads(g):
    """Cumulative spread per update of a scenario fixture, from the three reference
    outputs it holds (written by oracle/make_golden.py from the compiled reference).  Diagnostic."""
    ups = [int(i) for op, i, _ in g["ops"] if int(op) == 0]
    out, cur = [], 0.0
    for u, idx in enumerate(ups):
        o = [g[k][u] for k in ("f_out_f08", "f_out_f08vec", "f_out_c")]
        nx = max(np.linalg.norm(g["inputs"][idx]), 1e-300)
        cur = max(cur, max(float(np.linalg.norm(o[i] - o[j])) for i in range(3) for j in range(i)) / nx)
        out.append(cur)
    return out


def fixture_truth(g, oracle):
    """(exact outputs per update, cumulative err_ref per update) of a scenario fixture: the extended-precision
    restatement replays the fixture's operations; err_ref comes from the outputs of the COMPILED reference's three
    flavours that the fixture holds.  Also checks that the exact run takes the fixture's decisions."""
    acc = oracle.OracleExact(int(g["n"]), int(g["mvec"]), oracle.F08)
    exact, errs, cur, u = [], [], 0.0, 0
    for op, idx, val in g["ops"]:
        op, idx = int(op), int(idx)
        if op == 0:
            x = g["inputs"][idx]
            f = x.copy()
            acc.accel_update(f)
            nx = max(float(np.linalg.norm(x)), 1e-300)
            cur = max(cur, max(float(np.linalg.norm(g[k][u] - f)) for k in ("f_out_f08", "f_out_f08vec", "f_out_c")) / nx)
            exact.append(f)
            errs.append(cur)
            u += 1
        elif op == 1:
            acc.restart()
        elif op == 2:
            acc.relax()
        elif op == 3:
            acc.set_vec_tol(float(val))
    return exact, errs


def pivot_min(state):
    live = state.list_order()[1:]
    return min([abs(state.h[k - 1, k - 1]) for k in live] + [1.0])


def truth_factor(n, mvec=None):
    """2 beyond one tile of the kernels (n > 512); 4 for vectors of at most one tile.
    An EMPIRICAL bar with a counted exceedance rate (profiles/r04/fuzz_soak.txt: 18 916 soak records of
    round 4): the device's blocked, fused summation (error ~ eps log n against the reference's sequential ~ eps sqrt(n)) puts
    it closer to the extended-precision trajectory than the reference almost always -- err_dev / err_ref has median 0.05 for
    n > 2048, 0.14 for 513..2048, 0.20 for 129..512, 0.56 for n <= 16.  Within one tile a sum is exact to a few units in the
    last place in any order: device and reference are then two draws of the same rounding-error distribution (switching on
    FMA alone, or the blocked order alone, in the reference's own arithmetic moves its error by factors between 0.3 and 3.6
    there: profiles/r04/error_attribution.txt), and the ratio of two draws has a tail -- hence 4.  Beyond the allowance: 25 of
    the 18 916 records; 22 with n <= 9, and the three rank records of ONE sequence beyond one tile (sharded soak seed 3319,
    n = 1660: err_dev 1.05 .. 1.14e-12 against the base of 1e-12; profiles/r04/sharded_seed_3319_replay.txt shows what it
    is); the later soak of the final tree added two more beyond one tile (abstract-vector flavour, 1 028 and 771 elements: 3.5 x, 3.4 x)
    and, sharded over three ranks, 13 records with at most 5 elements.  Round 5 (profiles/r05/fuzz_soak.txt: 26 127 sharded records,
    3 453 other sequences): 13 more -- nine within one tile, three with 765 ... 1 024 elements (2.05 ... 2.7 x) and the first beyond
    2 048: 8 191 elements, 2.7 x.  All of them beyond one tile are replayed by tests/test_soak_regressions_gpu.py.  The thresholds were not
    moved for any of them."""
    return TRUTH_FACTOR_TINY if (n is not None and n <= TINY_N) else TRUTH_FACTOR


def truth_hard(n):
    """The per-call stop that catches a real defect where it happens: 8 x the reference's distance SO FAR; 32 x within one
    tile, where early in a sequence the reference's running maximum has often not met its own bad draw yet."""
    return TRUTH_HARD_TINY if (n is not None and n <= TINY_N) else TRUTH_HARD


def tolerance(state, base=1e-12, spread=None, truth=None):
    """-> (tol, pivot, rule).  With `truth` = (err_dev, err_ref) the tolerance applies to err_dev."""
    piv = pivot_min(state)
    if truth is not None:
        hard = truth_hard(truth[2] if len(truth) > 2 else None)
        tol = max(base * (hard / TRUTH_HARD), hard * truth[1])
        return tol, piv, ("stated" if tol == base else f"per call: {hard:g} x reference-vs-exact so far")
    if spread is not None:
        tol = max(base, K_SPREAD * spread)
        return tol, piv, ("stated" if tol == base else "reference spread")
    if piv > 0.5:
        return base, piv, "stated"
    return base / (piv * piv), piv, "conditioning (no spread supplied)"


HARD_BEYOND = []   # (key, err_dev, hard tolerance, err_ref) of sequences beyond the contract's HARD line (finish())
STOPS = []      # (key, where, err_dev, tol) of per-call stops tripped under check(stop=False): the soak tool's record


def check(err, state, key, base=1e-12, where=None, spread=None, truth=None, stop=True):
    """Record `err` = ||f_device - f_reference|| / ||f_in|| under `key` and assert the rule above.
    stop=False (the soak tool only; every test asserts): a tripped per-call stop is RECORDED (STOPS, rec["stops"]) and the
    sequence goes on, so that it is still judged by THE rule at its end -- the per-call stop measures against the reference's
    distance SO FAR, and where the errors of both grow 25-fold per update (sharded soak seed 3583: 6e-14, 1.5e-12, 3.9e-11 on
    three successive updates of the reference) being one update ahead trips it.
    truth = (err_dev, err_ref) against the extended-precision trajectory (Spread.truth / fixture_truth): THE rule,
    err_dev <= max(base, 2 err_ref); `spread` is then only recorded ("K needed", the rounds-2/3 diagnostic).
    Without truth: the old rules (spread, else pivot)."""
    tol, piv, rule = tolerance(state, base, spread, truth)
    rec = WORST.setdefault(key, {"err": 0.0, "tol": tol, "pivot": piv, "checks": 0, "worst_well_conditioned": 0.0,
                                 "rule": rule, "k_needed": 0.0, "err_dev_exact": None, "err_ref_exact": None,
                                 "truth_ratio": 0.0})
    rec["checks"] += 1
    if piv > 0.5:
        rec["worst_well_conditioned"] = max(rec["worst_well_conditioned"], float(err))
    if spread and err > base:
        rec["k_needed"] = max(rec["k_needed"], float(err) / spread)
    judged = float(err) if truth is None else float(truth[0])
    if truth is not None:
        TOUCHED.add(key)
        rec["base"] = float(base)
        rec["err_dev_exact"] = max(rec["err_dev_exact"] or 0.0, float(truth[0]))       # worst over the sequence(s) so far
        rec["err_ref_exact"] = max(rec["err_ref_exact"] or 0.0, float(truth[1]))       # (cumulative in the caller already)
        if len(truth) > 2 and truth[2] is not None:
            rec["n"] = int(truth[2])
        if len(truth) > 3 and truth[3] is not None:
            rec["mvec"] = int(truth[3])
        rec["seq_dev"] = max(rec.get("seq_dev", 0.0), float(truth[0]))                 # this sequence (reset by finish())
        rec["seq_ref"] = max(rec.get("seq_ref", 0.0), float(truth[1]))
    if err >= rec["err"]:
        rec.update(err=float(err), pivot=float(piv))
    if judged >= rec.get("judged", -1.0):
        rec.update(judged=judged, tol=float(tol), rule=rule)
    if not stop and truth is not None and judged > tol:
        rec["stops"] = rec.get("stops", 0) + 1
        STOPS.append((key, where, judged, float(tol)))
        return err
    assert judged <= tol, (key, where, judged, float(tol), float(piv), rule, float(err))
    # THE DIRECT FIGURE: device against reference, the bar SURVEY.md 8(c) states, stays an ASSERTED bound wherever
    # the reference itself is within half the stated tolerance of the truth so far -- there ||f_dev - f_ref|| <= DIRECT_FACTOR x
    # base, no conditioning allowance of any kind.  (Where the reference's own sequential sums put it further from the truth
    # than the stated tolerance -- the three flavours still AGREE with one another there, they share the summation order: K in
    # the hundreds in the list-word tests -- a bound on dev-ref would measure the reference's error, and the truth rule
    # above is the bar.)  Counted per key: rec["direct_checks"].
    if truth is not None and float(truth[1]) <= 0.5 * base:
        rec["direct_checks"] = rec.get("direct_checks", 0) + 1
        rec["direct_worst"] = max(rec.get("direct_worst", 0.0), float(err))
        if stop:
            assert err <= DIRECT_FACTOR * base, (key, where, "device vs reference, direct", float(err), DIRECT_FACTOR * base)
    return err


def finish(keys=None, strict=True):
    """End of a call sequence (a test, a soak seed): THE rule for every key checked with `truth` since the last call --
    max err_dev <= max(base, TRUTH_FACTOR * max err_ref) over the sequence.  Records the share of the allowance used."""
    bad, hard_bad = [], []
    for key in sorted(TOUCHED if keys is None else keys):
        rec = WORST.get(key)
        if not rec or "seq_dev" not in rec:
            continue
        dev, ref, base = rec.pop("seq_dev"), rec.pop("seq_ref"), rec.get("base", 1e-12)
        fac = truth_factor(rec.get("n"), rec.get("mvec"))
        tol = max(base, fac * ref)
        if dev > base:          # how much of the allowance the device needed (1 = all of it)
            rec["truth_ratio"] = max(rec.get("truth_ratio", 0.0), dev / max(fac * ref, 1e-300))
        rec["tol"], rec["rule"] = float(tol), ("stated" if tol == base else f"{fac:g} x reference-vs-exact (per sequence)")
        if dev > tol:
            bad.append((key, dev, tol, ref))
        # THE HARD LINE of the numerical contract (include/nka_hip.h, item 3), per sequence: F = 8 beyond one tile, 32 with a base
        # of 1e-11 within -- exceeded by no record of any soak so far (profiles/r06/soak_paired.txt).  Asserted whatever `strict`
        # says: the soak tool records exceedances of the TYPICAL bar and goes on, but this one ends the run.
        hard = max(base * (10.0 if fac == TRUTH_FACTOR_TINY and base <= 1e-12 else 1.0), truth_hard(rec.get("n")) * ref)
        if dev > hard:
            hard_bad.append((key, dev, hard, ref))
            HARD_BEYOND.append((key, dev, hard, ref))
    if keys is None:
        TOUCHED.clear()
    else:
        TOUCHED.difference_update(keys)
    assert not hard_bad, ("device beyond the HARD line of the numerical contract (never exceeded before)", hard_bad)
    if strict:
        assert not bad, ("device further from the extended-precision trajectory than the rule allows", bad)
    return bad


def record(err, tol, key):
    """Record an error checked against a fixed tolerance (no conditioning rule)."""
    rec = WORST.setdefault(key, {"err": 0.0, "tol": tol, "pivot": None, "checks": 0, "worst_well_conditioned": 0.0,
                                 "rule": "fixed", "k_needed": 0.0})
    rec["checks"] += 1
    rec["worst_well_conditioned"] = max(rec["worst_well_conditioned"], float(err))
    if err >= rec["err"]:
        rec.update(err=float(err), tol=float(tol))
    assert err <= tol, (key, float(err), float(tol))
    return err


def dump_dir(root):
    """Where the worst-error record goes: $NKA_PARITY_OUT if set (os.devnull: nowhere); <root>/gpurun_out on a GPU box
    (gpurun exports GRAFT_REPO_ROOT there and merges that directory back); otherwise a directory under the system's temp
    dir -- a CPU run of the suite leaves the tree as it found it."""
    env = os.environ.get("NKA_PARITY_OUT")
    if env:
        return None if env == os.devnull else env
    if os.environ.get("GRAFT_REPO_ROOT"):
        return os.path.join(root, "gpurun_out")
    import tempfile
    return os.path.join(tempfile.gettempdir(), "nka_parity")


def dump(root):
    if not WORST:
        return None
    out = dump_dir(root)
    if out is None:
        return None
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_worst.json")
        old = {}
        if os.path.exists(path):
            try:
                with open(path) as fh:
                    old = json.load(fh)
            except Exception:
                old = {}
        old.update(WORST)
        with open(path, "w") as fh:
            json.dump(old, fh, indent=1, sort_keys=True)
        return path
    except OSError:
        return None
