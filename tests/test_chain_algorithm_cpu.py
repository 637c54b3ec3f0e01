"""The arithmetic behind the reference-order sums of long vectors (nka_amd/csrc/nka_kernels.hpp: chain_lane_pass,
chain_block_summary, chain_block_apply), restated in numpy and held to numpy's strictly sequential `add.accumulate` ON THE
BITS -- no GPU needed.  It is the executable form of the argument in docs/design/11_chain_sums.md:

  while 2^e <= |a| < 2^(e+1), fl(a + p) = u (S + R(p/u)),  u = 2^(e-52),  S = |a|/u an integer in [2^52, 2^53),
  R = rounding to an integer with halves going to whichever neighbour makes S + R even;

so a block of products reduces to a SUMMARY that depends on the running sum only through its sign and exponent -- the sum of
the rounded increments, the least / greatest prefix, and the corrections (+-1) the halfway cases owe under either parity of
the sum the block starts from -- and a block whose prefixes provably stay inside the binade goes in with one addition.
The device code does the same per lane of 16 products with ballots and scans; tests/test_chain_sums_gpu.py holds IT to the
same sequential sums.  (Test infrastructure: nothing under nka_amd/ imports this.)"""
import numpy as np
import pytest

M = 6755399441055744.0            # 1.5 * 2^52
EDGE = 2.0 ** 34
BLOCK = 1024


def sequential(start, p):
    with np.errstate(all="ignore"):
        return float(np.add.accumulate(np.concatenate([[start], p]))[-1])


def scale_of(a):
    """(S, scale, unscale) of a running sum the integer form applies to, else None (zero, subnormal, Inf, NaN, far exponents)."""
    if a == 0 or not np.isfinite(a):
        return None
    e = int(np.frexp(abs(a))[1]) - 1
    if e < -900 or e > 900:
        return None
    sgn = 1.0 if a > 0 else -1.0
    return a * sgn * 2.0 ** (52 - e), sgn * 2.0 ** (52 - e), sgn * 2.0 ** (e - 52)


def summarise(p, scale):
    """The summary of a block under `scale` (sign and exponent of the sum): total, least / greatest prefix, the corrections
    if the block starts even / odd, and whether every |t| allows exact integer arithmetic."""
    with np.errstate(all="ignore"):
        t = p * scale
        tm = t + M
        r = tm - M
        diff = t - r
    if not np.all(np.abs(r).reshape(-1, 16).sum(axis=1) < 2.0 ** 51):      # (NaN and Inf fail the comparison)
        return None
    halfway = np.abs(diff) == 0.5
    tau = np.where(diff > 0, 1, -1)
    rint = r.astype(np.int64)
    adj = []
    for start_parity in (0, 1):                       # the parity chain, once per parity of the sum the block starts from
        par, corr = start_parity, 0
        for i in range(len(p)):
            if halfway[i]:
                if par:                               # r is the even neighbour of t: an odd sum takes the other one
                    corr += int(tau[i])
                par = 0                               # ... and is even afterwards either way
            else:
                par ^= int(rint[i]) & 1
        adj.append(corr)
    prefix = np.cumsum(r)
    return float(prefix[-1]), float(min(0.0, prefix.min())), float(max(0.0, prefix.max())), adj[0], adj[1]


def chain(start, p):
    """The whole sum: blocks go in by their summaries where that is provably the sequential result, are walked elsewhere."""
    a, taken = float(start), 0
    for b0 in range(0, len(p) - len(p) % BLOCK, BLOCK):
        blk = p[b0:b0 + BLOCK]
        sc = scale_of(a)
        sm = summarise(blk, sc[1]) if sc else None
        if sm is not None:
            S, _, unscale = sc
            total, lo, hi, adj_even, adj_odd = sm
            if S + lo >= 2.0 ** 52 + EDGE and S + hi <= 2.0 ** 53 - EDGE:
                a = (S + (total + (adj_odd if int(S) & 1 else adj_even))) * unscale
                taken += 1
                continue
        a = sequential(a, blk)
    rest = p[len(p) - len(p) % BLOCK:]
    return (sequential(a, rest) if len(rest) else a), taken


def same(x, y):
    return np.float64(x).tobytes() == np.float64(y).tobytes() or (x != x and y != y)


def check(p, start, must_take=0, what=""):
    p = np.asarray(p, dtype=np.float64)
    got, taken = chain(start, p)
    want = sequential(start, p)
    assert same(got, want), (what, float(got).hex(), float(want).hex())
    assert taken >= must_take, (what, taken)


def test_random_walks_and_monotone_sums():
    rng = np.random.default_rng(1)
    for n in (1024, 5000, 40_000):
        x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        check(x * y, 0.0, what="signed")
        check(x * y, 1e6, must_take=n // BLOCK, what="signed on a large start")       # (far from every binade end: all blocks go in)
        check(x * x, 0.0, must_take=n // BLOCK - 6, what="squares")
        check(-(x * x), -3.0, what="negative squares")


@pytest.mark.parametrize("scale", [1.0, 2.0 ** -40, 2.0 ** 300])
def test_halfway_cases_under_both_parities_and_across_the_binade_ends(scale):
    rng = np.random.default_rng(2)
    for start_int in (2 ** 52, 2 ** 52 + 1, 2 ** 52 + 100_001, 2 ** 53 - 20_001, 3 * 2 ** 51 + 1):
        for lo, hi in ((-7, 8), (0, 6), (-6, 1)):
            halves = rng.integers(lo, hi, 4096).astype(np.float64) * 0.5
            for sign in (1.0, -1.0):
                check(halves * scale * sign, sign * start_int * scale, what=f"halves [{lo},{hi}) from {sign}*{start_int}*{scale}")
    # every addition a halfway case: from an even significand the sum never moves, from an odd one it moves once
    for start in (1.0, 1.0 + 2.0 ** -52, -1.0, -(1.0 + 2.0 ** -52), 1.5, 1.5 + 2.0 ** -52, -1.5, -(1.5 + 2.0 ** -52)):
        for k in (1, 3, -1, -3):                       # (from 1.0 the sum sits on an end of its binade: walked; from 1.5 the blocks go in)
            check(np.full(3000, k * 2.0 ** -53), start, must_take=2 if abs(start) >= 1.5 else 0, what=f"{k} half units from {start}")


def test_cancellation_giants_and_values_the_integer_form_must_refuse():
    rng = np.random.default_rng(3)
    n = 6 * BLOCK
    a = rng.uniform(-1, 1, n)
    pairs = np.empty(2 * n)
    pairs[0::2], pairs[1::2] = a, -a
    check(pairs, 0.0, what="pairs that return to zero")
    p = rng.uniform(-1, 1, n)
    for pos in rng.integers(0, n, 5):
        p[pos] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(11.5, 12.5)
    for start in (-5.2e11, 5.2e11, 3e11):
        check(p, start, what=f"giants that turn the sign, from {start}")
    check(rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-12, 12, n), 0.0, what="24 decades")
    check(np.full(n, 1e306) * 10.0, 0.0, what="overflow")
    q = rng.uniform(-1, 1, n)
    q[n // 3] = np.nan
    check(q, 1.0, what="a NaN on the way")
    check(rng.choice([-1.0, 1.0], n) * 2.0 ** rng.integers(-1074, -1000, n).astype(np.float64), 2.0 ** -1022, what="subnormals")
