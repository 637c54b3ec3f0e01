"""The reference-order sums of LONG vectors (k_chain_sums, round 5): one workgroup per sum, whole blocks of 1024 products
taken through the chain in integer arithmetic wherever that is provably the walk's result (chain_block_summary /
chain_block_apply), the walk itself elsewhere.  A sequential sum  ((start + p0) + p1) + ...  has exactly one right answer per input, so every
comparison here is on the BITS: the kernel against numpy's strictly sequential `add.accumulate`, and against its own
element-after-element walk (`walk=True`), on inputs built to sit on the algorithm's edges -- halfway cases under both
parities, binade crossings in both directions, cancellation to zero, sign changes, products far larger and far smaller
than the sum, zeros, subnormals, overflow, NaN.  (The update's own use of the kernel is held to the compiled reference's
bits by tests/test_hip_reference_order.py and tests/test_hip_fullsize.py.)"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench():
    import torch
    import nka_amd
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    torch.cuda.set_device(0)
    acc = nka_amd.nka(diagnostic=True).init(64, 3)

    def run(x, y, start=0.0):
        tx, ty = torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(np.ascontiguousarray(y)).cuda()
        fast, _ = acc.debug_chain_sum(tx, ty, start, walk=False)
        walk, _ = acc.debug_chain_sum(tx, ty, start, walk=True)
        many = acc.debug_chain_sum(tx, ty, start, many=True)[0] if len(x) >= 1024 else fast   # (the whole-device form needs a full block)
        with np.errstate(all="ignore"):
            want = float(np.add.accumulate(np.concatenate([[start], x * y]))[-1]) if len(x) else float(start)
        return fast, walk, want, many

    return run


def bits(v):
    return struct.pack("<d", v)


def same(a, b):
    return bits(a) == bits(b) or (a != a and b != b)       # (any NaN for a NaN: payloads are not part of the contract)


def check(bench, x, y, start=0.0, what=""):
    fast, walk, want, many = bench(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), float(start))
    assert same(walk, want), (what, "walk", walk.hex() if walk == walk else walk, want.hex() if want == want else want)
    assert same(fast, want), (what, "blocks", fast.hex() if fast == fast else fast, want.hex() if want == want else want)
    assert same(many, want), (what, "whole device", many.hex() if many == many else many, want.hex() if want == want else want)


LENGTHS = [0, 1, 7, 8, 9, 511, 512, 513, 1024, 2047, 2048, 2049, 4096 + 37, 10 * 2048, 100_003]


@pytest.mark.parametrize("n", LENGTHS)
def test_random_signed_products_at_every_length(bench, n):
    rng = np.random.default_rng(n)
    check(bench, rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), 0.0, f"uniform n={n}")
    check(bench, rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), -3.75, f"uniform from -3.75 n={n}")
    check(bench, rng.standard_normal(n), rng.standard_normal(n), 1e6, f"normal from 1e6 n={n}")


@pytest.mark.parametrize("n", [2048 * 5, 300_001])
def test_sums_of_squares_and_of_negatives(bench, n):
    rng = np.random.default_rng(7 * n)
    d = rng.uniform(-1, 1, n)
    check(bench, d, d, 0.0, "norm")
    check(bench, d, -d, 0.0, "minus norm")
    check(bench, d * 1e-150, d * 1e-150, 0.0, "norm near the bottom of the range")
    check(bench, d * 1e150, d * 1e150, 0.0, "norm near the top of the range")


@pytest.mark.parametrize("scale", [1.0, 2.0 ** -40, 2.0 ** 300, 2.0 ** -600])
@pytest.mark.parametrize("start_int", [2 ** 52, 2 ** 52 + 1, 2 ** 52 + 100_000, 2 ** 53 - 5_000, 2 ** 53 - 20_001, 3 * 2 ** 51 + 1])
def test_halfway_cases_under_both_parities_and_across_the_binade_ends(bench, start_int, scale):
    """Sum ~ 2^52 (one unit in the last place = 1), products = multiples of one half: a halfway case at almost every
    addition, decided by the parity of the running sum; starts next to both ends of the binade walk out of it."""
    rng = np.random.default_rng(start_int % 1000 + int(abs(np.log2(scale))))
    for n, lo, hi in ((5000, -7, 8), (5000, 0, 6), (5000, -6, 1), (40_000, -3, 4), (3000, -2001, 2002)):
        halves = rng.integers(lo, hi, n).astype(np.float64) * 0.5
        for sign in (1.0, -1.0):
            check(bench, halves * scale, np.full(n, sign), sign * start_int * scale, f"halves [{lo},{hi}) from {sign}*{start_int}*{scale}")


def test_constant_half_ulp_products(bench):
    """1 + 2^-53 + 2^-53 + ...: every addition is a halfway case; from an even significand the sum never moves, from an
    odd one it moves once.  Likewise 1.5 and 2.5 units."""
    n = 6000
    ones = np.ones(n)
    for start in (1.0, 1.0 + 2.0 ** -52, 1.0 + 2.0 ** -51, 2.0 - 2.0 ** -52, -1.0, -(1.0 + 2.0 ** -52)):
        for k in (1, 3, 5, -1, -3):
            check(bench, np.full(n, k * 2.0 ** -53), ones, start, f"{k} half units from {start.hex()}")
    rng = np.random.default_rng(5)
    mix = rng.choice([2.0 ** -53, -(2.0 ** -53), 3 * 2.0 ** -53, 2.0 ** -54, 2.0 ** -52, 0.0, 2.0 ** -60], 50_000)
    for start in (1.0, 1.0 + 2.0 ** -52, 1.75):
        check(bench, mix, np.ones(len(mix)), start, f"mixed sub-unit products from {start.hex()}")


def test_cancellation_zero_sums_and_sign_changes(bench):
    rng = np.random.default_rng(11)
    n = 2048 * 6 + 5
    a = rng.uniform(-1, 1, n)
    pairs = np.empty(2 * n)
    pairs[0::2], pairs[1::2] = a, -a                          # returns to exactly zero after every pair
    check(bench, pairs, np.ones(2 * n), 0.0, "pairs")
    check(bench, pairs, np.ones(2 * n), 1e-300, "pairs on a tiny start")
    saw = np.where(np.arange(n) % 1024 < 512, 1.0, -1.0) * rng.uniform(0.5, 1, n)      # long climbs and descents through zero
    check(bench, saw, np.ones(n), 0.0, "saw")
    check(bench, np.zeros(n), a, 0.0, "zeros")
    check(bench, np.zeros(n), a, -0.0, "zeros from -0")
    check(bench, np.zeros(n), a, 5.0, "zeros from 5")


def test_magnitudes_over_the_whole_range(bench):
    rng = np.random.default_rng(13)
    n = 60_000
    x = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-12, 12, n)
    check(bench, x, rng.uniform(0.5, 2, n), 0.0, "24 decades")
    x = rng.choice([-1.0, 1.0], n) * 2.0 ** rng.integers(-1074, -1000, n).astype(np.float64)
    check(bench, x, np.ones(n), 0.0, "subnormals")
    check(bench, x, np.ones(n), 2.0 ** -1022, "subnormals on the smallest normal")
    big = np.full(n, 1e306)
    check(bench, big, np.full(n, 10.0), 0.0, "overflow to +Inf")
    check(bench, big, np.where(np.arange(n) < n // 2, 10.0, -10.0), 0.0, "Inf - Inf")
    y = rng.uniform(-1, 1, n)
    y[n // 3] = np.nan
    check(bench, rng.uniform(-1, 1, n), y, 1.0, "a NaN on the way")
    rare = np.where(rng.random(n) < 0.001, 1e12, 1.0) * rng.uniform(-1, 1, n)          # a giant among small ones now and then
    check(bench, rare, np.ones(n), 0.0, "rare giants")


@pytest.mark.parametrize("seed", range(6))
def test_giants_that_turn_the_sign_inside_a_block(bench, seed):
    """A product larger than the running sum itself, of the other sign, somewhere inside a block: the sum changes sign and
    exponent in one step, the prefix sums at that lane are beyond 2^53 (rounded), the block must not go in as summarised --
    and the blocks behind it were summarised under a sign that no longer holds.  (This case caught a real bug in a
    lane-by-lane variant of the fallback: the exclusive prefix AT the lane of the giant is not exact.)"""
    rng = np.random.default_rng(500 + seed)
    n = 1024 * 6
    p = rng.uniform(-1, 1, n)
    for pos in rng.integers(0, n, 5):
        p[pos] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(11.5, 12.5)
    for start in (-5.2e11, 5.2e11, 3e11, -7.7e11, 1e12):
        check(bench, p, np.ones(n), start, f"giants seed {seed} from {start}")
    check(bench, p * 2.0 ** -300, np.ones(n), -5.2e11 * 2.0 ** -300, f"giants seed {seed}, scaled down")


@pytest.mark.parametrize("seed", range(12))
def test_random_regimes_stitched_together(bench, seed):
    """Segments of random length, each with its own magnitude, sign bias and grid (multiples of a power of two: halfway
    cases by the thousand), stitched into one sum."""
    rng = np.random.default_rng(1000 + seed)
    parts = []
    for _ in range(rng.integers(3, 12)):
        n = int(rng.integers(1, 9000))
        mag = 2.0 ** rng.integers(-60, 60)
        kind = rng.integers(0, 4)
        if kind == 0:
            p = rng.uniform(-1, 1, n) * mag
        elif kind == 1:
            p = rng.integers(-5, 6, n) * mag * 2.0 ** -int(rng.integers(0, 56))
        elif kind == 2:
            p = np.abs(rng.standard_normal(n)) * mag * rng.choice([-1.0, 1.0])
        else:
            p = rng.integers(-3, 4, n) * 0.5 * mag
        parts.append(p)
    p = np.concatenate(parts)
    check(bench, p, np.ones(len(p)), float(rng.choice([0.0, 1.0, -1.0, 2.0 ** 52, -(2.0 ** 30)])), f"stitched seed {seed}")
