"""The random call sequences of the soak tool (tools/fuzz_gpu.py), as data: one generator that both the GPU soak
run and the CPU error attribution (tools/error_attribution.py) consume, so that "fuzz seed 4" names the same
sequence everywhere."""
import numpy as np

SIZES = [1, 2, 3, 7, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4097, 8191, 8193,
         16385, 32767, 65537, 70001]


def array_shape(seed, hostdot=False):
    """-> (rng, n, mvec, flavor) of an array-flavour seed; the rng is positioned behind the shape draws."""
    rng = np.random.default_rng(seed + (50_000 if hostdot else 0))
    n = int(rng.choice(SIZES)) if rng.random() < 0.8 else int(rng.integers(1, 70001))
    if hostdot:
        n = min(n, 8193)                                   # (2 + L vectors cross PCIe per update on this path)
    m = int(rng.integers(1, 41))
    flavor = int(rng.integers(0, 3))
    return rng, n, m, flavor


def array_ops(rng, n, steps=120):
    """Yields ("update", x) | ("relax",) | ("restart",) | ("set_vec_tol", v) | ("copy",): 80 % updates (55 % fresh,
    30 % in a 3-dimensional span, 10 % a repeat of the previous input, 5 % zero), relax, restart, set_vec_tol, deep copy."""
    basis = rng.standard_normal((3, n))
    prev = rng.standard_normal(n)
    for _ in range(steps):
        r = rng.random()
        if r < 0.80:
            kind = rng.random()
            if kind < 0.55:
                x = rng.standard_normal(n)
            elif kind < 0.85:
                x = rng.standard_normal(3) @ basis
            elif kind < 0.95:
                x = prev.copy()
            else:
                x = np.zeros(n)
            prev = x
            yield ("update", x)
        elif r < 0.87:
            yield ("relax",)
        elif r < 0.91:
            yield ("restart",)
        elif r < 0.96:
            yield ("set_vec_tol", float(10.0 ** rng.uniform(-3, -0.3)))
        else:
            yield ("copy",)
