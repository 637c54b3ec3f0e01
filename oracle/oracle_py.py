"""ctypes bindings for the CHECKERS: the C restatement (libnka_oracle.so) and,
where it has been built, the compiled reference (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (nka_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

F08, F08_VECTOR, C_FLAVOR = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
DOT_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, _dp, _dp)
ACCEL_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, C.c_int64)


def build(ref: bool | None = None) -> None:
    """Compile the restatement; and the reference too when /root/reference exists."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref is None:
        ref = os.path.isdir("/root/reference")
    if ref:
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("NKA_ORACLE_LIB") or os.path.join(HERE, "libnka_oracle.so")     # (NKA_ORACLE_LIB: the sanitizer
        if not os.path.exists(path):                                                          #  build, oracle/Makefile `asan`)
            if os.environ.get("NKA_ORACLE_LIB"):
                raise FileNotFoundError(path)
            build(ref=False)
        L = C.CDLL(path)
        L.nka_oracle_init.restype = C.c_void_p
        L.nka_oracle_init.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.nka_oracle_delete.argtypes = [C.c_void_p]
        L.nka_oracle_set_vec_tol.argtypes = [C.c_void_p, C.c_double]
        L.nka_oracle_set_dot_prod.argtypes = [C.c_void_p, DOT_FN, C.c_void_p]
        L.nka_oracle_accel_update.argtypes = [C.c_void_p, _dp]
        L.nka_oracle_restart.argtypes = [C.c_void_p]
        L.nka_oracle_relax.argtypes = [C.c_void_p]
        for name in ("num_vec", "max_vec", "defined"):
            getattr(L, "nka_oracle_" + name).argtypes = [C.c_void_p]
            getattr(L, "nka_oracle_" + name).restype = C.c_int
        L.nka_oracle_vec_len.argtypes = [C.c_void_p]
        L.nka_oracle_vec_len.restype = C.c_int64
        L.nka_oracle_vec_tol.argtypes = [C.c_void_p]
        L.nka_oracle_vec_tol.restype = C.c_double
        L.nka_oracle_get_state.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _dp, _dp]
        L.nka_oracle_w.argtypes = [C.c_void_p, C.c_int]
        L.nka_oracle_w.restype = _dp
        L.nka_oracle_v.argtypes = [C.c_void_p, C.c_int]
        L.nka_oracle_v.restype = _dp
        L.nka_oracle_scalar_step.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp, _dp, _ip]
        # the same restatement in extended precision (nka_oracle_exact.c) and the error-attribution dots (nka_oracle_probe.c)
        L.nka_oraclex_init.restype = C.c_void_p
        L.nka_oraclex_init.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.nka_oraclex_delete.argtypes = [C.c_void_p]
        L.nka_oraclex_set_vec_tol.argtypes = [C.c_void_p, C.c_double]
        L.nka_oraclex_accel_update.argtypes = [C.c_void_p, _dp]
        L.nka_oraclex_restart.argtypes = [C.c_void_p]
        L.nka_oraclex_relax.argtypes = [C.c_void_p]
        for name in ("num_vec", "max_vec", "defined"):
            getattr(L, "nka_oraclex_" + name).argtypes = [C.c_void_p]
            getattr(L, "nka_oraclex_" + name).restype = C.c_int
        L.nka_oraclex_vec_len.argtypes = [C.c_void_p]
        L.nka_oraclex_vec_len.restype = C.c_int64
        L.nka_oraclex_vec_tol.argtypes = [C.c_void_p]
        L.nka_oraclex_vec_tol.restype = C.c_double
        L.nka_oraclex_get_state.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _dp, _dp]
        L.nka_oracle_set_gram_from_raw_sums.argtypes = [C.c_void_p, C.c_int]
        L.nka_example_solve.restype = C.c_int
        L.nka_example_solve.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                        C.c_double, ACCEL_FN, C.c_void_p, _dp, _dp]
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


class State:
    """Snapshot of the list / factor state in the Fortran numbering."""

    def __init__(self, subspace, pending, first, last, free, next_, prev, h, c=None):
        self.subspace, self.pending = bool(subspace), bool(pending)
        self.first, self.last, self.free = int(first), int(last), int(free)
        self.next, self.prev, self.h, self.c = next_, prev, h, c

    def list_order(self):
        out, k = [], self.first
        while k != 0:
            out.append(k)
            k = int(self.next[k - 1])
        return out

    def free_order(self):
        out, k = [], self.free
        while k != 0:
            out.append(k)
            k = int(self.next[k - 1])
        return out


class OracleNKA:
    """The C restatement behind the reference's method names
    (src-F08/nka_type.F90:169-181)."""

    def __init__(self, vlen: int, mvec: int, flavor: int = F08):
        self._L = lib()
        self._h = self._L.nka_oracle_init(vlen, mvec, flavor)
        if not self._h:
            raise ValueError("nka_oracle_init failed")
        self.vlen, self.mvec = vlen, mvec
        self._cb = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nka_oracle_delete(self._h)
            self._h = None

    def set_vec_tol(self, vtol: float):
        self._L.nka_oracle_set_vec_tol(self._h, vtol)

    def set_dot_prod(self, fn):
        """fn(x: ndarray, y: ndarray) -> float : the global dot product hook."""
        def tramp(_ctx, n, x, y):
            return float(fn(np.ctypeslib.as_array(x, (n,)), np.ctypeslib.as_array(y, (n,))))
        self._cb = DOT_FN(tramp)
        self._L.nka_oracle_set_dot_prod(self._h, self._cb, None)

    def accel_update(self, f: np.ndarray):
        assert f.shape == (self.vlen,)
        self._L.nka_oracle_accel_update(self._h, _ptr(f))

    def restart(self):
        self._L.nka_oracle_restart(self._h)

    def relax(self):
        self._L.nka_oracle_relax(self._h)

    def num_vec(self):
        return self._L.nka_oracle_num_vec(self._h)

    def max_vec(self):
        return self._L.nka_oracle_max_vec(self._h)

    def vec_len(self):
        return self._L.nka_oracle_vec_len(self._h)

    def vec_tol(self):
        return self._L.nka_oracle_vec_tol(self._h)

    def defined(self):
        return bool(self._L.nka_oracle_defined(self._h))

    def state(self) -> State:
        n = self.mvec + 1
        ints = [C.c_int() for _ in range(5)]
        nxt = np.zeros(n, np.int32)
        prv = np.zeros(n, np.int32)
        h = np.zeros((n, n), np.float64)  # filled column-major -> transpose view below
        c = np.zeros(n, np.float64)
        self._L.nka_oracle_get_state(self._h, *[C.byref(i) for i in ints],
                                     nxt.ctypes.data_as(_ip), prv.ctypes.data_as(_ip), _ptr(h), _ptr(c))
        return State(ints[0].value, ints[1].value, ints[2].value, ints[3].value, ints[4].value,
                     nxt, prv, h.T.copy(), c)

    def w(self, slot: int) -> np.ndarray:
        return np.ctypeslib.as_array(self._L.nka_oracle_w(self._h, slot), (self.vlen,)).copy()

    def v(self, slot: int) -> np.ndarray:
        return np.ctypeslib.as_array(self._L.nka_oracle_v(self._h, slot), (self.vlen,)).copy()

    def scalar_step(self, s: float, hrow_by_slot: np.ndarray, b_by_slot: np.ndarray) -> int:
        """hrow/b are indexed by slot with a leading unused entry (length mvec+2)."""
        new = C.c_int()
        self._L.nka_oracle_scalar_step(self._h, 0, s, _ptr(hrow_by_slot), _ptr(b_by_slot), C.byref(new))
        return new.value


class OracleExact(OracleNKA):
    """The restatement in EXTENDED precision (oracle/nka_oracle_exact.c: nka_oracle.c compiled a second time with
    long double arithmetic -- same statements, same list logic).  The "exact" trajectory against which the parity
    tests measure both the reference's and the device's error.  No user dot product, no stored-vector access."""

    def __init__(self, vlen: int, mvec: int, flavor: int = F08):      # noqa: super().__init__ not called on purpose
        self._L = lib()
        self._h = self._L.nka_oraclex_init(vlen, mvec, flavor)
        if not self._h:
            raise ValueError("nka_oraclex_init failed")
        self.vlen, self.mvec = vlen, mvec
        self._cb = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nka_oraclex_delete(self._h)
            self._h = None

    def set_vec_tol(self, vtol: float):
        self._L.nka_oraclex_set_vec_tol(self._h, vtol)

    def set_dot_prod(self, fn):
        raise NotImplementedError("the extended-precision flavour sums in extended precision itself")

    def accel_update(self, f: np.ndarray):
        assert f.shape == (self.vlen,)
        self._L.nka_oraclex_accel_update(self._h, _ptr(f))

    def restart(self):
        self._L.nka_oraclex_restart(self._h)

    def relax(self):
        self._L.nka_oraclex_relax(self._h)

    def num_vec(self):
        return self._L.nka_oraclex_num_vec(self._h)

    def max_vec(self):
        return self._L.nka_oraclex_max_vec(self._h)

    def vec_len(self):
        return self._L.nka_oraclex_vec_len(self._h)

    def vec_tol(self):
        return self._L.nka_oraclex_vec_tol(self._h)

    def defined(self):
        return bool(self._L.nka_oraclex_defined(self._h))

    def state(self) -> State:
        n = self.mvec + 1
        ints = [C.c_int() for _ in range(5)]
        nxt = np.zeros(n, np.int32)
        prv = np.zeros(n, np.int32)
        h = np.zeros((n, n), np.float64)
        c = np.zeros(n, np.float64)
        self._L.nka_oraclex_get_state(self._h, *[C.byref(i) for i in ints],
                                      nxt.ctypes.data_as(_ip), prv.ctypes.data_as(_ip), _ptr(h), _ptr(c))
        return State(ints[0].value, ints[1].value, ints[2].value, ints[3].value, ints[4].value,
                     nxt, prv, h.T.copy(), c)

    def w(self, slot):
        raise NotImplementedError

    v = scalar_step = w


def attribution_oracle(vlen: int, mvec: int, flavor: int = F08, fma=False, blocked=False, raw_sums=False) -> OracleNKA:
    """The double restatement with the device path's deliberate deviations switched on one at a time
    (oracle/nka_oracle_probe.c): fused multiply-adds in the inner products, the device's blocked summation order,
    the Gram row as fl(<d,w_k>/s) from raw sums.  All three = the arithmetic of the device's inner products."""
    a = OracleNKA(vlen, mvec, flavor)
    L = a._L
    fn = {(False, False): None, (True, False): L.nka_oracle_dot_fma, (False, True): L.nka_oracle_dot_blocked,
          (True, True): L.nka_oracle_dot_device}[(bool(fma), bool(blocked))]
    if fn is not None:
        a._cb = C.cast(fn, DOT_FN)
        L.nka_oracle_set_dot_prod(a._h, a._cb, None)
    if raw_sums:
        L.nka_oracle_set_gram_from_raw_sums(a._h, 1)
    return a


def example_solve(nx=50, ny=50, a=0.02, nsweep=2, omega=1.4, accel=None, maxitr=999, tol=1e-6):
    """Run the BASELINE config-1 problem (oracle/nka_example_problem.c).
    accel: object with accel_update(ndarray) or None.  Returns (rnorms, u)."""
    L = lib()
    rn = np.zeros(maxitr + 1)
    u = np.zeros(nx * ny)

    def tramp(_ctx, r, n):
        accel.accel_update(np.ctypeslib.as_array(r, (n,)))

    cb = ACCEL_FN(tramp) if accel is not None else C.cast(None, ACCEL_FN)
    nit = L.nka_example_solve(nx, ny, a, nsweep, omega, maxitr, tol, cb, None, _ptr(rn), _ptr(u))
    return rn[: nit + 1].copy(), u


def format_example_line(itr: int, rnorm: float, rnorm0: float) -> str:
    """The reference's print format (src-F08/nka_example.F90:253):
    (i3,a,es14.6,es13.3,f8.3)."""
    red = rnorm / rnorm0
    rate = red ** (1.0 / itr)
    return f"{itr:3d}:{rnorm:14.6E}{red:13.3E}{rate:8.3f}"


# --------------------------------------------------------------------------
# The compiled reference (exists only where oracle/Makefile `ref` has run).
# --------------------------------------------------------------------------

def have_ref() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, f))
               for f in ("libnka_ref_c.so", "libnka_ref_f08.so", "libnka_ref_f08vec.so"))


class RefF08:
    """The reference src-F08 module itself, through oracle/ref_f08_shim.F90."""
    _L = None

    def __init__(self, vlen: int, mvec: int):
        if RefF08._L is None:
            L = C.CDLL(os.path.join(REF_DIR, "libnka_ref_f08.so"))
            L.ref_f08_init.argtypes = [C.c_int, C.c_int]
            L.ref_f08_set_vec_tol.argtypes = [C.c_int, C.c_double]
            L.ref_f08_accel_update.argtypes = [C.c_int, _dp, C.c_int]
            L.ref_f08_vec_tol.restype = C.c_double
            RefF08._L = L
        self.vlen, self.mvec = vlen, mvec
        self._h = RefF08._L.ref_f08_init(vlen, mvec)
        if self._h < 1:
            raise RuntimeError("reference handle pool exhausted")

    def __del__(self):
        if getattr(self, "_h", 0) >= 1:
            RefF08._L.ref_f08_delete(self._h)
            self._h = 0

    def set_vec_tol(self, vtol):
        RefF08._L.ref_f08_set_vec_tol(self._h, vtol)

    def accel_update(self, f):
        RefF08._L.ref_f08_accel_update(self._h, _ptr(f), f.size)

    def restart(self):
        RefF08._L.ref_f08_restart(self._h)

    def relax(self):
        RefF08._L.ref_f08_relax(self._h)

    def num_vec(self):
        return RefF08._L.ref_f08_num_vec(self._h)

    def max_vec(self):
        return RefF08._L.ref_f08_max_vec(self._h)

    def vec_len(self):
        return RefF08._L.ref_f08_vec_len(self._h)

    def vec_tol(self):
        return RefF08._L.ref_f08_vec_tol(self._h)

    def defined(self):
        return bool(RefF08._L.ref_f08_defined(self._h))


class RefF08Vector:
    """The reference src-F08-vector module on its own grid_vector
    (oracle/ref_f08vec_shim.F90).  f is the full (nx+2)*(ny+2) array."""
    _L = None

    def __init__(self, nx: int, ny: int, mvec: int):
        if RefF08Vector._L is None:
            L = C.CDLL(os.path.join(REF_DIR, "libnka_ref_f08vec.so"))
            L.ref_f08vec_init.argtypes = [C.c_int, C.c_int, C.c_int]
            L.ref_f08vec_set_vec_tol.argtypes = [C.c_int, C.c_double]
            L.ref_f08vec_accel_update.argtypes = [C.c_int, _dp, C.c_int]
            RefF08Vector._L = L
        self.nx, self.ny, self.mvec = nx, ny, mvec
        self._h = RefF08Vector._L.ref_f08vec_init(nx, ny, mvec)
        if self._h < 1:
            raise RuntimeError("reference handle pool exhausted")

    def __del__(self):
        if getattr(self, "_h", 0) >= 1:
            RefF08Vector._L.ref_f08vec_delete(self._h)
            self._h = 0

    def set_vec_tol(self, vtol):
        RefF08Vector._L.ref_f08vec_set_vec_tol(self._h, vtol)

    def accel_update(self, f):
        RefF08Vector._L.ref_f08vec_accel_update(self._h, _ptr(f), f.size)

    def restart(self):
        RefF08Vector._L.ref_f08vec_restart(self._h)

    def relax(self):
        RefF08Vector._L.ref_f08vec_relax(self._h)

    def num_vec(self):
        return RefF08Vector._L.ref_f08vec_num_vec(self._h)


class RefC:
    """The reference src-C accelerator (oracle/ref_c_shim.c) with state dump."""
    _L = None

    def __init__(self, vlen: int, mvec: int, vtol: float = 0.01, dp=None):
        """dp(x: ndarray, y: ndarray) -> float: the reference's own user dot product argument
        (nka_init(..., dp), .c:211, 227-231), None = its default."""
        if RefC._L is None:
            L = C.CDLL(os.path.join(REF_DIR, "libnka_ref_c.so"))
            L.nka_init.restype = C.c_void_p
            L.nka_init.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p]
            for nm in ("nka_delete", "nka_restart", "nka_relax"):
                getattr(L, nm).argtypes = [C.c_void_p]
            L.nka_accel_update.argtypes = [C.c_void_p, _dp]
            L.nka_num_vec.argtypes = [C.c_void_p]
            L.ref_c_get_state.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _dp]
            L.ref_c_set_vec_tol.argtypes = [C.c_void_p, C.c_double]
            L.ref_c_w.argtypes = [C.c_void_p, C.c_int]
            L.ref_c_w.restype = _dp
            L.ref_c_v.argtypes = [C.c_void_p, C.c_int]
            L.ref_c_v.restype = _dp
            RefC._L = L
        self.vlen, self.mvec = vlen, mvec
        self._cb = None
        if dp is not None:
            def tramp(n, x, y):
                return float(dp(np.ctypeslib.as_array(x, (n,)), np.ctypeslib.as_array(y, (n,))))
            self._cb = C.CFUNCTYPE(C.c_double, C.c_int, _dp, _dp)(tramp)
        self._h = RefC._L.nka_init(vlen, mvec, vtol, C.cast(self._cb, C.c_void_p) if self._cb else None)

    def __del__(self):
        if getattr(self, "_h", None):
            RefC._L.nka_delete(self._h)
            self._h = None

    def set_vec_tol(self, vtol: float):
        """Not in the C API (vtol is a constructor argument, .c:211): writes the field
        the drop test reads (.c:362) -- see oracle/ref_c_shim.c."""
        RefC._L.ref_c_set_vec_tol(self._h, float(vtol))

    def accel_update(self, f):
        RefC._L.nka_accel_update(self._h, _ptr(f))

    def restart(self):
        RefC._L.nka_restart(self._h)

    def relax(self):
        RefC._L.nka_relax(self._h)

    def num_vec(self):
        return RefC._L.nka_num_vec(self._h)

    def state(self) -> State:
        n = self.mvec + 1
        ints = [C.c_int() for _ in range(5)]
        nxt = np.zeros(n, np.int32)
        prv = np.zeros(n, np.int32)
        h = np.zeros((n, n), np.float64)
        RefC._L.ref_c_get_state(self._h, *[C.byref(i) for i in ints],
                                nxt.ctypes.data_as(_ip), prv.ctypes.data_as(_ip), _ptr(h))
        return State(ints[0].value, ints[1].value, ints[2].value, ints[3].value, ints[4].value,
                     nxt, prv, h.T.copy())

    def w(self, slot):
        return np.ctypeslib.as_array(RefC._L.ref_c_w(self._h, slot), (self.vlen,)).copy()

    def v(self, slot):
        return np.ctypeslib.as_array(RefC._L.ref_c_v(self._h, slot), (self.vlen,)).copy()


def lcg_vectors(count: int, n: int, seed: int = 1) -> np.ndarray:
    """SURVEY.md 8(c) generator: x <- (1103515245 x + 12345) mod 2^31,
    value x / 2^30 - 1 (exact in binary64 in every language)."""
    out = np.empty((count, n))
    x = seed
    for t in range(count):
        for i in range(n):
            x = (1103515245 * x + 12345) % (1 << 31)
            out[t, i] = x / float(1 << 30) - 1.0
    return out
