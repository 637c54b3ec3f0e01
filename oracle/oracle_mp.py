"""oracle/oracle_mp.py -- TEST INFRASTRUCTURE ONLY: the accelerator in 60-digit arithmetic (mpmath), written a third time.

Why it exists: the parity rule of tests/parity_util.py measures device and reference against an
"exact" trajectory, and that trajectory came from oracle/nka_oracle_exact.c alone -- the builder's own restatement compiled
in extended precision (unit roundoff 5.4e-20).  This module is an INDEPENDENT check of that truth: the same algorithm, read
again from the reference and written differently (Python lists for the ordered subspace instead of linked slots in arrays,
a dictionary for the Gram / Cholesky entries, no shared code), in arithmetic 40 digits finer.  tests/test_oracle_exact_cpu.py
holds the extended-precision run to it on the golden scenarios and on random sequences: same decisions, outputs equal to
far below the distances the rule measures.  Pure Python: small cases only.

Follows /root/reference/src-F08/nka_type.F90 (F08 flavour; in exact arithmetic the three flavours coincide):
  accel_update :249-419   relax :439-457   restart :422-436   num_vec :221-231
Nothing under nka_amd/ imports this file.
"""
from __future__ import annotations

import mpmath as mp

DPS = 60


class MpNKA:
    """Same call surface as the other oracles: accel_update(f: numpy float64 array, in place), relax, restart,
    set_vec_tol, num_vec, list_order (slot numbers as the reference assigns them)."""

    def __init__(self, vlen: int, mvec: int, vtol: float = 0.01):
        assert vlen >= 0 and mvec > 0 and vtol > 0          # the reference's ASSERTs, :190-191, :205
        self.n, self.mvec = int(vlen), int(mvec)
        self.vtol = mp.mpf(vtol)
        self.w, self.v = {}, {}                              # slot -> list of mpf
        self.g = {}                                          # (slot_a, slot_b): raw Gram entry or factor entry, as in h(:,:)
        self.restart()

    # :422-436 -- the subspace is flushed; the free slots are handed out 1, 2, 3, ... again
    def restart(self):
        self.order = []                                      # slots of the subspace, newest first
        self.free = list(range(1, self.mvec + 2))            # next slot to hand out first (a stack: returned slots go on top)
        self.subspace = False
        self.pending = False

    # :439-457
    def relax(self):
        if self.pending:
            self.free.insert(0, self.order.pop(0))
            self.pending = False

    def set_vec_tol(self, vtol: float):
        assert vtol > 0
        self.vtol = mp.mpf(vtol)

    # :221-231
    def num_vec(self) -> int:
        return len(self.order) - (1 if self.pending else 0)

    def list_order(self):
        return list(self.order)

    @staticmethod
    def _dot(x, y):
        return mp.fsum(a * b for a, b in zip(x, y))          # (the default dp: dot_product, :216-219)

    def _factor(self):
        """:295-351 -- Cholesky of the Gram matrix row by row in list order; the last entry goes when the subspace is
        full (:301-309), an entry whose pivot is <= vtol^2 goes as (nearly) dependent (:326-345)."""
        g, order = self.g, self.order
        first = order[0]
        g[(first, first)] = mp.mpf(1)
        pos, kept = 1, 1
        while pos < len(order):
            k = order[pos]
            kept += 1
            if kept > self.mvec:                             # capacity: k is necessarily the last entry
                assert pos == len(order) - 1
                order.pop()
                self.free.insert(0, k)
                break
            hkk = mp.mpf(1)
            for jp in range(pos):
                j = order[jp]
                hkj = g[(j, k)]
                for ip in range(jp):
                    i = order[ip]
                    hkj -= g[(k, i)] * g[(j, i)]
                hkj /= g[(j, j)]
                hkk -= hkj * hkj
                g[(k, j)] = hkj
            if hkk > self.vtol * self.vtol:
                g[(k, k)] = mp.sqrt(hkk)
                pos += 1
            else:                                            # dependent: unlink; the scan goes on behind its predecessor
                order.pop(pos)
                self.free.insert(0, k)
                kept -= 1
        self.subspace = True
        self.pending = False

    def accel_update(self, f_io):
        with mp.workdps(DPS):
            f = [mp.mpf(float(x)) for x in f_io]             # the caller's doubles, exactly
            if self.pending:                                 # :263-275
                first = self.order[0]
                w1 = [a - b for a, b in zip(self.w[first], f)]
                self.w[first] = w1
                s = mp.sqrt(self._dot(w1, w1))
                if s == 0:
                    self.relax()
            if self.pending:                                 # :282-347
                first = self.order[0]
                self.v[first] = [a / s for a in self.v[first]]
                self.w[first] = [a / s for a in self.w[first]]
                for k in self.order[1:]:
                    self.g[(first, k)] = self._dot(self.w[first], self.w[k])
                self._factor()
            slot = self.free.pop(0)                          # :357-361
            self.w[slot] = list(f)
            if self.subspace:                                # :366-399
                c = {j: self._dot(f, self.w[j]) for j in self.order}
                for jp, j in enumerate(self.order):          # forward substitution, first -> last
                    cj = c[j]
                    for i in self.order[:jp]:
                        cj -= self.g[(j, i)] * c[i]
                    c[j] = cj / self.g[(j, j)]
                rev = self.order[::-1]
                for jp, j in enumerate(rev):                 # backward substitution, last -> first
                    cj = c[j]
                    for i in rev[:jp]:
                        cj -= self.g[(i, j)] * c[i]
                    c[j] = cj / self.g[(j, j)]
                for k in self.order:
                    ck, wk, vk = c[k], self.w[k], self.v[k]
                    f = [(x - ck * a) + ck * b for x, a, b in zip(f, wk, vk)]
            self.v[slot] = list(f)                           # :404
            self.order.insert(0, slot)                       # :406-417
            self.pending = True
            for i, x in enumerate(f):
                f_io[i] = float(x)
            self.last_exact = f                              # the unrounded result, for comparisons finer than double
