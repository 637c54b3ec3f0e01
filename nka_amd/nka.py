"""Python mirror of the reference's accelerator object.

Same method names, argument meaning and call-order rules as `type nka` of
/root/reference/src-F08/nka_type.F90:154-181 (init, set_vec_tol, set_dot_prod,
accel_update, relax, restart, num_vec, max_vec, vec_len, vec_tol, defined), so
the parity tests read like the reference's usage.  All arithmetic happens in
libnka_hip.so on the GPU; this file is plumbing (ctypes + torch for device
memory and streams).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

FLAVOR_F08, FLAVOR_F08_VECTOR, FLAVOR_C = 0, 1, 2
FLAVOR_DEFAULT = -1     # resolved by the library: NKA_HIP_FLAVOR, else compact storage (include/nka_hip.h)
SUMS_AUTO, SUMS_REFERENCE_ORDER, SUMS_BLOCKED, SUMS_BLOCKED_ROUNDED = 0, 1, 2, 3     # nka_hip_set_sum_order (include/nka_hip.h)


class NKAError(RuntimeError):
    pass


def _check(rc: int, what: str, L=None):
    if rc != 0:
        msg = (L or _lib.load()).nka_hip_last_error()        # (the error text lives in the library that failed)
        raise NKAError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class State:
    """Snapshot of the private list state (Fortran numbering; 0 = end of list)."""

    def __init__(self, subspace, pending, first, last, free, next_, prev, h, c):
        self.subspace, self.pending = bool(subspace), bool(pending)
        self.first, self.last, self.free = int(first), int(last), int(free)
        self.next, self.prev, self.h, self.c = next_, prev, h, c

    def list_order(self):
        out, k = [], self.first
        while k != 0 and len(out) <= len(self.next):
            out.append(k)
            k = int(self.next[k - 1])
        return out

    def free_order(self):
        out, k = [], self.free
        while k != 0 and len(out) <= len(self.next):
            out.append(k)
            k = int(self.next[k - 1])
        return out


class nka:  # noqa: N801  (the reference's type name)
    """MI355X accelerator object; see module docstring."""

    def __init__(self, diagnostic: bool = False, lib: str | None = None):
        """diagnostic=True: an object of the diagnostic build (libnka_hip_diag.so: the product plus set_tuning /
        set_grid, include/nka_hip_diag.h) -- for the tests that hold every kernel variant to the same bits and for
        the A/B tools."""
        self._h = None
        self._diag = bool(diagnostic)
        self._libpath = lib
        if lib is not None:            # another build of the diagnostic ABI (A/B of compile-time variants, tools/ab_libs.py)
            self._diag = True
            self._L = _lib.load_diag_at(lib)
        else:
            self._L = _lib.load_diag() if diagnostic else _lib.load()  # raises if the HIP library is missing: no CPU path
        self._cb = None
        self._hd = None

    # -- call a%init(vlen, mvec)                      F08:185-200
    def init(self, vlen: int, mvec: int, *, flavor: int = FLAVOR_DEFAULT, device: int | None = None,
             stream: int | None = None):
        """vlen is THIS rank's slice length.  Like the Fortran intent(out) dummy,
        init resets vtol (0.01) and the dot-product hook.  `stream` is a raw
        hipStream_t (default: torch's current stream on `device`).  Without
        `flavor` the object runs the build's default -- the same one the Fortran
        `call a%init(vlen, mvec)` runs: compact storage unless NKA_HIP_FLAVOR
        says otherwise (include/nka_hip.h); flavor() reports it."""
        import torch

        self.delete()
        if not torch.cuda.is_available():
            raise NKAError("no HIP device visible: nka_amd has no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        explicit_stream = stream
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        h = C.c_void_p()
        _check(self._L.nka_hip_create(C.byref(h), int(vlen), int(mvec), 0.01, int(flavor), int(device),
                                      C.c_void_p(stream)), "nka_hip_create", self._L)
        self._h, self._device, self._vlen, self._mvec = h, device, int(vlen), int(mvec)
        self._stream, self._follow_torch_stream = int(stream), explicit_stream is None
        return self

    def copy(self):
        """b = a of the reference's type is a DEEP copy (allocatable components,
        F08:154-168): an independent accelerator with the same stored vectors, lists,
        factor and tolerance; the two then evolve separately.  Python-level hooks
        (set_dot_prod, set_host_dot) are carried over; the built-in RCCL communicator
        is not (nka_hip_clone)."""
        other = nka(diagnostic=self._diag, lib=self._libpath)
        h = C.c_void_p()
        _check(self._L.nka_hip_clone(self._handle(), C.byref(h)), "nka_hip_clone", self._L)
        other._h, other._device, other._vlen, other._mvec = h, self._device, self._vlen, self._mvec
        other._stream, other._follow_torch_stream = self._stream, self._follow_torch_stream
        other._cb, other._hd = self._cb, self._hd          # keep the ctypes trampolines alive
        return other

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()

    def delete(self):
        if self._h is not None:
            self._L.nka_hip_destroy(self._h)
            self._h = None
            self.__dict__.pop("_views", None)           # (views of library buffers die with the handle)
            self.__dict__.pop("_swap_keep", None)

    def __del__(self):
        try:
            self.delete()
        except Exception:
            pass

    def _handle(self):
        if self._h is None:
            raise NKAError("nka object used before init")  # the reference would ASSERT in defined()
        return self._h

    # -- call a%set_vec_tol(vtol)                     F08:202-207
    def set_vec_tol(self, vtol: float):
        _check(self._L.nka_hip_set_vec_tol(self._handle(), float(vtol)), "set_vec_tol", self._L)

    # -- call a%set_dot_prod(dot_prod)                F08:209-214
    def set_dot_prod(self, allreduce):
        """Distribution hook.  The reference asks for a global dot product; the
        device build keeps the local partial sums on the GPU and asks only for
        their global SUM: allreduce(ptr:int, count:int, stream:int) -> None must
        sum `count` doubles at device address `ptr` over all ranks in place,
        ordered on `stream`.  None restores the single-rank default."""
        if allreduce is None:
            self._cb = None
            _check(self._L.nka_hip_set_allreduce(self._handle(), C.cast(None, _lib.ALLREDUCE_FN), None), "set_allreduce", self._L)
            return

        def tramp(_ctx, buf, count, stream):
            try:
                allreduce(int(buf), int(count), int(stream or 0))
                return 0
            except Exception as exc:  # surface as NKA_HIP_ECOMM
                import sys
                print(f"nka_amd: allreduce hook raised: {exc!r}", file=sys.stderr)
                return 1

        self._cb = _lib.ALLREDUCE_FN(tramp)
        _check(self._L.nka_hip_set_allreduce(self._handle(), self._cb, None), "set_allreduce", self._L)

    def set_host_dot(self, dot):
        """Source compatibility with the reference's user dot product
        (set_dot_prod(dp), F08:209-219): dot(x, y) -> float over HOST numpy views
        of this rank's slices, returning the GLOBAL dot product.  The update then
        evaluates its inner products by calling it in the reference's order on host
        copies (slow, synchronous: see nka_hip_set_host_dot).  None restores the
        device sums."""
        if dot is None:
            self._hd = None
            _check(self._L.nka_hip_set_host_dot(self._handle(), C.cast(None, _lib.HOST_DOT_FN), None), "set_host_dot", self._L)
            return

        def tramp(_ctx, n, x, y):
            xa = np.ctypeslib.as_array(x, shape=(n,)) if n else np.zeros(0)
            ya = np.ctypeslib.as_array(y, shape=(n,)) if n else np.zeros(0)
            return float(dot(xa, ya))

        self._hd = _lib.HOST_DOT_FN(tramp)
        _check(self._L.nka_hip_set_host_dot(self._handle(), self._hd, None), "set_host_dot", self._L)

    def use_rccl(self, unique_id: bytes, nranks: int, rank: int):
        """Built-in hook: the RCCL all-reduces of an update on the object's stream (two small ones with the default sums -- the norm,
        then the rows --, one in the fast mode SUMS_BLOCKED)."""
        buf = C.create_string_buffer(unique_id, 128)
        _check(self._L.nka_hip_comm_init_rank(self._handle(), buf, nranks, rank), "comm_init_rank", self._L)

    def comm_info(self):
        """(nranks, rank) as the handle's built-in RCCL communicator reports them; (0, -1) without one."""
        n, r = C.c_int32(), C.c_int32()
        _check(self._L.nka_hip_comm_info(self._handle(), C.byref(n), C.byref(r)), "comm_info", self._L)
        return int(n.value), int(r.value)

    def p2p_export(self, nranks: int) -> bytes:
        """Peer-to-peer exchange, step 1 (nka_hip_p2p_export): allocate this rank's mailbox; returns the 64-byte hipIpc
        handle every peer needs."""
        buf = C.create_string_buffer(64)
        _check(self._L.nka_hip_p2p_export(self._handle(), int(nranks), buf), "p2p_export", self._L)
        return buf.raw

    def p2p_attach(self, handles, nranks: int, rank: int):
        """Step 2 (nka_hip_p2p_attach): `handles` = the nranks 64-byte handles in rank order (bytes or a list of bytes)."""
        blob = b"".join(handles) if not isinstance(handles, (bytes, bytearray)) else bytes(handles)
        if len(blob) != 64 * nranks:
            raise NKAError("p2p_attach: need nranks handles of 64 bytes")
        buf = C.create_string_buffer(blob, len(blob))
        _check(self._L.nka_hip_p2p_attach(self._handle(), buf, int(nranks), int(rank)), "p2p_attach", self._L)

    def p2p_mailbox(self) -> int:
        """Device address of this handle's mailbox (after p2p_export): what handles of the SAME process exchange instead
        of hipIpc handles (nka_hip_p2p_mailbox)."""
        p = C.c_void_p()
        _check(self._L.nka_hip_p2p_mailbox(self._handle(), C.byref(p)), "p2p_mailbox", self._L)
        return int(p.value)

    def p2p_attach_local(self, mailboxes, rank: int):
        """Step 2 for handles that share a process (nka_hip_p2p_attach_local): `mailboxes` = every handle's p2p_mailbox()
        in rank order."""
        arr = (C.c_void_p * len(mailboxes))(*[C.c_void_p(int(m)) for m in mailboxes])
        _check(self._L.nka_hip_p2p_attach_local(self._handle(), arr, len(mailboxes), int(rank)), "p2p_attach_local", self._L)

    def p2p_detach(self):
        _check(self._L.nka_hip_p2p_detach(self._handle()), "p2p_detach", self._L)

    def drop_rccl(self):
        _check(self._L.nka_hip_comm_destroy(self._handle()), "comm_destroy", self._L)

    @staticmethod
    def default_flavor() -> int:
        """What FLAVOR_DEFAULT resolves to in this process (the library's rule, mirrored:
        NKA_HIP_FLAVOR if set, else compact storage)."""
        import os
        v = os.environ.get("NKA_HIP_FLAVOR", "").strip().lower()
        return {"f08": FLAVOR_F08, "0": FLAVOR_F08, "f08vec": FLAVOR_F08_VECTOR, "f08_vector": FLAVOR_F08_VECTOR,
                "1": FLAVOR_F08_VECTOR}.get(v, FLAVOR_C)

    @staticmethod
    def rccl_library() -> str:
        """Path of the RCCL shared object the library bound (one copy per process)."""
        buf = C.create_string_buffer(1024)
        _check(_lib.load().nka_hip_comm_library(buf, 1024), "comm_library")
        return buf.value.decode()

    def allreduce_now(self, t):
        """Run the installed all-reduce hook on a float64 CUDA tensor (in place)."""
        _check(self._L.nka_hip_allreduce_now(self._handle(), C.c_void_p(t.data_ptr()), int(t.numel())), "allreduce_now", self._L)
        return t

    def state_digest(self) -> int:
        """Digest of the replicated scalar state; equal on every rank of a sharded run."""
        d = C.c_uint64()
        _check(self._L.nka_hip_state_digest(self._handle(), C.byref(d)), "state_digest", self._L)
        return int(d.value)

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(_lib.load().nka_hip_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    # -- call a%accel_update(f)                       F08:249-419
    def accel_update(self, f):
        """f: torch.float64 CUDA tensor of vec_len() elements (updated in place,
        asynchronously on the object's stream), or a numpy float64 array (host
        compatibility path: copied to the device and back, synchronous)."""
        h = self._handle()
        if isinstance(f, np.ndarray):
            if f.dtype != np.float64 or not f.flags["C_CONTIGUOUS"] or f.size != self._vlen:
                raise NKAError("accel_update: need a contiguous float64 array of vec_len() elements")
            _check(self._L.nka_hip_accel_update_host(h, C.c_void_p(f.ctypes.data)), "accel_update_host", self._L)
            return f
        import torch
        if not (isinstance(f, torch.Tensor) and f.is_cuda and f.dtype == torch.float64 and f.is_contiguous()
                and f.numel() == self._vlen):
            raise NKAError("accel_update: need a contiguous float64 CUDA tensor of vec_len() elements")
        if f.device.index != self._device:
            raise NKAError("accel_update: tensor lives on another device than the accelerator")
        if self._follow_torch_stream:      # stay on torch's CURRENT stream (e.g. inside `with torch.cuda.stream(s)`)
            cur = int(torch.cuda.current_stream(self._device).cuda_stream)
            if cur != self._stream:
                self.set_stream(cur)
        _check(self._L.nka_hip_accel_update(h, C.c_void_p(f.data_ptr())), "accel_update", self._L)
        return f

    def accel_update_swap(self, f, views: bool = True):
        """Out-of-place update (nka_hip_accel_update_swap): `f`, a float64 CUDA tensor holding the correction, is HANDED
        to the accelerator -- it becomes the storage of the new pair's w, so it is kept alive here and must not be
        written again.  Returns (buf, acc): `buf`, a tensor view of a free library buffer for the caller's next input;
        `acc`, a view of the accelerated correction, to be read only, valid until the next call on this object.
        views=False returns the two raw device addresses instead (wrapping a pointer in a tensor costs ~0.4 ms the
        first time it is seen)."""
        import torch
        h = self._handle()
        if not (isinstance(f, torch.Tensor) and f.is_cuda and f.dtype == torch.float64 and f.is_contiguous()
                and f.numel() == self._vlen and f.device.index == self._device):
            raise NKAError("accel_update_swap: need a contiguous float64 CUDA tensor of vec_len() elements on the accelerator's device")
        if self._follow_torch_stream:
            cur = int(torch.cuda.current_stream(self._device).cuda_stream)
            if cur != self._stream:
                self.set_stream(cur)
        io, acc = C.c_void_p(f.data_ptr()), C.c_void_p()
        _check(self._L.nka_hip_accel_update_swap(h, C.byref(io), C.byref(acc)), "accel_update_swap", self._L)
        if not hasattr(self, "_swap_keep"):
            self._swap_keep = []
        if not any(t.data_ptr() == f.data_ptr() for t in self._swap_keep):
            self._swap_keep.append(f)                   # the library keeps using this memory
            self.__dict__.setdefault("_views", {}).setdefault(f.data_ptr(), f)
        if not views:
            return int(io.value), int(acc.value)
        return self._view(io.value), self._view(acc.value)

    def _view(self, ptr):
        import torch
        cache = self.__dict__.setdefault("_views", {})          # (a handful of buffers circulate: wrap each once)
        if ptr in cache:
            return cache[ptr]

        class _Alias:
            def __init__(self, p, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (p, False), "version": 2}
        if self._vlen == 0:
            return torch.empty(0, dtype=torch.float64, device=f"cuda:{self._device}")
        cache[ptr] = torch.as_tensor(_Alias(ptr, self._vlen), device=f"cuda:{self._device}")
        return cache[ptr]

    def capture_safe(self) -> bool:
        """True once an accel_update captured into a hipGraph stays valid on replay
        (steady state: see nka_hip_capture_safe in include/nka_hip.h)."""
        return self._L.nka_hip_capture_safe(self._handle()) == 1

    def set_sum_order(self, order: int):
        """How the inner products are summed (nka_hip_set_sum_order): SUMS_REFERENCE_ORDER = every sum as the reference forms
        it, so that an update returns the reference's bits at any n (single rank; slow beyond a few thousand elements);
        SUMS_AUTO (default) = reference order where it costs nothing (n <= 64), else SUMS_BLOCKED_ROUNDED = the fast passes with
        the norm first and the Gram row on the rounded w1' (what every front end runs since round 6); SUMS_BLOCKED = the opt-in
        single-pass fast mode (raw-sum Gram row; one exchange per update)."""
        _check(self._L.nka_hip_set_sum_order(self._handle(), int(order)), "set_sum_order", self._L)
        return self

    def set_shard(self, rank: int, nranks: int):
        """Where this object's slice lies in the global vector (slice `rank` of `nranks`, in rank order): needed by the
        sharded reference-order sums only (nka_hip_set_shard); use_rccl sets it by itself."""
        _check(self._L.nka_hip_set_shard(self._handle(), int(rank), int(nranks)), "set_shard", self._L)
        return self

    def list_bound(self) -> int:
        """The host's upper bound on the list length at the entry of the next update (no synchronisation):
        its own count tightened by the device's list word (nka_hip_list_bound)."""
        n = self._L.nka_hip_list_bound(self._handle())
        if n < 0:
            _check(n, "list_bound")
        return n

    def set_stream(self, stream: int):
        """Rebind to another hipStream_t; earlier work stays ordered before later work."""
        _check(self._L.nka_hip_set_stream(self._handle(), C.c_void_p(int(stream))), "set_stream", self._L)
        self._stream = int(stream)

    # -- call a%restart() / a%relax()                 F08:422-457
    def restart(self):
        _check(self._L.nka_hip_restart(self._handle()), "restart", self._L)

    def relax(self):
        _check(self._L.nka_hip_relax(self._handle()), "relax", self._L)

    # -- accessors                                    F08:221-246
    def num_vec(self) -> int:
        n = self._L.nka_hip_num_vec(self._handle())
        if n < 0:
            _check(n, "num_vec")
        return n

    def flavor(self) -> int:
        """FLAVOR_* this object runs (the default resolved)."""
        return self._L.nka_hip_flavor(self._handle())

    def max_vec(self) -> int:
        return self._L.nka_hip_max_vec(self._handle())

    def vec_len(self) -> int:
        return self._L.nka_hip_vec_len(self._handle())

    def vec_tol(self) -> float:
        return self._L.nka_hip_vec_tol(self._handle())

    # -- a%defined()                                  F08:460-524
    def defined(self) -> bool:
        return self._h is not None and bool(self._L.nka_hip_defined(self._h))

    # -- test / bench instrumentation (not in the reference) ---------------
    def state(self) -> State:
        n = self._mvec + 1
        ints = [C.c_int32() for _ in range(5)]
        nxt = np.zeros(n, np.int32)
        prv = np.zeros(n, np.int32)
        h = np.zeros((n, n), np.float64)
        c = np.zeros(n, np.float64)
        _check(self._L.nka_hip_get_state(self._handle(), *[C.byref(i) for i in ints],
                                         nxt.ctypes.data_as(_lib._i32p), prv.ctypes.data_as(_lib._i32p),
                                         h.ctypes.data_as(_lib._dp), c.ctypes.data_as(_lib._dp)), "get_state", self._L)
        return State(ints[0].value, ints[1].value, ints[2].value, ints[3].value, ints[4].value, nxt, prv,
                     h.T.copy(), c)

    def reductions(self) -> np.ndarray:
        """[<d,d>, <f,d>, <d,w_p>..., <f,w_p>...] (d = w1 - f) of the most recent update."""
        out = np.zeros(2 + 2 * self._mvec)
        _check(self._L.nka_hip_get_reductions(self._handle(), out.ctypes.data_as(_lib._dp)), "get_reductions", self._L)
        return out

    def w(self, slot: int) -> np.ndarray:
        out = np.zeros(self._vlen)
        _check(self._L.nka_hip_get_w(self._handle(), slot, out.ctypes.data_as(_lib._dp)), "get_w", self._L)
        return out

    def v(self, slot: int) -> np.ndarray:
        out = np.zeros(self._vlen)
        _check(self._L.nka_hip_get_v(self._handle(), slot, out.ctypes.data_as(_lib._dp)), "get_v", self._L)
        return out

    def set_timing(self, capacity: int = 1, stride: int = 1):
        """Keep HIP-event timings of the last `capacity` recorded updates (0 = off); with `stride` > 1 only every
        stride-th update is recorded (four event records widen an update by ~15 us: matters below n ~ 1e7)."""
        _check(self._L.nka_hip_set_timing(self._handle(), int(capacity)), "set_timing", self._L)
        _check(self._L.nka_hip_set_timing_stride(self._handle(), int(stride)), "set_timing_stride", self._L)

    def timing_ms(self, back: int = 0):
        """(PA dots, solve, PB combine, whole update) in ms for the update `back` calls ago."""
        ms = (C.c_float * 4)()
        _check(self._L.nka_hip_get_timing(self._handle(), int(back), ms), "get_timing", self._L)
        return tuple(ms)

    def _need_diag(self, what):
        if not self._diag:
            raise NKAError(f"{what} exists only in the diagnostic build: construct the object with nka(diagnostic=True)")

    def set_grid(self, pa=0, pb=0):
        self._need_diag("set_grid")
        _check(self._L.nka_hip_set_grid(self._handle(), pa, pb), "set_grid", self._L)

    def set_tuning(self, key: str, value: int):
        """Kernel-variant switch for in-process A/B measurements (nka_hip_set_tuning, include/nka_hip_diag.h)."""
        self._need_diag("set_tuning")
        _check(self._L.nka_hip_set_tuning(self._handle(), key.encode(), int(value)), "set_tuning", self._L)

    def debug_chain_sum(self, x, y, start: float = 0.0, walk: bool = False, many: bool = False):
        """start + x[0]*y[0] + x[1]*y[1] + ... formed by the reference-order kernel of long vectors over two device
        tensors (nka_hip_debug_chain_sum, include/nka_hip_diag.h); many: through the many-compute-unit kernels.  Returns
        (sum, kernel milliseconds)."""
        self._need_diag("debug_chain_sum")
        assert x.dtype == y.dtype and x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous()
        out, ms = C.c_double(), C.c_float()
        _check(self._L.nka_hip_debug_chain_sum(self._handle(), x.data_ptr(), y.data_ptr(), x.numel(), float(start),
                                               int(bool(walk)) | (2 if many else 0), C.byref(out), C.byref(ms)), "debug_chain_sum", self._L)
        return out.value, ms.value

    def device_info(self):
        name = C.create_string_buffer(64)
        ncu = C.c_int32()
        _check(self._L.nka_hip_device_info(self._handle(), name, C.byref(ncu)), "device_info", self._L)
        return name.value.decode(), ncu.value
