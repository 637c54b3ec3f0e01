"""Device implementations of the abstract-vector hooks (nka_amd/csrc/vec_ops.hip)
through the C ABI: elementwise results BIT-EXACT against the Fortran expressions
of the reference's grid_vector (src-F08-vector/grid_vector_type.F90:104-165,
evaluated with numpy: IEEE, left to right, no FMA); reductions within tolerance;
the batched hooks equal the loops they replace."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ws():
    import torch
    import nka_amd
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    L = nka_amd.load()
    h = C.c_void_p()
    assert L.nka_hip_vec_workspace_create(C.byref(h), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    yield L, h, torch
    L.nka_hip_vec_workspace_destroy(h)


@pytest.fixture(scope="module")
def ws_diag():
    """A workspace of the DIAGNOSTIC build (libnka_hip_diag.so): the tests that force the combine stage onto 0 / 1 / 2 / 8
    ticket counters (nka_hip_vec_set_tuning, include/nka_hip_diag.h); -1 there is the product's automatic rule."""
    import torch
    from nka_amd import _lib
    torch.cuda.set_device(0)
    L = _lib.load_diag()
    h = C.c_void_p()
    assert L.nka_hip_vec_workspace_create(C.byref(h), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    yield L, h, torch
    L.nka_hip_vec_workspace_destroy(h)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n", [1, 2, 511, 512, 513, 100003])
def test_elementwise_hooks_bit_exact(ws, n):
    L, h, torch = ws
    rng = np.random.default_rng(n)
    x, y, z0 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    a, b, c = 0.7312, -1.25e-3, 3.5
    xd, yd = _dev(torch, x), _dev(torch, y)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    zd = _dev(torch, z0)
    assert L.nka_hip_vec_scale(h, n, P(zd), a) == 0
    assert np.array_equal(zd.cpu().numpy(), a * z0)
    zd = _dev(torch, z0)
    assert L.nka_hip_vec_update1(h, n, P(zd), a, P(xd)) == 0
    assert np.array_equal(zd.cpu().numpy(), a * x + z0)
    zd = _dev(torch, z0)
    assert L.nka_hip_vec_update2(h, n, P(zd), a, P(xd), b) == 0
    assert np.array_equal(zd.cpu().numpy(), a * x + b * z0)
    zd = _dev(torch, z0)
    assert L.nka_hip_vec_update3(h, n, P(zd), a, P(xd), b, P(yd)) == 0
    assert np.array_equal(zd.cpu().numpy(), (a * x + b * y) + z0)
    zd = _dev(torch, z0)
    assert L.nka_hip_vec_update4(h, n, P(zd), a, P(xd), b, P(yd), c) == 0
    assert np.array_equal(zd.cpu().numpy(), (a * x + b * y) + c * z0)
    zd = _dev(torch, z0)
    assert L.nka_hip_vec_setval(h, n, P(zd), 2.5) == 0
    assert np.array_equal(zd.cpu().numpy(), np.full(n, 2.5))
    assert L.nka_hip_vec_copy(h, n, P(zd), P(xd)) == 0
    assert np.array_equal(zd.cpu().numpy(), x)

    r = C.c_double()
    assert L.nka_hip_vec_dot(h, n, P(xd), P(yd), C.byref(r)) == 0
    assert r.value == pytest.approx(float(x @ y), abs=1e-13 * np.linalg.norm(x) * np.linalg.norm(y))
    assert L.nka_hip_vec_norm2(h, n, P(xd), C.byref(r)) == 0
    assert r.value == pytest.approx(float(np.linalg.norm(x)), rel=1e-14)


@pytest.mark.parametrize("n,count", [(1000, 1), (4099, 5), (100003, 20), (777, 37)])
def test_batched_hooks_equal_the_loops_they_replace(ws, n, count):
    L, h, torch = ws
    rng = np.random.default_rng(n + count)
    X = rng.standard_normal((count, n))
    Y = rng.standard_normal((count, n))
    z0 = rng.standard_normal(n)
    a, b = rng.standard_normal(count), rng.standard_normal(count)
    Xd = [_dev(torch, X[j]) for j in range(count)]
    Yd = [_dev(torch, Y[j]) for j in range(count)]
    xs = (C.c_void_p * count)(*[t.data_ptr() for t in Xd])
    ys = (C.c_void_p * count)(*[t.data_ptr() for t in Yd])
    zd = _dev(torch, z0)
    dp = C.POINTER(C.c_double)

    # update_many == count successive update3_ calls, bit for bit
    assert L.nka_hip_vec_update_many(h, n, C.c_void_p(zd.data_ptr()), a.ctypes.data_as(dp), xs,
                                     b.ctypes.data_as(dp), ys, count) == 0
    ref = z0.copy()
    for j in range(count):
        ref = (a[j] * X[j] + b[j] * Y[j]) + ref
    assert np.array_equal(zd.cpu().numpy(), ref)

    # axpy_many == count successive update1_ calls, bit for bit
    zd2 = _dev(torch, z0)
    assert L.nka_hip_vec_axpy_many(h, n, C.c_void_p(zd2.data_ptr()), a.ctypes.data_as(dp), xs, count) == 0
    ref2 = z0.copy()
    for j in range(count):
        ref2 = a[j] * X[j] + ref2
    assert np.array_equal(zd2.cpu().numpy(), ref2)

    # dot_many == count dot_ calls within the reduction tolerance
    vals = np.zeros(count)
    wd = _dev(torch, z0)
    assert L.nka_hip_vec_dot_many(h, n, C.c_void_p(wd.data_ptr()), xs, count, vals.ctypes.data_as(dp)) == 0
    for j in range(count):
        assert vals[j] == pytest.approx(float(z0 @ X[j]), abs=1e-13 * np.linalg.norm(z0) * np.linalg.norm(X[j]))

    # dot_pair_many: both rows and the cross term in one pass
    v0, v1, cross = np.zeros(count), np.zeros(count), C.c_double()
    ud = _dev(torch, Y[0])
    assert L.nka_hip_vec_dot_pair_many(h, n, C.c_void_p(wd.data_ptr()), C.c_void_p(ud.data_ptr()), xs, count,
                                       v0.ctypes.data_as(dp), v1.ctypes.data_as(dp), C.byref(cross)) == 0
    assert cross.value == pytest.approx(float(z0 @ Y[0]), abs=1e-13 * np.linalg.norm(z0) * np.linalg.norm(Y[0]))
    for j in range(count):
        assert v0[j] == pytest.approx(float(z0 @ X[j]), abs=1e-13 * np.linalg.norm(z0) * np.linalg.norm(X[j]))
        assert v1[j] == pytest.approx(float(Y[0] @ X[j]), abs=1e-13 * np.linalg.norm(Y[0]) * np.linalg.norm(X[j]))


def test_unaligned_operands_take_the_scalar_path(ws):
    L, h, torch = ws
    n = 3001
    base = torch.arange(2 * n + 2, dtype=torch.float64, device="cuda") * 0.001
    x = base[1:n + 1]                      # 8-byte aligned only
    z = base[n + 1:2 * n + 1].clone()
    z0 = z.cpu().numpy().copy()
    assert x.data_ptr() % 16 == 8
    assert L.nka_hip_vec_update1(h, n, C.c_void_p(z.data_ptr()), 2.0, C.c_void_p(x.data_ptr())) == 0
    assert np.array_equal(z.cpu().numpy(), 2.0 * x.cpu().numpy() + z0)


@pytest.mark.parametrize("n,count,subtract", [(1, 0, 0), (513, 3, 1), (4099, 5, 0), (100003, 20, 0), (100003, 20, 1),
                                              (777, 37, 1), (1100003, 7, 0)] +
                         # every unroll width of the rolling-window kernels (exact widths 1..24, ring sizes 1..23);
                         # 300 007 elements = 586 tiles over 256 blocks: the window crosses tile boundaries
                         [(300007, c, c % 2) for c in range(1, 25)])
def test_fused_stage_hooks_equal_the_hook_sequences_they_replace(ws_diag, n, count, subtract):
    """update_norm2, scale_dot_pair_many, update_many_keep, axpy_many_keep: stored
    vectors BIT-EXACT against the sequences of deferred-hook expressions they
    fuse (F08V:237-238, 255-264, 336+374+382), reductions within tolerance."""
    ws = ws_diag
    L, h, torch = ws
    rng = np.random.default_rng(7 * n + count)
    dp = C.POINTER(C.c_double)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    f, w0, v0 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    Y = rng.standard_normal((max(count, 1), n))
    Yd = [_dev(torch, Y[j]) for j in range(count)]
    ys = (C.c_void_p * max(count, 1))(*[t.data_ptr() for t in Yd])

    # stage 1, storing form: w <- (-1)*f + w ; s = ||w||
    wd, fd, vd = _dev(torch, w0), _dev(torch, f), _dev(torch, v0)
    s = C.c_double()
    assert L.nka_hip_vec_update_norm2(h, n, P(wd), -1.0, P(fd), 1, C.byref(s)) == 0
    d = -1.0 * f + w0
    assert np.array_equal(wd.cpu().numpy(), d)
    assert s.value == pytest.approx(float(np.linalg.norm(d)), rel=1e-14)
    # stage 1, deferring form: same norm (same bits), w untouched
    wd2, s2 = _dev(torch, w0), C.c_double()
    assert L.nka_hip_vec_update_norm2(h, n, P(wd2), -1.0, P(fd), 0, C.byref(s2)) == 0
    assert np.array_equal(wd2.cpu().numpy(), w0) and s2.value == s.value

    # stage 2: scale both, optional subtract, both rows + cross with the NEW w --
    # once on the stored difference, once applying the deferred update itself
    a = 1.0 / s.value
    wn = a * d
    vn = a * v0
    if subtract:
        vn = -1.0 * wn + vn
    nf, nw = np.linalg.norm(f), np.linalg.norm(wn)
    for pre, wdev in ((0, wd), (1, wd2)):
        vdev = _dev(torch, v0)
        vw, vf, cross = np.zeros(max(count, 1)), np.zeros(max(count, 1)), C.c_double()
        assert L.nka_hip_vec_scale_dot_pair_many(h, n, P(wdev), P(vdev), a, subtract, pre, -1.0, P(fd), ys, count,
                                                 vw.ctypes.data_as(dp), vf.ctypes.data_as(dp), C.byref(cross)) == 0
        assert np.array_equal(wdev.cpu().numpy(), wn), pre
        assert np.array_equal(vdev.cpu().numpy(), vn), pre
        assert cross.value == pytest.approx(float(f @ wn), abs=1e-13 * nf * nw)
        for j in range(count):
            ny = np.linalg.norm(Y[j])
            assert vw[j] == pytest.approx(float(wn @ Y[j]), abs=1e-13 * nw * ny)
            assert vf[j] == pytest.approx(float(f @ Y[j]), abs=1e-13 * nf * ny)

    # stage 3: keep_in <- z ; z <- combine ; keep_out <- z
    X = rng.standard_normal((max(count, 1), n))
    Xd = [_dev(torch, X[j]) for j in range(count)]
    xs = (C.c_void_p * max(count, 1))(*[t.data_ptr() for t in Xd])
    ca, cb = rng.standard_normal(max(count, 1)), rng.standard_normal(max(count, 1))
    # (the combine stage also with its tiles taken from 1, 2 or 8 global ticket counters: same bits,
    #  and the counters are back at zero for the launch that follows)
    for pairs, tickets in [(True, -1), (False, -1)] + ([(True, 1), (False, 2), (True, 8), (False, 1), (True, -1)] if n >= 300007 else []):
        assert L.nka_hip_vec_set_tuning(h, b"tickets", tickets) == 0
        zd = _dev(torch, f)
        kin = torch.zeros(n, dtype=torch.float64, device="cuda")
        kout = torch.zeros(n, dtype=torch.float64, device="cuda")
        if pairs:
            assert L.nka_hip_vec_update_many_keep(h, n, P(zd), ca.ctypes.data_as(dp), xs, cb.ctypes.data_as(dp), ys,
                                                  count, P(kin), P(kout)) == 0
        else:
            assert L.nka_hip_vec_axpy_many_keep(h, n, P(zd), ca.ctypes.data_as(dp), xs, count, P(kin), P(kout)) == 0
        ref = f.copy()
        for j in range(count):
            ref = (ca[j] * X[j] + cb[j] * Y[j]) + ref if pairs else ca[j] * X[j] + ref
        assert np.array_equal(kin.cpu().numpy(), f)
        assert np.array_equal(zd.cpu().numpy(), ref)
        assert np.array_equal(kout.cpu().numpy(), ref)


@pytest.mark.parametrize("unaligned", [0, 1])
@pytest.mark.parametrize("n,count,pre,tickets", [(1, 1, 1, -1), (513, 1, 0, -1), (4099, 4, 1, -1), (100003, 21, 1, -1),
                                                 (100003, 24, 0, -1), (100003, 25, 1, -1), (300007, 7, 1, 1), (1100003, 21, 1, 2),
                                                 (777, 23, 1, -1)])
def test_deferred_normalisation_equals_the_storing_stages(ws_diag, n, count, pre, tickets, unaligned):
    """The scale-and-dot stage as a pure read (dot_pair_many_scaled) followed by a
    combine that normalises the pending pair itself (update_many_keep_pend /
    axpy_many_keep_pend) must leave EVERY bit where the storing stages
    (scale_dot_pair_many, then update_many_keep / axpy_many_keep) leave it: the new
    pair, f, both kept copies, and the three rows of inner products.  `count` counts
    the pending pair; 16-byte path and (unaligned = 1) the 8-byte fallback; with
    the tiles of the combine taken from ticket counters where the size allows."""
    ws = ws_diag
    L, h, torch = ws
    rng = np.random.default_rng(11 * n + count + pre)
    dp = C.POINTER(C.c_double)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    off = unaligned

    def dev(a):                      # a device copy whose data pointer is 8 (not 16) bytes aligned when asked
        t = torch.empty(a.size + 1, dtype=torch.float64, device="cuda")
        t[off:off + a.size] = torch.from_numpy(a)
        return t[off:off + a.size]

    f = rng.standard_normal(n)
    w1, v1 = rng.standard_normal(n), rng.standard_normal(n)
    nold = count - 1
    Wo, Vo = rng.standard_normal((max(nold, 1), n)), rng.standard_normal((max(nold, 1), n))
    s = float(np.linalg.norm(-1.0 * f + w1)) if pre else float(np.linalg.norm(w1))
    a = 1.0 / s
    ca, cb = rng.standard_normal(count), rng.standard_normal(count)
    assert L.nka_hip_vec_set_tuning(h, b"tickets", tickets) == 0
    for compact in (0, 1):
        res = []
        for deferred in (0, 1):
            fd, w1d, v1d = dev(f), dev(w1), dev(v1)
            Wd = [dev(Wo[j]) for j in range(nold)]
            Vd = [dev(Vo[j]) for j in range(nold)]
            kin, kout = dev(np.zeros(n)), dev(np.zeros(n))
            ys = (C.c_void_p * max(nold, 1))(*[t.data_ptr() for t in Wd])
            vw, vf, cross = np.zeros(max(nold, 1)), np.zeros(max(nold, 1)), C.c_double()
            if deferred:
                assert L.nka_hip_vec_dot_pair_many_scaled(h, n, P(w1d), a, pre, -1.0, P(fd), ys, nold,
                                                          vw.ctypes.data_as(dp), vf.ctypes.data_as(dp), C.byref(cross)) == 0
                assert np.array_equal(w1d.cpu().numpy(), w1) and np.array_equal(v1d.cpu().numpy(), v1)   # untouched
            else:
                assert L.nka_hip_vec_scale_dot_pair_many(h, n, P(w1d), P(v1d), a, compact, pre, -1.0, P(fd), ys, nold,
                                                         vw.ctypes.data_as(dp), vf.ctypes.data_as(dp), C.byref(cross)) == 0
            # the combine over the whole list, newest (the pending pair) first
            if compact:
                xs = (C.c_void_p * count)(*([v1d.data_ptr()] + [t.data_ptr() for t in Vd]))
                if deferred:
                    assert L.nka_hip_vec_axpy_many_keep_pend(h, n, P(fd), ca.ctypes.data_as(dp), xs, count, P(kin), P(kout),
                                                             P(w1d), a, pre, -1.0) == 0
                else:
                    assert L.nka_hip_vec_axpy_many_keep(h, n, P(fd), ca.ctypes.data_as(dp), xs, count, P(kin), P(kout)) == 0
            else:
                xs = (C.c_void_p * count)(*([w1d.data_ptr()] + [t.data_ptr() for t in Wd]))
                yv = (C.c_void_p * count)(*([v1d.data_ptr()] + [t.data_ptr() for t in Vd]))
                if deferred:
                    assert L.nka_hip_vec_update_many_keep_pend(h, n, P(fd), ca.ctypes.data_as(dp), xs, cb.ctypes.data_as(dp),
                                                               yv, count, P(kin), P(kout), a, pre, -1.0, 0) == 0
                else:
                    assert L.nka_hip_vec_update_many_keep(h, n, P(fd), ca.ctypes.data_as(dp), xs, cb.ctypes.data_as(dp), yv,
                                                          count, P(kin), P(kout)) == 0
            res.append([t.cpu().numpy() for t in (w1d, v1d, fd, kin, kout)] + [vw.copy(), vf.copy(), np.array([cross.value])])
        for x, y in zip(res[0], res[1]):
            assert np.array_equal(x, y), (compact, np.abs(x - y).max())
        # and against the expressions themselves (numpy: IEEE, left to right, no FMA)
        d = -1.0 * f + w1 if pre else w1
        wn = a * d
        vn = a * v1
        if compact:
            vn = -1.0 * wn + vn
        assert np.array_equal(res[1][0], wn) and np.array_equal(res[1][1], vn)
        ref = f.copy()
        for j in range(count):
            xj = (vn if j == 0 else Vo[j - 1]) if compact else (wn if j == 0 else Wo[j - 1])
            yj = vn if j == 0 else Vo[j - 1]
            ref = ca[j] * xj + ref if compact else (ca[j] * xj + cb[j] * yj) + ref
        assert np.array_equal(res[1][2], ref) and np.array_equal(res[1][3], f) and np.array_equal(res[1][4], ref)
    assert L.nka_hip_vec_set_tuning(h, b"tickets", -1) == 0


def test_reduction_hooks_see_every_sum_in_the_canonical_layout(ws):
    """Parallel-aware reductions (include/nka_hip.h): with hooks installed on the workspace EVERY
    sum a reduction returns goes through them -- the device-side hook on a device buffer, ordered on the
    workspace stream, then the host-side hook on host memory -- in a layout that depends only on the list
    length: [row 0 (count), row 1 (count), cross, <d,d>].  The hooks here play a second rank that holds the
    same slice (device hook: x2) and a third one holding zeros (host hook: records what it sees), so every
    result must double, the norm must grow by sqrt(2), and the recorded layouts must be the canonical ones
    -- for aligned operands (rolling-window kernels, exact widths) AND unaligned ones (8-byte path, widths
    padded to a multiple of 4: the padding must not reach the hook)."""
    import nka_amd
    from nka_amd import _lib
    L, h, torch = ws
    n, count = 70001, 5
    rng = np.random.default_rng(5)
    seen = []

    def dev_hook(_ctx, buf, cnt, stream):
        class _Alias:
            def __init__(self, ptr, c):
                self.__cuda_array_interface__ = {"shape": (c,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        with torch.cuda.stream(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()):
            t = torch.as_tensor(_Alias(buf, cnt), device="cuda")
            t.mul_(2.0)
        return 0

    def host_hook(_ctx, vals, cnt):
        seen.append([vals[i] for i in range(cnt)])
        return 0

    dcb, hcb = _lib.ALLREDUCE_FN(dev_hook), _lib.HOST_ALLREDUCE_FN(host_hook)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    try:
        for shift in (0, 1):                     # shift 1: operands start 8 bytes off a 16-byte boundary
            base = [torch.from_numpy(rng.standard_normal(n + 1)).cuda() for _ in range(count + 2)]
            vecs = [b[shift:shift + n] for b in base]
            w, f, ys = vecs[0], vecs[1], vecs[2:]
            ptrs = (C.c_void_p * count)(*[y.data_ptr() for y in ys])
            wn, fn, yn = w.cpu().numpy(), f.cpu().numpy(), [y.cpu().numpy() for y in ys]
            d = (-1.0) * fn + wn

            def run():
                out = {}
                r = C.c_double()
                assert L.nka_hip_vec_dot(h, n, P(w), P(f), C.byref(r)) == 0
                out["dot"] = r.value
                assert L.nka_hip_vec_norm2(h, n, P(w), C.byref(r)) == 0
                out["norm2"] = r.value
                vz, vx = (C.c_double * count)(), (C.c_double * count)()
                dd, cr = C.c_double(), C.c_double()
                assert L.nka_hip_vec_diff_norm_dot_pair_many(h, n, P(w), -1.0, P(f), ptrs, count, C.byref(dd), vz, vx,
                                                             C.byref(cr)) == 0
                out["fused"] = (dd.value, list(vz), list(vx), cr.value)
                vm = (C.c_double * count)()
                assert L.nka_hip_vec_dot_many(h, n, P(f), ptrs, count, vm) == 0
                out["many"] = list(vm)
                return out

            plain = run()
            assert plain["dot"] == pytest.approx(float(np.dot(wn, fn)), rel=1e-12, abs=1e-9)
            assert plain["fused"][0] == pytest.approx(float(np.dot(d, d)), rel=1e-12)
            for j in range(count):
                assert plain["fused"][1][j] == pytest.approx(float(np.dot(d, yn[j])), rel=1e-10, abs=1e-9)
                assert plain["fused"][2][j] == pytest.approx(float(np.dot(fn, yn[j])), rel=1e-10, abs=1e-9)
            assert plain["fused"][3] == pytest.approx(float(np.dot(fn, d)), rel=1e-10, abs=1e-9)
            assert L.nka_hip_vec_set_allreduce(h, dcb, None) == 0
            assert L.nka_hip_vec_set_host_allreduce(h, hcb, None) == 0
            seen.clear()
            hooked = run()
            assert L.nka_hip_vec_set_allreduce(h, C.cast(None, _lib.ALLREDUCE_FN), None) == 0
            assert L.nka_hip_vec_set_host_allreduce(h, C.cast(None, _lib.HOST_ALLREDUCE_FN), None) == 0
            assert hooked["dot"] == 2.0 * plain["dot"]                     # powers of two: exact
            assert hooked["norm2"] == pytest.approx(np.sqrt(2.0) * plain["norm2"], rel=1e-15)   # sqrt of the GLOBAL sum
            assert hooked["many"] == [2.0 * v for v in plain["many"]]
            dd, vz, vx, cr = plain["fused"]
            assert hooked["fused"] == (2.0 * dd, [2.0 * v for v in vz], [2.0 * v for v in vx], 2.0 * cr)
            # what the host hook saw, call by call: dot (1), norm2's dot (1), the fused stage (2*count + 2), dot_many (count)
            assert [len(s) for s in seen] == [1, 1, 2 * count + 2, count], (shift, [len(s) for s in seen])
            assert seen[2] == [2.0 * v for v in vz] + [2.0 * v for v in vx] + [2.0 * cr, 2.0 * dd]

        # a failing hook fails the reduction with NKA_HIP_ECOMM and leaves the workspace usable
        bad = _lib.HOST_ALLREDUCE_FN(lambda _c, _v, _n: 1)
        assert L.nka_hip_vec_set_host_allreduce(h, bad, None) == 0
        r = C.c_double()
        assert L.nka_hip_vec_dot(h, n, P(w), P(f), C.byref(r)) == -4
        assert b"all-reduce hook failed" in L.nka_hip_last_error()
    finally:
        L.nka_hip_vec_set_allreduce(h, C.cast(None, _lib.ALLREDUCE_FN), None)
        L.nka_hip_vec_set_host_allreduce(h, C.cast(None, _lib.HOST_ALLREDUCE_FN), None)
    r = C.c_double()
    assert L.nka_hip_vec_dot(h, n, P(w), P(f), C.byref(r)) == 0 and r.value == plain["dot"]
