"""GPU tests of round 4: the LIST WORD (the device tells the host, without being asked, how long the list is after
dependence drops; include/nka_hip.h: nka_hip_list_bound), and what hangs on it."""
import os

import numpy as np
import pytest

import parity_util as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def span_inputs(n, calls, dim, seed):
    """Inputs confined to a `dim`-dimensional span: every difference of two of them lies in it, so the subspace can
    hold `dim` vectors and every further update drops one as dependent (F08:326-345)."""
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((dim, n))
    return [rng.standard_normal(dim) @ B for _ in range(calls)]


@pytest.mark.parametrize("flavor", [0, 2])
@pytest.mark.parametrize("n,m,dim", [(40961, 10, 4), (100003, 20, 12), (2049, 6, 1)])
def test_list_word_tightens_the_launch_widths_and_changes_no_bit(torch_cuda, oracle, flavor, n, m, dim):
    """A caller that synchronises once per update (every solver does: it reads a residual norm) must see the host's
    bound equal the TRUE list length after a dependence drop, where the host's own count (+1 per update up to mvec+1)
    stays at the full width; the narrower launches change no bit of any output; decisions equal the oracle's."""
    import nka_amd
    torch = torch_cuda
    X = span_inputs(n, m + 10, dim, seed=31 * m + flavor)
    tight = nka_amd.nka().init(n, m, flavor=flavor)
    loose = nka_amd.nka(diagnostic=True).init(n, m, flavor=flavor)       # (libnka_hip_diag.so: include/nka_hip_diag.h)
    loose.set_tuning("list_word", 0)                 # the behaviour before round 4: the host's own count only
    ora = oracle.OracleNKA(n, m, flavor)
    spread = P.Spread(oracle, n, m)
    seen_short = 0
    for t, x in enumerate(X):
        f = x.copy()
        ora.accel_update(f)
        spread.update(x)
        ft, fl = torch.from_numpy(x.copy()).cuda(), torch.from_numpy(x.copy()).cuda()
        tight.accel_update(ft)
        loose.accel_update(fl)
        torch.cuda.synchronize()
        assert torch.equal(ft, fl), t                                  # same bits whatever the width
        st = tight.state()
        assert st.list_order() == ora.state().list_order(), t          # decisions exact
        true_len = len(st.list_order())
        assert tight.list_bound() == true_len, (t, tight.list_bound(), true_len)      # exact after a synchronisation
        assert loose.list_bound() == min(t + 1, m + 1)                                # the plain count
        seen_short += tight.list_bound() < loose.list_bound()
        err = np.linalg.norm(ft.cpu().numpy() - f) / np.linalg.norm(x)
        P.check(err, ora.state(), f"list word n={n} m={m} dim={dim} flavor {flavor}", base=1e-12, where=t, spread=spread.value,
                truth=spread.truth(ft.cpu().numpy(), x))
    assert seen_short >= m + 10 - (dim + 2) - 1          # from the first drop on the count is too long, the word is not
    assert tight.num_vec() == min(dim, m)
    assert tight.state_digest() == loose.state_digest()
    assert not tight.capture_safe() and tight.list_bound() < m + 1     # a graph captured now would be too narrow later


def test_list_bound_is_an_upper_bound_through_relax_restart_and_unsynchronised_runs(torch_cuda):
    """Random call sequences with synchronisation only now and then: the bound is never below the true list length
    (checked after the fact), equals it whenever the caller has just synchronised, relax / restart included."""
    import nka_amd
    torch = torch_cuda
    rng = np.random.default_rng(7)
    n, m, dim = 30011, 8, 3
    B = rng.standard_normal((dim, n))
    acc = nka_amd.nka().init(n, m)
    synced = True
    for t in range(150):
        r = rng.random()
        if r < 0.08:
            acc.relax()
        elif r < 0.12:
            acc.restart()
        elif r < 0.18:
            acc.set_vec_tol(float(rng.choice([0.01, 0.3])))          # (synchronises)
            synced = True
        else:
            x = rng.standard_normal(dim) @ B if rng.random() < 0.7 else rng.standard_normal(n)
            acc.accel_update(torch.from_numpy(x).cuda())
            synced = False
        bound = acc.list_bound()                  # BEFORE looking at the device
        if rng.random() < 0.4:
            torch.cuda.synchronize()
            synced = True
            bound_synced = acc.list_bound()
            true_len = len(acc.state().list_order())
            assert bound_synced == true_len, (t, bound_synced, true_len)
        true_len = len(acc.state().list_order())              # (synchronises; after the fact)
        assert bound >= true_len, (t, bound, true_len)
        assert acc.defined()


def test_capture_switches_the_list_word_off_for_good(torch_cuda):
    """A captured update is replayed with the widths of the capture: the handle stops using and publishing the word
    the moment its stream is seen capturing; replays and later eager calls stay correct (full width)."""
    import nka_amd
    torch = torch_cuda
    n, m = 20000, 4
    rng = np.random.default_rng(3)
    acc, ref = nka_amd.nka().init(n, m), nka_amd.nka().init(n, m)
    xs = [rng.standard_normal(n) for _ in range(m + 6)]
    for x in xs[:m + 2]:
        acc.accel_update(torch.from_numpy(x.copy()).cuda())
        ref.accel_update(torch.from_numpy(x.copy()).cuda())
    torch.cuda.synchronize()
    assert acc.capture_safe() and acc.list_bound() == m + 1
    buf = torch.from_numpy(xs[m + 2].copy()).cuda()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        acc.set_stream(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            acc.accel_update(buf)
    outs = []
    for x in xs[m + 2:m + 5]:
        buf.copy_(torch.from_numpy(x))
        g.replay()
        torch.cuda.synchronize()
        outs.append(buf.clone())
    for x, o in zip(xs[m + 2:m + 5], outs):
        fr = torch.from_numpy(x.copy()).cuda()
        ref.accel_update(fr)
        assert torch.equal(fr, o)
    # the word has stopped: the bound is the plain count again, and eager calls after the replays are right
    assert acc.list_bound() == m + 1
    fa, fr = torch.from_numpy(xs[m + 5].copy()).cuda(), torch.from_numpy(xs[m + 5].copy()).cuda()
    with torch.cuda.stream(s):
        acc.accel_update(fa)
    ref.accel_update(fr)
    torch.cuda.synchronize()
    assert torch.equal(fa, fr) and acc.state_digest() == ref.state_digest()


def test_a_copy_of_a_sharded_accelerator_refuses_to_run_without_an_all_reduce(torch_cuda):
    """nka_hip_clone leaves the built-in RCCL communicator with the original.  The copy must not silently form
    rank-local sums: accel_update fails with NKA_HIP_ECOMM until the caller gives it a communicator or a
    hook -- NULL included, which says "this copy really is single-rank"."""
    import nka_amd
    torch = torch_cuda
    n, m = 5000, 3
    rng = np.random.default_rng(11)
    a = nka_amd.nka().init(n, m)
    a.use_rccl(nka_amd.nka.rccl_unique_id(), 1, 0)            # a one-rank communicator: the sharded path on one GPU
    assert a.comm_info() == (1, 0)
    xs = [rng.standard_normal(n) for _ in range(4)]
    for x in xs[:2]:
        a.accel_update(torch.from_numpy(x.copy()).cuda())
    b = a.copy()
    assert b.comm_info() == (0, -1)
    with pytest.raises(nka_amd.NKAError, match="copy of a sharded"):
        b.accel_update(torch.from_numpy(xs[2].copy()).cuda())
    assert b.state_digest() == a.state_digest()               # the refused call changed nothing
    b.set_dot_prod(None)                                      # an explicit choice: single rank
    fa, fb = torch.from_numpy(xs[2].copy()).cuda(), torch.from_numpy(xs[2].copy()).cuda()
    a.accel_update(fa)
    b.accel_update(fb)
    assert torch.equal(fa, fb)
    c = b.copy()                                              # a copy of an unsharded accelerator runs at once
    c.accel_update(torch.from_numpy(xs[3].copy()).cuda())


def test_a_rank_with_a_bad_pointer_still_joins_the_collective_of_a_parallel_reduction(torch_cuda):
    """A parallel-aware reduction hook of the device vectors (row e'): a rank whose own arguments fail the pointer check
    must report its error AND take part in the collective (with zeros), or its peers wait for ever."""
    import ctypes as C
    import nka_amd
    from nka_amd import _lib
    torch = torch_cuda
    L = nka_amd.load()
    h = C.c_void_p()
    assert L.nka_hip_vec_workspace_create(C.byref(h), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    calls = []

    def hook(_ctx, vals, count):
        calls.append([vals[i] for i in range(count)])
        return 0
    cb = _lib.HOST_ALLREDUCE_FN(hook)
    assert L.nka_hip_vec_set_host_allreduce(h, cb, None) == 0
    n = 1 << 22          # (32 MB: beyond any segment torch's caching allocator would have put a 4-element tensor into)
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    short = torch.ones(4, dtype=torch.float64, device="cuda")               # too short for n elements
    out = C.c_double(-1.0)
    assert L.nka_hip_vec_dot(h, n, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.byref(out)) == 0
    assert out.value == float(n) and len(calls) == 1
    rc = L.nka_hip_vec_dot(h, n, C.c_void_p(x.data_ptr()), C.c_void_p(short.data_ptr()), C.byref(out))
    assert rc != 0 and b"shorter" in L.nka_hip_last_error()
    assert len(calls) == 2 and calls[1] == [0.0]              # joined, with zeros
    ys = (C.c_void_p * 3)(x.data_ptr(), short.data_ptr(), x.data_ptr())
    vals = (C.c_double * 3)()
    rc = L.nka_hip_vec_dot_many(h, n, C.c_void_p(x.data_ptr()), ys, 3, vals)
    assert rc != 0 and len(calls) == 3 and calls[2] == [0.0, 0.0, 0.0]
    L.nka_hip_vec_workspace_destroy(h)


@pytest.mark.parametrize("flavor", [0, 1, 2])
@pytest.mark.parametrize("n,m", [(100003, 6), (40960, 20), (513, 3), (300007, 40)])
def test_out_of_place_updates_give_the_same_bits_with_two_stores_less(torch_cuda, oracle, flavor, n, m):
    """nka_hip_accel_update_swap: the caller's buffer becomes w_new, the accelerated f is stored once (as v_new) and
    lent to the caller.  Against an accelerator driven through the in-place entry on the same inputs: every output
    bit, the replicated state and the stored vectors of the live entries -- through growth, capacity and dependence
    drops, a repeated input (s == 0), relax, restart, a deep copy in mid-stream and in-place calls mixed in."""
    import nka_amd
    torch = torch_cuda
    rng = np.random.default_rng(7 * m + flavor)
    basis = rng.standard_normal((3, n))
    X = [rng.standard_normal(3) @ basis if t % 6 == 4 else rng.standard_normal(n) for t in range(m + 14)]
    X[m + 4] = X[m + 3].copy()
    a = nka_amd.nka().init(n, m, flavor=flavor)        # in place
    b = nka_amd.nka().init(n, m, flavor=flavor)        # out of place
    ora = oracle.OracleNKA(n, m, flavor)
    buf = torch.empty(n, dtype=torch.float64, device="cuda")
    seen = set()
    for t, x in enumerate(X):
        fo = x.copy()
        ora.accel_update(fo)
        fa = torch.from_numpy(x.copy()).cuda()
        a.accel_update(fa)
        if t % 7 == 5:                                  # an in-place call in between: the entries can be mixed
            fb = torch.from_numpy(x.copy()).cuda()
            b.accel_update(fb)
            acc = fb
        else:
            buf.copy_(torch.from_numpy(x))
            seen.add(buf.data_ptr())
            buf, acc = b.accel_update_swap(buf)
            assert acc.data_ptr() != buf.data_ptr()
        if t % 3 == 0:
            torch.cuda.synchronize()                    # (sometimes the record is fresh, sometimes the call must wait)
        assert torch.equal(acc, fa), (t, float((acc - fa).abs().max()))
        assert a.state_digest() == b.state_digest(), t
        assert b.state().list_order() == ora.state().list_order(), t
        if t == m + 6:
            a.relax(); b.relax(); ora.relax()
        if t == m + 9:
            a.restart(); b.restart(); ora.restart()
        if t == m + 2:
            c = b.copy()                                # deep copy of an accelerator whose vectors live in foreign buffers
            sa, sc = a.state(), c.state()
            assert c.state_digest() == a.state_digest()
            for k in sa.list_order():
                assert np.array_equal(a.w(k), c.w(k)) and np.array_equal(a.v(k), c.v(k)), (t, k)
                assert np.array_equal(a.w(k), b.w(k)) and np.array_equal(a.v(k), b.v(k)), (t, k)
            fc, fa2 = torch.from_numpy(X[0].copy()).cuda(), torch.from_numpy(X[0].copy()).cuda()
            c.accel_update(fc)
            a2 = a.copy()
            a2.accel_update(fa2)
            assert torch.equal(fc, fa2)
    assert len(seen) >= 3 and a.defined() and b.defined()        # the buffers really circulate


def test_out_of_place_update_argument_checks(torch_cuda):
    import ctypes as C
    import nka_amd
    torch = torch_cuda
    a = nka_amd.nka().init(1000, 3)
    with pytest.raises(nka_amd.NKAError):
        a.accel_update_swap(torch.zeros(999, dtype=torch.float64, device="cuda"))
    odd = torch.zeros(1001, dtype=torch.float64, device="cuda")[1:]          # not 16-byte aligned
    with pytest.raises(nka_amd.NKAError, match="aligned"):
        a.accel_update_swap(odd)
    a.set_host_dot(lambda x, y: float(np.dot(x, y)))
    with pytest.raises(nka_amd.NKAError, match="user dot product"):
        a.accel_update_swap(torch.zeros(1000, dtype=torch.float64, device="cuda"))
    a.set_host_dot(None)
    f = torch.ones(1000, dtype=torch.float64, device="cuda")
    buf, acc = a.accel_update_swap(f)
    assert torch.equal(acc, torch.ones_like(acc))         # the first update returns its input
    buf2, acc2 = a.accel_update_swap(buf)                 # the lent buffer comes back as the next input: the normal protocol
    assert buf2.data_ptr() not in (buf.data_ptr(), acc2.data_ptr()) and a.num_vec() == 1
    with pytest.raises(nka_amd.NKAError, match="read only"):
        a.accel_update_swap(acc2)                         # the accelerated f is the stored v of the pending pair: not an input
    with pytest.raises(nka_amd.NKAError, match="read only"):
        a.accel_update(acc2)                              # ... nor the in-place entry's f (it would be overwritten while read)
    assert a.num_vec() == 1 and a.defined()
