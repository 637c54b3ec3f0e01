"""Edge cases of the C ABI beyond the scenario fixtures (round 3): a copy of an accelerator whose scalar state
lives in global memory (mvec > 140), a stream change in mid-sequence, the host-array entry at n = 0 / 1 / 3, the
wrap of the timing ring, storms of relax / restart -- each against the oracle."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _track(torch, acc, ora, x, tag, tol=1e-9):
    f = x.copy()
    ora.accel_update(f)
    ft = torch.from_numpy(x.copy()).cuda()
    acc.accel_update(ft)
    assert acc.num_vec() == ora.num_vec(), (tag, acc.num_vec(), ora.num_vec())
    assert acc.state().list_order() == ora.state().list_order(), tag
    err = np.linalg.norm(ft.cpu().numpy() - f) / max(np.linalg.norm(x), 1e-300)
    assert err < tol, (tag, err)


def test_copy_of_an_accelerator_with_its_state_in_global_memory(torch_cuda, oracle):
    import nka_amd
    rng = np.random.default_rng(0)
    n, m = 300, 141
    a = nka_amd.nka().init(n, m)
    oa, ob = oracle.OracleNKA(n, m, a.flavor()), oracle.OracleNKA(n, m, a.flavor())
    for t in range(12):
        x = rng.standard_normal(n)
        _track(torch_cuda, a, oa, x, ("fill", t))
        ob.accel_update(x.copy())
    b = a.copy()
    assert b.state_digest() == a.state_digest()
    for t in range(6):
        _track(torch_cuda, a, oa, rng.standard_normal(n), ("a", t))
        _track(torch_cuda, b, ob, rng.standard_normal(n), ("b", t))
    assert a.defined() and b.defined()


def test_stream_change_in_mid_sequence(torch_cuda, oracle):
    import nka_amd
    torch = torch_cuda
    rng = np.random.default_rng(1)
    n, m = 40001, 6
    a = nka_amd.nka().init(n, m)
    oa = oracle.OracleNKA(n, m, a.flavor())
    side = torch.cuda.Stream()
    for t in range(20):
        x = rng.standard_normal(n)
        if t == 7:
            a.set_stream(side.cuda_stream)
            a._follow_torch_stream = False
        if t == 13:
            a.set_stream(torch.cuda.current_stream().cuda_stream)
        if 7 <= t < 13:
            f = x.copy()
            oa.accel_update(f)
            with torch.cuda.stream(side):
                ft = torch.from_numpy(x.copy()).cuda()
                a.accel_update(ft)
                side.synchronize()
            assert a.num_vec() == oa.num_vec()
            assert np.linalg.norm(ft.cpu().numpy() - f) / np.linalg.norm(x) < 1e-11
        else:
            torch.cuda.synchronize()
            _track(torch, a, oa, x, ("s", t), tol=1e-11)


@pytest.mark.parametrize("n", [0, 1, 3])
def test_host_array_entry_at_tiny_lengths(torch_cuda, oracle, n):
    import nka_amd
    rng = np.random.default_rng(2)
    a = nka_amd.nka().init(n, 3)
    oa = oracle.OracleNKA(n, 3, a.flavor())
    for t in range(7):
        x = rng.standard_normal(n)
        f = x.copy()
        oa.accel_update(f)
        g = x.copy()
        a.accel_update(g)                      # numpy array: nka_hip_accel_update_host (the reference's own signature)
        assert a.num_vec() == oa.num_vec(), (n, t)
        if n:
            assert np.linalg.norm(g - f) <= 1e-9 * max(np.linalg.norm(f), np.linalg.norm(x)), (n, t, g, f)


def test_timing_ring_wraps(torch_cuda):
    import nka_amd
    rng = np.random.default_rng(3)
    a = nka_amd.nka().init(5000, 4)
    a.set_timing(3)
    for _ in range(11):
        a.accel_update(torch_cuda.from_numpy(rng.standard_normal(5000)).cuda())
    for back in range(3):
        ms = a.timing_ms(back)
        assert all(v >= 0 for v in ms) and ms[3] > 0
    with pytest.raises(nka_amd.NKAError):
        a.timing_ms(3)


def test_relax_and_restart_storm(torch_cuda, oracle):
    import nka_amd
    rng = np.random.default_rng(4)
    a = nka_amd.nka().init(777, 5)
    oa = oracle.OracleNKA(777, 5, a.flavor())
    for t in range(80):
        r = rng.random()
        if r < 0.2:
            a.relax(); oa.relax()
        elif r < 0.3:
            a.restart(); oa.restart()
        elif r < 0.35:
            a.relax(); a.relax(); oa.relax(); oa.relax()
        else:
            _track(torch_cuda, a, oa, rng.standard_normal(777), ("storm", t))
        assert a.num_vec() == oa.num_vec() and a.defined()


def test_distinct_objects_driven_from_concurrent_host_threads(torch_cuda, oracle):
    """The reference's object is plain mutable state and distinct objects are independent (SURVEY.md 8b,
    Threading): four host threads, each with its own accelerator on its own stream, run different sequences at the
    same time (ctypes releases the interpreter lock inside the C ABI); each must track its own oracle -- decisions
    exact, and bit for bit the result of the same sequence run alone afterwards."""
    import threading
    import nka_amd
    torch = torch_cuda
    nthreads, calls = 4, 40
    shapes = [(30011, 5, 0), (4099, 12, 2), (257, 33, 1), (100003, 8, 2)]
    inputs, outs, errors = [], [None] * nthreads, []
    for k, (n, m, fl) in enumerate(shapes):
        rng = np.random.default_rng(500 + k)
        basis = rng.standard_normal((3, n))
        inputs.append([rng.standard_normal(3) @ basis if t % 6 == 4 else rng.standard_normal(n) for t in range(calls)])

    def run(k, record):
        n, m, fl = shapes[k]
        res = []
        with torch.cuda.stream(torch.cuda.Stream()):
            acc = nka_amd.nka().init(n, m, flavor=fl)
            ora = oracle.OracleNKA(n, m, fl)
            for t, x in enumerate(inputs[k]):
                f = x.copy()
                ora.accel_update(f)
                ft = torch.from_numpy(x.copy()).cuda(non_blocking=False)
                acc.accel_update(ft)
                if t == 17:
                    acc.relax(); ora.relax()
                assert acc.num_vec() == ora.num_vec(), (k, t)
                assert acc.state().list_order() == ora.state().list_order(), (k, t)
                res.append(ft.cpu().numpy())
            assert acc.defined()
        record[k] = res

    def guarded(k):
        try:
            run(k, outs)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=guarded, args=(k,)) for k in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    alone = [None] * nthreads
    for k in range(nthreads):
        run(k, alone)
        for t in range(calls):
            assert np.array_equal(outs[k][t], alone[k][t]), (k, t)


def test_register_reductions_build_the_shuffle_butterfly_trees(torch_cuda):
    """The reductions of every kernel (wave_sum, block_reduce_store: gfx950 v_permlane32_swap / v_permlane16_swap and
    DPP row shifts, the butterflies of a block sharing their steps) must add the SAME pairs in the SAME order as one
    __shfl_down butterfly per sum -- the documented fixed order of the inner products: tools/wave_sum_check compares
    65 536 wavefront sums and the block sums for 1...66 accumulators per thread bit for bit on the device."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "wave_sum_check")
    if not os.path.exists(exe):
        pytest.skip("tools/wave_sum_check not built (python -c 'import __graft_entry__ as g; g.build()')")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 13 and all(" 0 of " in ln for ln in lines), p.stdout


def test_capture_safe_is_never_true_where_an_update_touches_the_host(torch_cuda, oracle, monkeypatch):
    """nka_hip_capture_safe: a steady-state update is capturable into a hipGraph -- but not in the debug mode (state read
    back after every update) and not with a user dot product installed (it runs on the host)."""
    import nka_amd
    rng = np.random.default_rng(3)
    n, m = 1000, 3

    def fill(acc):
        for _ in range(m + 3):
            acc.accel_update(torch_cuda.from_numpy(rng.standard_normal(n)).cuda())

    plain = nka_amd.nka().init(n, m)
    fill(plain)
    assert plain.capture_safe()
    hd = nka_amd.nka().init(n, m)
    hd.set_host_dot(lambda x, y: float(np.dot(x, y)))
    fill(hd)
    assert not hd.capture_safe()
    hd.set_host_dot(None)
    assert hd.capture_safe()
    hd.set_dot_prod(lambda ptr, count, stream: None)          # a caller's all-reduce hook: a host callback
    assert not hd.capture_safe()
    hd.set_dot_prod(None)
    assert hd.capture_safe()
    monkeypatch.setenv("NKA_HIP_DEBUG", "1")
    dbg = nka_amd.nka().init(n, m)
    fill(dbg)
    assert not dbg.capture_safe()


def test_an_update_with_a_host_callback_refuses_to_be_captured(torch_cuda):
    """Round 6 (include/nka_hip_ext.h, table of supported combinations): a caller's all-reduce hook is a HOST callback -- a
    captured update would call it once, at capture time, and every replay would run on stale sums with NKA_HIP_OK.  The
    update now refuses (NKA_HIP_EINVAL) while the stream is capturing; eager updates on the same handle go on working, and a
    handle without the hook captures as before."""
    import nka_amd
    torch = torch_cuda
    rng = np.random.default_rng(7)
    n, m = 4099, 3
    calls = [0]

    def hook(ptr, count, stream):
        calls[0] += 1                        # (one rank: the sum of one contribution is the contribution)

    acc = nka_amd.nka().init(n, m)
    acc.set_dot_prod(hook)
    side = torch.cuda.Stream()
    static = torch.empty(n, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(m + 3):
            static.copy_(torch.from_numpy(rng.standard_normal(n)))
            acc.accel_update(static)
    torch.cuda.synchronize()
    assert calls[0] >= m + 3 - 1 and not acc.capture_safe()      # (one exchange per update in the fast mode, two by default)
    per_update = calls[0] // (m + 3 - 1)
    before = calls[0]
    g = torch.cuda.CUDAGraph()
    with pytest.raises(nka_amd.NKAError, match="cannot be captured"):
        with torch.cuda.graph(g, stream=side):
            acc.accel_update(static)
    assert calls[0] == before                # refused BEFORE the hook ran
    torch.cuda.synchronize()
    with torch.cuda.stream(side):            # eager updates carry on (the failed call changed nothing)
        static.copy_(torch.from_numpy(rng.standard_normal(n)))
        acc.accel_update(static)
    torch.cuda.synchronize()
    assert calls[0] == before + per_update and acc.num_vec() == m and acc.defined()


def test_a_buffer_lent_back_to_the_caller_is_no_longer_held_by_the_library(torch_cuda):
    """the host's set of buffers it has taken over only ever grew.  A caller's buffer X that was handed in, later
    displaced and LENT BACK for the caller's next input belongs to the caller again; a different buffer Y that overlaps X's
    range at another base address was then refused as "held by the library".  Now X leaves the set when it is handed out: Y is
    accepted, X itself (still lent) is accepted, and a buffer the library really holds is still refused."""
    import nka_amd
    torch = torch_cuda
    rng = np.random.default_rng(11)
    n, m = 2048, 2
    acc = nka_amd.nka().init(n, m)
    big = torch.zeros(n + 64, dtype=torch.float64, device="cuda")
    X = big[:n]
    X.copy_(torch.from_numpy(rng.standard_normal(n)))
    x_addr = X.data_ptr()
    held = [X]
    buf, _ = acc.accel_update_swap(X, views=False)
    came_back = False
    for _ in range(3 * (m + 2)):
        t = acc._view(buf)
        t.copy_(torch.from_numpy(rng.standard_normal(n)))
        held.append(t)
        buf, _ = acc.accel_update_swap(t, views=False)
        if buf == x_addr:
            came_back = True
            break
    assert came_back, "the caller's first buffer never came back as the free buffer"
    torch.cuda.synchronize()
    Y = big[2:n + 2]                          # overlaps X, other base, 16-byte aligned; X is the caller's again
    Y.copy_(torch.from_numpy(rng.standard_normal(n)))
    buf2, _ = acc.accel_update_swap(Y, views=False)          # was: NKAError "already held by the library"
    assert buf2 != Y.data_ptr()
    torch.cuda.synchronize()
    with pytest.raises(nka_amd.NKAError, match="held by the library"):
        acc.accel_update_swap(big[4:n + 4], views=False)    # overlaps Y, which the library holds NOW
    assert acc.defined() and acc.num_vec() == m


def test_the_environment_chooses_the_sum_order_a_handle_starts_with(torch_cuda):
    """NKA_HIP_SUMS = auto | rounded | blocked | reference (round 6): for callers that cannot call nka_hip_set_sum_order -- the
    reference's own programs relinked against the front ends.  Seen through the exchanges a one-slice 'sharded' handle makes
    per update: two by default (the norm, then the rows), one in the fast mode; an unknown word is refused at create."""
    import subprocess
    import sys
    prog = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import nka_amd\n"
            "counts = []\n"
            "a = nka_amd.nka().init(4099, 3)\n"
            "a.set_dot_prod(lambda ptr, count, stream: counts.append(count))\n"
            "rng = np.random.default_rng(1)\n"
            "for _ in range(2): a.accel_update(torch.from_numpy(rng.standard_normal(4099)).cuda())\n"
            "torch.cuda.synchronize(); print('COUNTS', counts)\n" % ROOT)
    want = {None: "[1, 7]", "auto": "[1, 7]", "rounded": "[1, 7]", "blocked": "[8]", "BLOCKED": "[8]"}
    for value, text in want.items():
        env = {k: v for k, v in os.environ.items() if k != "NKA_HIP_SUMS"}
        if value is not None:
            env["NKA_HIP_SUMS"] = value
        p = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "COUNTS " + text in p.stdout, (value, p.stdout[-500:], p.stderr[-1500:])
    p = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, NKA_HIP_SUMS="fastest"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "NKA_HIP_SUMS" in p.stderr, (p.stdout[-300:], p.stderr[-1500:])
