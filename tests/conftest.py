import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored; a fresh checkout builds them once
    (hipcc cross-compiles gfx950 without a GPU; ~15 s)."""
    import nka_amd
    from nka_amd import _lib
    if not os.path.exists(nka_amd.lib_path()) or not os.path.exists(_lib.diag_lib_path()):
        nka_amd.build()          # (libnka_hip.so AND libnka_hip_diag.so: the variant tests load the diagnostic build)


@pytest.fixture(autouse=True)
def _truth_rule_per_test():
    """THE parity rule is per call sequence (tests/parity_util.py): whatever a test checked against the extended-
    precision trajectory is judged when the test ends."""
    import parity_util as P
    P.TOUCHED.clear()
    yield
    P.finish()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (checker).  Built on demand with gcc."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Worst observed parity error per test (tests/parity_util.py)."""
    try:
        import parity_util as P
    except Exception:
        return
    if not P.WORST:
        return
    path = P.dump(ROOT)
    tr = terminalreporter
    tr.write_sep("-", "worst observed parity errors (||f_hip - f_ref|| / ||f_in||)")
    for key in sorted(P.WORST):
        r = P.WORST[key]
        piv = "   -  " if r["pivot"] is None else f"{r['pivot']:6.3f}"
        k = f"; spread diagnostic: K {r['k_needed']:.2f}" if r.get("k_needed") else ""
        if r.get("err_dev_exact") is not None:
            # THE rule: device vs the extended-precision trajectory, against the reference's own distance from it
            tr.write_line(f"{key:<78s} dev-exact {r['err_dev_exact']:.2e} | ref-exact {r['err_ref_exact']:.2e} | tol {r['tol']:.1e} "
                          f"({r.get('rule', '-')}; used {r.get('truth_ratio', 0.0):.2f} of the allowance) | dev-ref {r['err']:.2e}, "
                          f"pivot {piv}; {r['checks']} checks{k}")
        else:
            tr.write_line(f"{key:<78s} worst {r['err']:.2e} (tol {r['tol']:.1e}, pivot {piv}, rule: {r.get('rule', '-')}); "
                          f"well-conditioned worst {r['worst_well_conditioned']:.2e}; {r['checks']} checks{k}")
    if path:
        tr.write_line(f"written to {path}")
