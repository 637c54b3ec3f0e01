"""The soak's recorded exceedances as regression fixtures (re-based in round 6).
tests/golden/soak_cases.json names the sequences of the soak runs that ended beyond the TYPICAL bar of the numerical contract --
by generator call (kind, seed, world) and the fast sum mode they were recorded in, not by arrays.  Each one is replayed here
through the same code the soak ran (tools/fuzz_gpu.py), in that mode: decisions must still be exact after every call, and the
sequence must end within the contract's HARD line (include/nka_hip.h item 3: 8 x the reference's own distance from the truth
beyond one tile; 32 x with a base of 1e-11 within) -- the line no record of any soak has exceeded.  Until round 5 each case was
capped at `recorded ratio x 1.25`: a test that passes on a known out-of-rule figure; the recorded ratio is now reported, not
asserted."""
import json
import os
import re
import subprocess
import sys

import pytest

import parity_util as P
import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

with open(os.path.join(ROOT, "tests", "golden", "soak_cases.json")) as _fh:
    CASES = json.load(_fh)["cases"]


def _ids(kind):
    return [pytest.param(c, id=f"{c['kind']}-seed{c['seed']}-{c['elements']}el") for c in CASES if c["kind"] == kind]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _judge(case, dev, ref):
    assert ref > 0.0
    ratio = dev / ref
    tiny = case["elements"] <= P.TINY_N
    hard = max(1e-11 if tiny else 1e-12, P.truth_hard(case["elements"]) * ref)
    print(f"{case['kind']} seed {case['seed']} ({case['elements']} elements, sums {case.get('sums', 'blocked')}): err_dev / err_ref = "
          f"{ratio:.2f} x (recorded {case['ratio']} x); HARD line {hard:.2e}, err_dev {dev:.2e}")
    assert dev <= hard, (case["kind"], case["seed"], f"err_dev {dev:.3e} beyond the HARD line {hard:.3e} (err_ref {ref:.3e}, {ratio:.2f} x)")
    return ratio


def _mode(case, nka_amd):
    return {"blocked": nka_amd.SUMS_BLOCKED, "rounded": nka_amd.SUMS_BLOCKED_ROUNDED}[case.get("sums", "blocked")]


@pytest.mark.parametrize("case", _ids("array"))
def test_array_flavour_soak_case(torch_cuda, oracle, case):
    import fuzz_gpu
    import nka_amd
    key = fuzz_gpu.one_seed(case["seed"], torch_cuda, oracle, P, S, nka_amd, strict=False, sums=_mode(case, nka_amd))   # (decisions: asserted inside)
    assert case["shape"] in key, (key, case["shape"])                                            # the generator still draws this shape
    rec = P.WORST[key]
    _judge(case, rec["err_dev_exact"], rec["err_ref_exact"])


@pytest.mark.parametrize("case", _ids("vector"))
def test_abstract_vector_flavour_soak_case(oracle, tmp_path, case):
    import fuzz_gpu
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "fortran")], check=True)
    # (the abstract-vector flavour keeps its fused default: a recorded case is replayed as its seed draws it)
    key = fuzz_gpu.one_seed_vector(case["seed"], oracle, P, S, str(tmp_path), world=case["world"], strict=False)
    assert case["shape"] in key, (key, case["shape"])
    rec = P.WORST[key]
    _judge(case, rec["err_dev_exact"], rec["err_ref_exact"])


LAUNCHER_ERRORS = ("EADDRINUSE", "address already in use", "RendezvousConnectionError", "failed to listen on", "RendezvousTimeoutError")


def _run_rank_group(cmd, env):
    """One rank group of the sharded soak tool.  A second attempt is made ONLY when the first died in the LAUNCHER -- the local
    port found free was taken before torch.distributed.run bound it, the rendezvous did not come up (as tests/launch_util.py
    does) -- never on any other failure: an intermittent defect of the peer-to-peer mailboxes or of the rank-to-rank sum chain
    must fail the test the first time it shows."""
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    words = p.stdout + p.stderr
    if p.returncode != 0 and any(w in words for w in LAUNCHER_ERRORS) and "FAIL seed" not in words:
        print("LAUNCH REPEATED (rc %d, a launcher error); its last words:\n%s" % (p.returncode, words[-3000:]), flush=True)
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    return p


@pytest.mark.parametrize("case", _ids("sharded"))
def test_sharded_array_flavour_soak_case(case, tmp_path):
    """`world` ranks sharing the GPU, the sums staged through gloo (tools/fuzz_gpu.py --sharded): one record per rank."""
    out = str(tmp_path / "seed.txt")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "--sharded", str(case["world"]), "--first-seed",
           str(case["seed"]), "--seeds", "1", "--out", out]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               NKA_FUZZ_FORCE_SUMS=case.get("sums", "blocked"))          # (the mode the case was recorded in)
    p = _run_rank_group(cmd, env)
    assert p.returncode in (0, 2), p.stdout[-2000:] + p.stderr[-4000:]      # (2: beyond the TYPICAL bar -- that is what is recorded)
    worst = 0.0
    for r in range(case["world"]):
        text = open(f"{out}.rank{r}").read()
        assert "FAIL" not in text, text[-2000:]
        mt = re.search(r"^(?:ok|stop)\s+(fuzz sharded seed %d .*?): dev-exact (\S+) ref-exact (\S+)" % case["seed"], text, re.M)
        assert mt, text[-1000:]
        assert case["shape"] in mt.group(1), (mt.group(1), case["shape"])
        worst = max(worst, _judge(case, float(mt.group(2)), float(mt.group(3))))
    assert worst > 0.0


@pytest.mark.parametrize("case", [c for c in _ids("sharded") if c.values[0]["elements"] > 512 and "rounded" not in c.values[0]["shape"]])
def test_sharded_soak_cases_beyond_one_tile_stay_within_the_rule_with_the_rounded_gram_row(case, tmp_path):
    """The recorded sharded exceedances beyond one tile (n = 1 660 and n = 1 013 over three ranks), replayed with
    NKA_HIP_SUMS_BLOCKED_ROUNDED (the norm first -- a second exchange per update --, the Gram row on the rounded w1'): the one
    deviation of the fast passes that is not "a more accurate sum" is gone, and each rank must end within the rule's factor 2."""
    out = str(tmp_path / "seed.txt")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "--sharded", str(case["world"]), "--first-seed",
           str(case["seed"]), "--seeds", "1", "--out", out]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NKA_FUZZ_FORCE_ROUNDED="1")
    p = _run_rank_group(cmd, env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    for r in range(case["world"]):
        text = open(f"{out}.rank{r}").read()
        mt = re.search(r"^(?:ok|stop)\s+(fuzz sharded seed %d .*?): dev-exact (\S+) ref-exact (\S+)" % case["seed"], text, re.M)
        assert mt and "sums rounded" in mt.group(1), text[-1000:]
        dev, ref = float(mt.group(2)), float(mt.group(3))
        assert dev <= max(1e-12, 2.0 * ref), (case["seed"], r, dev, ref)
