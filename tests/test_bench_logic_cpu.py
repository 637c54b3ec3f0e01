"""Host logic of bench.py that needs no GPU: the byte model, the physical
roofline block (every `frac` = bytes moved / time / 8 TB/s), and the contract
ratio kept apart from it."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_byte_model_matches_the_schedule(bench):
    # PA reads w1, f and the L stored w; PB reads f + pairs and writes 5 streams
    assert bench.words_moved("f08", 20, 20) == {"PA_k_dots": 22, "PB_k_combine": 46}
    assert bench.words_moved("c", 20, 20) == {"PA_k_dots": 22, "PB_k_combine": 27}      # compact: one vector per pair
    assert bench.words_moved("f08vec", 10, 10) == {"PA_k_dots": 12, "PB_k_combine": 26}
    # whole update: 8n(8+L+2k) for F08, 8n(9+L+k) compact
    for L in (5, 10, 20):
        assert sum(bench.words_moved("f08", L, L).values()) == 8 + 3 * L
        assert sum(bench.words_moved("c", L, L).values()) == 9 + 2 * L


@pytest.mark.parametrize("flavor,pa_ms,pb_ms", [("c", 2.7, 3.6), ("f08", 2.7, 6.5), ("c", 1.0, 1.2)])
def test_roofline_block_is_physical(bench, flavor, pa_ms, pb_ms):
    n, m = 10**8, 20
    mean = [pa_ms, 0.03, pb_ms, pa_ms + 0.03 + pb_ms + 0.01]
    r = bench.roofline_block(flavor, n, m, mean, {"pure_read_22_streams_GBps": 7100.0})
    w = bench.words_moved(flavor, m, m)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    dom = "PB_k_combine" if pb_ms >= pa_ms else "PA_k_dots"
    assert r["kernel"] == dom
    # achieved = bytes the launch moves / its duration; frac = achieved / peak
    want = 8.0 * n * w[dom] / (mean[2 if dom == "PB_k_combine" else 0] * 1e-3) / 1e9
    assert r["achieved"] == pytest.approx(want)
    assert r["frac"] == pytest.approx(want / 8000.0)
    moved = 8.0 * n * sum(w.values())
    assert r["whole_update"]["bytes_moved"] == moved
    assert r["whole_update"]["frac"] == pytest.approx(moved / (mean[3] * 1e-3) / 1e9 / 8000.0)
    assert r["bytes_moved_per_update"] == moved
    # the same figures as flat scalars (what a scalars-only record of the line keeps)
    assert r["whole_update_frac"] == pytest.approx(r["whole_update"]["frac"])
    assert r["PA_k_dots_frac"] == pytest.approx(r["kernels"]["PA_k_dots"]["frac"])
    assert r["PB_k_combine_frac"] == pytest.approx(r["kernels"]["PB_k_combine"]["frac"])
    assert r["probe_ceiling_pure_read_GBps"] == 7100.0
    if flavor == "c":
        # compact storage moves FEWER bytes than B_alg: NO contract figure of any kind on its line --
        # nothing a reader could divide by the time and read as > 100 % of the peak
        assert not [k for k in r if k.startswith("contract")]
    else:
        # the contract's B_alg never enters a fraction: it is a ratio and a work rate
        assert r["contract_bytes_per_update"] == 8.0 * n * (11 + 3 * m)
        assert r["contract_bytes_ratio"] == pytest.approx(8.0 * n * (11 + 3 * m) / moved)
        assert r["contract_GBps"] == pytest.approx(r["whole_update"]["achieved"] * r["contract_bytes_ratio"])
        assert r["contract_frac_of_peak"] == pytest.approx(r["contract_GBps"] / 8000.0) and r["contract_bytes_ratio"] < 1.05
    if pb_ms > 2.0:                          # realistic timings: below the peak
        assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["whole_update"]["frac"] <= 1.0
        for k in ("PA_k_dots", "PB_k_combine"):
            assert 0.0 < r["kernels"][k]["frac"] <= 1.0


def test_reference_rounding_entry_carries_the_contract_comparable_figures(bench):
    """the src-F08-rounding measurement of the same run sits INSIDE `roofline` of the compact line -- as
    a nested object and as flat scalars -- so that a record which keeps only scalars still shows the BASELINE.md-comparable
    fraction next to the compact flavour's physical one."""
    n, m = 10**8, 20
    mean = [2.55, 0.017, 6.16, 8.71]
    also = {"value": 1e3 / 8.708, "unit": "updates/s", "ms_per_step": 8.708, "steady_state": True,
            "roofline": bench.roofline_block("f08", n, m, mean)}
    nested, flat = bench.reference_rounding_entry(also)
    b_alg = 8.0 * n * (11 + 3 * m)
    assert nested["contract_frac_of_peak"] == pytest.approx(b_alg / 8.71e-3 / 1e9 / 8000.0)
    assert nested["physical_frac"] == pytest.approx(8.0 * n * 68 / 8.71e-3 / 1e9 / 8000.0)
    assert set(nested["kernel_fracs"]) == {"PA_k_dots", "PB_k_combine"} and nested["value"] == also["value"]
    assert flat["reference_rounding_contract_frac_of_peak"] == nested["contract_frac_of_peak"]
    assert flat["reference_rounding_updates_per_s"] == also["value"]
    assert all(isinstance(v, float) for v in flat.values())                 # scalars only
    assert 0.7 < flat["reference_rounding_contract_frac_of_peak"] < 0.9 and flat["reference_rounding_PB_frac"] < 1.0


def test_committed_pmc_traffic_agrees_with_the_byte_model(bench):
    """profiles/r*/pmc_traffic_*.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE) vs 8n x words."""
    n, m = 10**8, 20
    for flavor in ("c", "f08"):
        pm, src = bench.pmc_traffic(flavor, n, m)
        assert pm is not None, "no committed PMC summary for the headline workload"
        w = bench.words_moved(flavor, m, m)
        for key, name in (("k_dots", "PA_k_dots"), ("k_combine", "PB_k_combine")):
            k = pm["kernels"][key]
            assert (k["read_bytes"] + k["write_bytes"]) == pytest.approx(8.0 * n * w[name], rel=2e-3), (src, key)


# ---- the plain launch form: `python bench.py --gpus N` supervises a child rank group (VERDICT r3 task 1) ----------
FAKE = os.path.join(ROOT, "tests", "_fake_rank_script.py")


def _launch(bench, capsys, extra, timeout=60, fallback=True):
    argv = ["--gpus", "2"] + extra
    args = bench.parse(["--gpus", "2", "--launch-timeout", str(timeout)] + ([] if fallback else ["--no-fallback"]))
    rc = bench.launch_ranks(args, argv, script=FAKE)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    return rc, lines


def test_plain_form_relays_rank_zeros_line_and_exit_code(bench, capsys):
    import json
    rc, lines = _launch(bench, capsys, ["--mode", "ok"])
    assert rc == 0 and len(lines) == 1                     # ONE line, the chatter is not relayed to stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "launch" not in d


def test_plain_form_second_attempt_names_the_fallback_hook(bench, capsys):
    import json
    rc, lines = _launch(bench, capsys, ["--mode", "fail-unless-staged"])
    assert rc == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["parallelism"] == "all-reduce=staged" and d["config"]["backend"] == "gloo"
    assert "exit code" in d["launch"]["first_attempt"] and "ncclCommInitRank" in d["launch"]["last_stderr_line"]
    # without the fallback the failure is the result: no line, the child's code
    rc, lines = _launch(bench, capsys, ["--mode", "fail-unless-staged"], fallback=False)
    assert rc != 0 and lines == []
    # a failure a retry cannot cure (too few GPUs) is not retried
    rc, lines = _launch(bench, capsys, ["--mode", "noretry"])
    assert rc != 0 and lines == []
    # ... nor one that shows no sign of the communication path (an out-of-memory rank, an exception elsewhere, a
    # build failure would only fail again, slowly): the staged second attempt WOULD have succeeded here, and is not made
    rc, lines = _launch(bench, capsys, ["--mode", "fail-other-unless-staged"])
    assert rc != 0 and lines == []


def test_plain_form_watchdog_kills_the_whole_rank_group(bench, capsys, tmp_path):
    import json
    import time
    pidfile = str(tmp_path / "pid")
    t0 = time.time()
    rc, lines = _launch(bench, capsys, ["--mode", "hang", "--pidfile", pidfile], timeout=8, fallback=False)
    assert rc == 124 and lines == [] and time.time() - t0 < 60
    pids = [int(open(f"{pidfile}.{r}").read()) for r in range(2)]
    time.sleep(0.5)
    for pid in pids:                                        # the RANKS are gone, not just torchrun
        alive = os.path.exists(f"/proc/{pid}") and "Z" not in open(f"/proc/{pid}/stat").read().split(")")[1].split()[0]
        assert not alive, pid
    # a hang in the first attempt, a result from the second
    rc, lines = _launch(bench, capsys, ["--mode", "hang-unless-staged"], timeout=8)
    assert rc == 0 and json.loads(lines[0])["launch"]["first_attempt"].startswith("watchdog expired")


def test_roofline_block_at_a_shrunk_subspace(bench):
    """--workload drops: the bytes follow the ACTUAL list (L stored vectors read by PA, k pairs combined by PB)."""
    n, m, D = 10**8, 20, 12
    r = bench.roofline_block("c", n, m, [1.7, 0.02, 2.4, 4.13], L=D, k=D)
    assert r["kernels"]["PA_k_dots"]["words_per_element"] == 2 + D
    assert r["kernels"]["PB_k_combine"]["words_per_element"] == D + 2 + 5
    assert r["whole_update"]["bytes_moved"] == 8.0 * n * (9 + 2 * D)
    assert r["traffic"] is None                      # the committed full-subspace PMC summary does not apply


def test_cpu_baseline_hands_every_output_to_the_twin(bench):
    """The CPU leg shows every output of the compiled reference (or of the port) to `after_update` -- that is where the
    device's reference-order twin compares itself with it, bit for bit, in the headline run."""
    import numpy as np
    seen = []
    out = bench.cpu_baseline(3, 200, timed=2, after_update=lambda t, f, kind: seen.append((t, kind, np.array(f, copy=True))))
    assert [t for t, _, _ in seen] == list(range(3 + 2 + 2))
    assert all(kind == out["kind"] for _, kind, _ in seen) and out["kind"] in ("reference", "port")
    assert all(f.shape == (200,) and np.isfinite(f).all() for _, _, f in seen)
    # (and the outputs are the accelerated ones, not the inputs: once the subspace is there they differ from the generator's vectors)
    from nka_amd import synth
    assert not np.array_equal(seen[-1][2], synth.fill_numpy(bench.SEED, seen[-1][0], 0, 200, 200))


def test_a_rank_group_launch_that_finds_its_port_taken_is_repeated_with_another():
    """tests/launch_util.py: between probing a free port and the launcher's bind another process can take it (seen on a GPU
    box: EADDRINUSE in the static rendezvous failed a test that had nothing to do with ports)."""
    import sys
    from launch_util import run_ranks
    script = ("import sys; p = sys.argv[sys.argv.index('--master-port') + 1]; "
              "sys.exit(0) if p != '1234' else (sys.stderr.write('... code: -98, name: EADDRINUSE, message: address already in use'), sys.exit(1))")
    p = run_ranks([sys.executable, "-c", script, "--master-port", "1234"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0
    other = run_ranks([sys.executable, "-c", "import sys; sys.stderr.write('boom'); sys.exit(3)", "--master-port", "1234"],
                      capture_output=True, text=True, timeout=60)
    assert other.returncode == 3 and other.stderr == "boom"        # (any other failure comes back as it is, once)


def _as_a_truncating_record_keeps_it(obj, keys=19, chars=120):
    """What BENCH_r05.parsed kept of round 5's line: of `roofline` only scalar entries, the first `keys` of them, and of
    every string its first `chars` characters."""
    kept = {}
    for k, v in obj.items():
        if isinstance(v, (dict, list)):
            continue
        if len(kept) == keys:
            break
        kept[k] = v[:chars] if isinstance(v, str) else v
    return kept


def test_the_line_survives_a_record_that_keeps_19_scalars_and_120_characters(bench):
    """the contract-comparable src-F08-rounding figures, the per-pass fractions and the same-run ceiling
    must come BEFORE the prose and the per-kernel extras of `roofline`, and the config strings must say what they have to
    say within 100 characters."""
    n, m = 10**8, 20
    probe = {"mix_22R_5W_GBps": 6300.0, "mix_42R_5W_GBps": 6100.0, "pure_read_22_streams_GBps": 7100.0}
    pm = {"kernels": {"k_dots": {"read_bytes": 17.6e9, "write_bytes": 1e5}, "k_combine": {"read_bytes": 17.6e9, "write_bytes": 4.0e9}},
          "hbm_bytes_per_update": 39.2e9}
    rl = bench.roofline_block("c", n, m, [2.52, 0.017, 3.44, 5.98], probe, None, pm)
    also = {"flavor": bench.FLAVOR_TEXT["f08"], "value": 116.9, "unit": "updates/s", "ms_per_step": 8.55, "steady_state": True,
            "roofline": bench.roofline_block("f08", n, m, [2.53, 0.017, 5.98, 8.55])}
    nested, flat = bench.reference_rounding_entry(also)
    rl["reference_rounding"] = nested
    rl.update(flat)
    rl["sum_mode_blocked_rounded_updates_per_s"] = 151.0
    out = {"value": 166.8, "config": {"workload": "BASELINE configs[2]: synthetic n=100000000 (global), mvec=20, fp64, subspace full (num_vec=20)",
                                      "flavor": bench.FLAVOR_TEXT["c"],
                                      "flavor_note": "the flavour `call a%init(vlen, mvec)` (Fortran), nka_init (F95) and nka().init "
                                                     "(Python) run when the caller names none (include/nka_hip.h: NKA_HIP_FLAVOR_DEFAULT)",
                                      "parallelism": "contiguous n-slices over 1 GPU(s); all-reduce=none"},
           "roofline": rl}
    out = bench.finish_line(out)
    kept = _as_a_truncating_record_keeps_it(out["roofline"])
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "whole_update_frac", "PA_k_dots_frac", "PB_k_combine_frac",
                "frac_of_probe_ceiling", "reference_rounding_updates_per_s", "reference_rounding_contract_frac_of_peak",
                "reference_rounding_PB_frac", "probe_ceiling_dominant_mix_GBps", "traffic_source"):
        assert key in kept, (key, list(kept))
    assert kept["reference_rounding_updates_per_s"] == 116.9
    assert abs(kept["reference_rounding_contract_frac_of_peak"] - 56.8e9 / 8.55e-3 / 8e12) < 1e-3
    assert abs(kept["frac_of_probe_ceiling"] - kept["achieved"] / 6300.0) < 1e-12
    assert kept["traffic_source"] == out["roofline"]["traffic_source"] and kept["traffic_source"].startswith("same run")
    # nothing is lost: the long texts live on under *_detail, every entry of the block is still there
    assert set(rl) <= set(out["roofline"]) and "traffic_source_detail" in out["roofline"]
    for key in ("workload", "flavor", "flavor_note", "parallelism"):
        assert len(out["config"][key]) <= bench.SHORT, (key, len(out["config"][key]))
    assert "configs[2]" in out["config"]["workload"] and "mvec=20" in out["config"]["workload"]
    assert "compact" in out["config"]["flavor"]


def test_a_committed_counter_figure_is_labelled_as_not_of_this_run(bench):
    """N > 1 lines take `traffic` from profiles/ -- the line must say so in the first words."""
    r = bench.roofline_block("c", 12_500_000, 20, [0.33, 0.017, 0.42, 0.78])
    out = bench.finish_line({"roofline": r, "config": {}})
    src = out["roofline"]["traffic_source"]
    if out["roofline"]["traffic"] is None:
        assert src is None
    else:
        assert src.startswith("committed file, not this run: profiles/"), src
