"""Everything of this repository that runs WITHOUT a GPU, under AddressSanitizer + UBSan (SURVEY.md
section 5 "race detection / sanitizers" -- the reference's own stand-in is its `defined` invariant check,
/root/reference/src-F08/nka_type.F90:460-524).  GPU AddressSanitizer / XNACK are not available on this pool, so the device
code is held by the parity tests; what a sanitizer CAN see runs here, in `pytest -m "not gpu"`:

  * the checker itself (oracle/*.c: the C restatement, its extended-precision twin, the probe, the example problem) built by
    `make -C oracle asan`, with the oracle's own CPU tests run against that build in a child pytest (libasan preloaded into
    python for the ctypes path; the compiled reference under oracle/_ref is loaded beside it, uninstrumented);
  * the pure host-side arithmetic of libnka_hip.so -- pass widths, launch groups, the decoding of the list word, the
    buffer book of the out-of-place entry: nka_amd/csrc/host_logic.hpp is the very text nka_hip.hip / vec_ops.hip compile --
    against brute-force models (tests/c/host_logic_check.cpp, `make -C nka_amd/csrc hostcheck`);
  * the host all-reduce of the sharded abstract-vector tests (tests/c/shm_allreduce.c) inside the CPU ranks of
    tests/test_vector_sharded.py (Fortran drivers linked with the sanitizer runtime);
  * the C front end (include/nka_c_compat.h + tests/c/nka_c_driver.c) on its failure path: without a GPU it must fail
    loudly in nka_init, and cleanly.

And the proof that the builds ARE instrumented: a planted heap overflow and a planted undefined shift must each end their
run with a sanitizer report (a test that passes on an uninstrumented build would prove nothing)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_DIR = os.path.join(ROOT, "oracle", "_san")
HOSTB = os.path.join(ROOT, "nka_amd", "csrc", "build_host")
REPORT = ("AddressSanitizer", "runtime error:", "LeakSanitizer", "UndefinedBehaviorSanitizer")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def san_build():
    if _libasan() is None:
        pytest.skip("gcc's libasan is not installed")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    import nka_amd
    if not os.path.exists(nka_amd.lib_path()):
        nka_amd.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "nka_amd", "csrc"), "hostcheck"], check=True)
    return True


def _clean(text):
    return not any(mark in text for mark in REPORT)


def test_a_planted_overflow_and_a_planted_undefined_shift_are_caught(san_build):
    """The instrumentation works: the same program passes on buffers of the right size, dies with a heap-buffer-overflow
    report when f is one element short, and with a UBSan report on a shift by 40."""
    exe = os.path.join(SAN_DIR, "planted_overflow")
    ok = subprocess.run([exe, "ok"], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and "num_vec 4" in ok.stdout and _clean(ok.stderr), ok.stderr[-2000:]
    bad = subprocess.run([exe, "plant"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in bad.stderr, (bad.returncode, bad.stderr[-1500:])
    ub = subprocess.run([exe, "shift"], capture_output=True, text=True, timeout=60)
    assert ub.returncode != 0 and "shift exponent 40 is too large" in ub.stderr, (ub.returncode, ub.stderr[-1500:])
    host = subprocess.run([os.path.join(HOSTB, "host_logic_check"), "plant"], capture_output=True, text=True, timeout=120)
    assert host.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in host.stderr and "balanced_widths" in host.stderr, \
        (host.returncode, host.stderr[-1500:])


def test_host_logic_of_the_library_under_sanitizers(san_build):
    """nka_amd/csrc/host_logic.hpp against brute-force models: balanced pass widths (sum, range, no more heavy primes than the
    plain split, nothing written outside the array), launch groups of the vector hooks, the list word (never below the true
    list length; exact once the newest word has arrived; stale and future words ignored; 2^43 updates), the buffer book."""
    p = subprocess.run([os.path.join(HOSTB, "host_logic_check")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout and _clean(p.stderr), (p.returncode, p.stdout[-500:], p.stderr[-3000:])


def test_oracle_cpu_tests_against_the_sanitizer_build_of_the_checker(san_build):
    """tests/test_oracle_golden.py + test_oracle_exact_cpu.py (the restatement against every golden vector, the compiled
    reference, the extended-precision and 60-digit twins) in a child pytest whose oracle is oracle/_san/libnka_oracle.so."""
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               NKA_ORACLE_LIB=os.path.join(SAN_DIR, "libnka_oracle.so"))      # (the parent's threading: a fixture holds numpy norms)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_exact_cpu.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = p.stdout[-2500:] + p.stderr[-2500:]
    assert p.returncode == 0 and " passed" in p.stdout and " failed" not in p.stdout, tail
    assert _clean(p.stdout) and _clean(p.stderr), tail
    # the child really loaded the instrumented library
    chk = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import oracle_py as O; "
                          "print(O.lib()._name)" % ROOT], env=env, capture_output=True, text=True, timeout=120)
    assert chk.stdout.strip().endswith(os.path.join("_san", "libnka_oracle.so")), (chk.stdout, chk.stderr[-1500:])


def test_cpu_ranks_of_the_sharded_vector_flavour_with_the_instrumented_host_allreduce(san_build):
    """tests/test_vector_sharded.py, CPU part (2 and 3 processes, the vector flavour of the accelerator on a CPU vector type,
    parallel-aware reductions through tests/c/shm_allreduce.c), the drivers rebuilt with -fsanitize=address,undefined."""
    env = dict(os.environ, NKA_TEST_SANITIZE="1", ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="1",
               NKA_PARITY_OUT=os.devnull)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_vector_sharded.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = p.stdout[-2500:] + p.stderr[-2500:]
    assert p.returncode == 0 and " passed" in p.stdout, tail
    assert _clean(p.stdout) and _clean(p.stderr), tail


def test_c_front_end_fails_loudly_and_cleanly_without_a_gpu(san_build):
    """The instrumented C driver over include/nka_c_compat.h: on a box without a GPU nka_init must stop the program with the
    library's message -- no CPU path, no sanitizer report on the way there."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the driver runs for real in tests/test_fortran_front_end.py")
    p = subprocess.run([os.path.join(HOSTB, "nka_c_driver_san")], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert p.returncode != 0 and "nka_init failed" in (p.stdout + p.stderr), (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert _clean(p.stdout) and _clean(p.stderr), p.stderr[-2000:]
