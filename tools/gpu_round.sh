timeout 2000 python -m pytest tests/test_hip_parity.py tests/test_hip_round2.py tests/test_hip_fullsize.py tests/test_fortran_front_end.py -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -3
bash tools/sweep.sh f08 2>&1 | tail -16
