timeout 900 python -m pytest tests/test_hip_round2.py -m gpu -q --tb=short -k "variant" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -12
