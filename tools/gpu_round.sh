timeout 1500 python -m pytest tests/test_example_dev_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
