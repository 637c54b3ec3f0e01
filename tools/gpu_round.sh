mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vec_ops_gpu.py tests/test_hip_parity.py tests/test_example_dev_gpu.py -m gpu -q --tb=short 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -4
for r in 1 2 3; do
  echo "checks on (cached): $(nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 30 0 | sed -n 2p)"
  echo "checks off        : $(NKA_HIP_CHECK_POINTERS=0 nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 30 0 | sed -n 2p)"
done
