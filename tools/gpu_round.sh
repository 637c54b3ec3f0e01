mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_vec_ops_gpu.py tests/test_fortran_front_end.py -m gpu -x -q 2>&1 | tail -4
B=nka_amd/fortran/build
for m in 5 10 20; do
$B/nka_vector_driver bench 4 10000000 $m 30 0 | sed -n 2p
NKA_HIP_VEC_WIN=0 $B/nka_vector_driver bench 4 10000000 $m 30 0 | sed -n 2p
done
