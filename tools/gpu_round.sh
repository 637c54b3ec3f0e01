mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vec_ops_gpu.py tests/test_fortran_front_end.py -m gpu -q --tb=short -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -5
nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 20 0
nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 20 1
for n in 1.25e7 1e6; do python tools/ab_inproc.py --flavor f08 --vlen $n --key pb_pipe --values 0 2 4 --rounds 8 --steps 20; done
python tools/ab_inproc.py --flavor c --vlen 1.25e7 --key pb_pipe --values 0 2 4 --rounds 8 --steps 20
