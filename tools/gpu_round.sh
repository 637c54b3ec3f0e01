B=nka_amd/fortran/build
for r in 1 2 3; do
$B/nka_vector_driver bench 4 10000000 20 30 0 | sed -n 2p
NKA_HIP_VEC_PAIR_RING4=1 $B/nka_vector_driver bench 4 10000000 20 30 0 | sed -n 2p
done
$B/nka_vector_driver bench 4 2500000 20 30 0 | sed -n 2p
NKA_HIP_VEC_PAIR_RING4=1 $B/nka_vector_driver bench 4 2500000 20 30 0 | sed -n 2p
NKA_HIP_VEC_WIN=0 $B/nka_vector_driver bench 4 2500000 20 30 0 | sed -n 2p
