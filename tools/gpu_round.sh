for r in 1 2 3; do
  echo "win   : $(nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 30 0 | sed -n 2p)   compact $(nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 30 1 | sed -n 2p)"
  echo "no win: $(NKA_HIP_VEC_WIN=0 nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 30 0 | sed -n 2p)   compact $(NKA_HIP_VEC_WIN=0 nka_amd/fortran/build/nka_vector_driver bench 4 10000000 20 30 1 | sed -n 2p)"
done
