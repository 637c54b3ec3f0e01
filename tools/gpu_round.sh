B=nka_amd/fortran/build
$B/nka_vector_driver bench 4 10000000 20 30 0
$B/nka_vector_driver benchgrid 6324 6325 20 30 0
$B/nka_vector_driver benchgrid 6324 6325 20 30 1
$B/nka_vector_driver benchgrid 6325 6325 20 30 0
