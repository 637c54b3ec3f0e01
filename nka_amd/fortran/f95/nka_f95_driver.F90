!! Drives the F95 procedural wrappers through a short state-machine scenario and
!! prints num_vec and a checksum per call (compared by tests with the oracle).
program nka_f95_driver
  use nka_type
  implicit none
  integer, parameter :: r8 = selected_real_kind(15), i8 = selected_int_kind(18)
  integer, parameter :: n = 501, mvec = 4, ncalls = 12
  type(nka) :: acc
  real(r8) :: f(n)
  integer(i8) :: x = 1
  integer :: t, i
  call nka_init(acc, n, mvec)
  call nka_set_vec_tol(acc, 0.05_r8)
  if (nka_real_kind(acc) /= kind(f)) stop 'kind mismatch'
  if (nka_vec_len(acc) /= n .or. nka_max_vec(acc) /= mvec) stop 'accessor mismatch'
  do t = 1, ncalls
    do i = 1, n
      x = mod(1103515245_i8*x + 12345_i8, 2147483648_i8)
      f(i) = real(x, r8)/1073741824.0_r8 - 1.0_r8
    end do
    call nka_accel_update(acc, f)
    if (t == 6) call nka_relax(acc)
    if (t == 9) call nka_restart(acc)
    write(*,'(i3,i3,2es25.16)') t, nka_num_vec(acc), sum(f), sqrt(sum(f*f))
  end do
  if (.not. nka_defined(acc)) stop 'not defined'
  call nka_delete(acc)
  if (nka_defined(acc)) stop 'still defined after delete'
end program
