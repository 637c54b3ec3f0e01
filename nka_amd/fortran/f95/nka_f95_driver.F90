!! Drives the F95 procedural wrappers through a short state-machine scenario and
!! prints num_vec and a checksum per call (compared by tests with the oracle).
!! With the argument `dp` every update passes the optional dot-product dummy of
!! the reference's nka_accel_update (src-F95/nka_type.F90:278-291): a sum in
!! REVERSE index order, so that the test can tell it was really used.
module user_dot
  implicit none
  integer, parameter, private :: r8 = selected_real_kind(15)
contains
  pure function reverse_dot(x, y) result(d)
    real(r8), intent(in) :: x(:), y(:)
    real(r8) :: d
    integer :: i
    d = 0.0_r8
    do i = size(x), 1, -1
      d = d + x(i)*y(i)
    end do
  end function
end module

program nka_f95_driver
  use nka_type
  use user_dot
  implicit none
  integer, parameter :: r8 = selected_real_kind(15), i8 = selected_int_kind(18)
  integer, parameter :: n = 501, mvec = 4, ncalls = 12
  type(nka) :: acc
  real(r8) :: f(n)
  integer(i8) :: x = 1
  integer :: t, i, lun
  logical :: with_dp
  character(16) :: arg
  character(256) :: outfile
  outfile = ''
  if (command_argument_count() >= 2) call get_command_argument(2, outfile)   ! raw outputs for a bit-exact check
  if (len_trim(outfile) > 0) open(unit=17, file=trim(outfile), access='stream', form='unformatted', status='replace')
  lun = 17
  with_dp = .false.
  if (command_argument_count() >= 1) then
    call get_command_argument(1, arg)
    with_dp = (trim(arg) == 'dp')
  end if
  call nka_init(acc, n, mvec)
  call nka_set_vec_tol(acc, 0.05_r8)
  if (nka_real_kind(acc) /= kind(f)) stop 'kind mismatch'
  if (nka_vec_len(acc) /= n .or. nka_max_vec(acc) /= mvec) stop 'accessor mismatch'
  do t = 1, ncalls
    do i = 1, n
      x = mod(1103515245_i8*x + 12345_i8, 2147483648_i8)
      f(i) = real(x, r8)/1073741824.0_r8 - 1.0_r8
    end do
    if (with_dp) then
      call nka_accel_update(acc, f, reverse_dot)
    else
      call nka_accel_update(acc, f)
    end if
    if (len_trim(outfile) > 0) write(lun) f
    if (t == 6) call nka_relax(acc)
    if (t == 9) call nka_restart(acc)
    write(*,'(i3,i3,2es25.16)') t, nka_num_vec(acc), sum(f), sqrt(sum(f*f))
  end do
  if (.not. nka_defined(acc)) stop 'not defined'
  call nka_delete(acc)
  if (nka_defined(acc)) stop 'still defined after delete'
end program
