!! Drives the array flavour with a USER dot product installed through the
!! reference's own call (call a%set_dot_prod(dot_prod), src-F08/nka_type.F90:
!! 209-214), exactly as a caller written for the reference would: a procedure
!! pointer to a function with assumed-shape arguments.  The dot product sums in
!! REVERSE index order so that the test can tell it was really used.  Prints
!! num_vec and checksums per call (compared by tests with the oracle's
!! set_dot_prod run on the same inputs).
!!
!!   nka_dp_driver [OUTFILE]        the run above
!!   nka_dp_driver copy OUTFILE     b = a in mid-stream (after call 5): the reference's type has
!!       allocatable components, so its intrinsic assignment is a DEEP copy (F08:154-168) and the
!!       two objects then evolve independently, user dot product included.  From call 6 on `a`
!!       continues on its own input stream and `b` on a second one; both outputs are written
!!       (a's then b's, per call) for a bit-exact comparison with two oracles.
module user_dot
  use, intrinsic :: iso_fortran_env, only: r8 => real64
  implicit none
  integer :: ncalls_dp = 0
contains
  function reverse_dot(x, y) result(d)
    real(r8), intent(in) :: x(:), y(:)
    real(r8) :: d
    integer :: i
    ncalls_dp = ncalls_dp + 1
    d = 0.0_r8
    do i = size(x), 1, -1
      d = d + x(i)*y(i)
    end do
  end function
end module

program nka_dp_driver
  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use nka_type
  use user_dot
  implicit none
  integer, parameter :: n = 501, mvec = 4, ncalls = 12
  abstract interface
    function dp_iface(x, y)
      import :: r8
      real(r8), intent(in) :: x(:), y(:)
      real(r8) :: dp_iface
    end function
  end interface
  procedure(dp_iface), pointer :: my_dp
  type(nka) :: acc, bcc
  real(r8) :: f(n), g(n)
  integer(i8) :: x = 1, y = 7
  integer :: t, i, lun
  character(256) :: outfile
  logical :: copy_mode
  outfile = ''
  copy_mode = .false.
  if (command_argument_count() >= 1) call get_command_argument(1, outfile)   ! raw outputs for a bit-exact check
  if (trim(outfile) == 'copy') then
    copy_mode = .true.
    call get_command_argument(2, outfile)
  end if
  if (len_trim(outfile) > 0) open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
  call acc%init(n, mvec)
  call acc%set_vec_tol(0.05_r8)
  my_dp => reverse_dot
  call acc%set_dot_prod(my_dp)
  do t = 1, ncalls
    do i = 1, n
      x = mod(1103515245_i8*x + 12345_i8, 2147483648_i8)
      f(i) = real(x, r8)/1073741824.0_r8 - 1.0_r8
    end do
    call acc%accel_update(f)
    if (len_trim(outfile) > 0) write(lun) f
    if (copy_mode .and. t > 5) then              ! the copy, on its own inputs
      do i = 1, n
        y = mod(1103515245_i8*y + 12345_i8, 2147483648_i8)
        g(i) = real(y, r8)/1073741824.0_r8 - 1.0_r8
      end do
      call bcc%accel_update(g)
      write(lun) g
      if (t == 8) call bcc%relax
      write(*,'(i3,i3,2es25.16)') -t, bcc%num_vec(), sum(g), sqrt(sum(g*g))
    end if
    if (copy_mode .and. t == 5) then
      bcc = acc                                  ! deep copy (F08:154-168)
      if (bcc%num_vec() /= acc%num_vec()) error stop 'the copy does not hold the same subspace'
    end if
    if (t == 6) call acc%relax
    if (t == 9) call acc%restart
    write(*,'(i3,i3,2es25.16)') t, acc%num_vec(), sum(f), sqrt(sum(f*f))
  end do
  if (copy_mode) then
    if (.not. bcc%defined()) stop 'copy not defined'
  end if
  if (.not. acc%defined()) stop 'not defined'
  if (ncalls_dp == 0) error stop 'the user dot product was never called'
end program
