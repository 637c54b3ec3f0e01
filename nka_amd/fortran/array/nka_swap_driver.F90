!! nka_swap_driver -- the out-of-place update and the list bound through the FORTRAN front end
!! (module nka_type -> iso_c_binding -> libnka_hip.so), held to the in-place entry bit for bit.
!!
!!   nka_swap_driver [N [MVEC [CALLS [FLAVOR]]]]      defaults 100003 6 20 2
!!
!! Two accelerators see the same inputs: A through accel_update_dev (in place), B through
!! accel_update_swap -- the buffer that holds f is handed over (it becomes the storage of w of the new
!! pair), the accelerated f comes back as a read-only view, and a free buffer comes back for the next
!! input.  The first ten inputs lie in a 2-dimensional span -- so do their differences, the subspace holds two vectors
!! and every further update takes a dependence drop --, the rest are independent (the list then grows to capacity);
!! one input is repeated (s == 0 -> relax inside the update).  After every call: the two results equal bit for bit, num_vec equal,
!! and -- the program synchronises after every call, like a solver that reads its residual norm --
!! list_bound() == num_vec() + 1, the exact list length, never the padded mvec + 1 of the host's own count.
!! Prints "OK" and the number of calls in which the bound was below the plain count.

program nka_swap_driver

  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use, intrinsic :: iso_c_binding
  use nka_hip_c
  use nka_type
  implicit none

  integer(i8) :: n = 100003_i8
  integer :: mvec = 6, calls = 20, flavor = NKA_HIP_FLAVOR_C
  character(64) :: arg
  type(nka) :: a, b
  type(c_ptr) :: ws, fa, buf, acc
  real(r8), allocatable :: x(:), basis(:,:), ra(:), rb(:), coef(:)
  integer :: t, below, nva, nvb, lb

  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg,*) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg,*) mvec; end if
  if (command_argument_count() >= 3) then; call get_command_argument(3, arg); read(arg,*) calls; end if
  if (command_argument_count() >= 4) then; call get_command_argument(4, arg); read(arg,*) flavor; end if

  call nka_hip_check(nka_hip_vec_workspace_create(ws, 0_c_int32_t, c_null_ptr), 'vec_workspace_create')
  call a%init(int(n), mvec, flavor=flavor)
  call b%init(int(n), mvec, flavor=flavor)
  call nka_hip_check(nka_hip_vec_alloc(ws, n, fa), 'vec_alloc')
  call nka_hip_check(nka_hip_vec_alloc(ws, n, buf), 'vec_alloc')      ! the caller's first buffer: B keeps it
  allocate(x(n), basis(n,2), ra(n), rb(n), coef(2))
  call random_number(basis)
  below = 0
  do t = 1, calls
    if (t == 8) then
      continue                              ! repeats the input of call 7: s == 0
    else if (t <= 10) then
      call random_number(coef)
      x = (coef(1) - 0.5_r8)*basis(:,1) + (coef(2) - 0.5_r8)*basis(:,2)
    else
      call random_number(x)
      x = 2.0_r8*x - 1.0_r8
    end if
    call nka_hip_check(nka_hip_vec_h2d(ws, n, fa, x), 'vec_h2d')
    call nka_hip_check(nka_hip_vec_h2d(ws, n, buf, x), 'vec_h2d')
    call a%accel_update_dev(fa)
    call b%accel_update_swap(buf, acc)      ! buf: now a free buffer for the next input; acc: the accelerated f
    call nka_hip_check(nka_hip_vec_d2h(ws, n, ra, fa), 'vec_d2h')
    call nka_hip_check(nka_hip_vec_d2h(ws, n, rb, acc), 'vec_d2h')
    if (any(ra /= rb)) then
      print '(a,i0)', 'MISMATCH at call ', t
      error stop 1
    end if
    nva = a%num_vec();  nvb = b%num_vec()   ! (synchronise)
    if (nva /= nvb) error stop 'num_vec differs'
    lb = b%list_bound()
    if (lb /= nvb + 1) then
      print '(a,i0,a,i0,a,i0)', 'call ', t, ': list_bound ', lb, ' but the list holds ', nvb + 1
      error stop 2
    end if
    if (lb < min(t, mvec + 1)) below = below + 1
  end do
  print '(a,i0,a,i0,a,i0)', 'OK calls=', calls, ' num_vec=', nvb, ' calls_with_a_bound_below_the_plain_count=', below

end program nka_swap_driver
