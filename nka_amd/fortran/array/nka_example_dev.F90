!! nka_example_dev -- BASELINE config 1 with the WHOLE solve resident on the GPU
!! (SURVEY.md 8 row f4).
!!
!! The same fixed-point iteration as nka_example (reference: src-F08/
!! nka_example.F90:226-256),
!!     r <- SSOR(residual(u)) ;  accelerate r ;  u <- u - r ,
!! but u, r and every coefficient array live in HBM: the residual / coefficient
!! stencils, the SSOR sweeps (as anti-diagonal wavefronts: same bits as the
!! reference's lexicographic loops), the solution update and the residual norm are
!! device kernels (include/nka_example_dev.h, nka_hip_vec_norm2), and the
!! accelerator is called on device memory (accel%accel_update_dev).  Per iteration
!! only the 8-byte norm returns to the host.  The printed table is the
!! reference's (same options, same format, :239-253).

module nka_example_dev_c

  use, intrinsic :: iso_c_binding
  implicit none
  interface
    integer(c_int) function nka_ex_create(sys, nx, ny, a, device, stream) bind(C)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), intent(out) :: sys
      integer(c_int32_t), value :: nx, ny, device
      real(c_double), value :: a
      type(c_ptr), value :: stream
    end function
    integer(c_int) function nka_ex_destroy(sys) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: sys
    end function
    integer(c_int) function nka_ex_residual(sys, uext, r) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: sys, uext, r
    end function
    integer(c_int) function nka_ex_pc_ssor(sys, nsweep, omega, r) bind(C)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), value :: sys, r
      integer(c_int32_t), value :: nsweep
      real(c_double), value :: omega
    end function
    integer(c_int) function nka_ex_update_solution(sys, uext, r) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: sys, uext, r
    end function
  end interface

end module nka_example_dev_c


program nka_example_dev

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use nka_hip_c
  use nka_example_dev_c
  use nka_type
  implicit none

  integer :: nx = 50, nsweep = 2, mvec = 0, flavor = NKA_HIP_FLAVOR_DEFAULT, maxitr = 999
  real(r8) :: a = 0.02_r8, omega = 1.4_r8

  call read_options
  call run

contains

  subroutine run
    type(nka) :: accel
    type(c_ptr) :: sys, ws, u_dev, r_dev
    real(r8) :: rnorm, rnorm0, red, rate
    integer :: itr
    integer(c_int64_t) :: n, next
    real(r8), parameter :: TOL = 1.0e-6_r8

    n = int(nx, c_int64_t) * nx
    next = int(nx+2, c_int64_t) * (nx+2)
    call nka_hip_check(nka_hip_vec_workspace_create(ws, 0_c_int32_t, c_null_ptr), 'vec_workspace_create')
    call nka_hip_check(nka_ex_create(sys, int(nx, c_int32_t), int(nx, c_int32_t), a, 0_c_int32_t, c_null_ptr), &
                       'nka_ex_create')
    call nka_hip_check(nka_hip_vec_alloc(ws, next, u_dev), 'vec_alloc(u)')
    call nka_hip_check(nka_hip_vec_alloc(ws, n, r_dev), 'vec_alloc(r)')
    call nka_hip_check(nka_hip_vec_setval(ws, next, u_dev, 0.0_r8), 'vec_setval(u)')   ! u = 0, boundary ring included
    if (mvec > 0) call accel%init(int(n), mvec, flavor=flavor)

    write(*,'(a4,a14,a13,a8)') 'Iter', 'Residual Norm', 'Reduction', 'Rate'
    call nka_hip_check(nka_ex_residual(sys, u_dev, r_dev), 'residual')
    call nka_hip_check(nka_hip_vec_norm2(ws, n, r_dev, rnorm0), 'norm2')
    write(*,'(i3,a,es14.6)') 0, ':', rnorm0
    do itr = 1, maxitr
      call nka_hip_check(nka_ex_pc_ssor(sys, int(nsweep, c_int32_t), omega, r_dev), 'pc_ssor')
      if (mvec > 0) call accel%accel_update_dev(r_dev)      ! <-- the hot path, device memory in, device memory out
      call nka_hip_check(nka_ex_update_solution(sys, u_dev, r_dev), 'update_solution')
      call nka_hip_check(nka_ex_residual(sys, u_dev, r_dev), 'residual')
      call nka_hip_check(nka_hip_vec_norm2(ws, n, r_dev, rnorm), 'norm2')
      red = rnorm / rnorm0
      rate = red**(1.0_r8/itr)
      write(*,'(i3,a,es14.6,es13.3,f8.3)') itr, ':', rnorm, red, rate
      if (rnorm < TOL*rnorm0) exit
    end do
    call nka_hip_check(nka_hip_vec_free(ws, u_dev), 'vec_free')
    call nka_hip_check(nka_hip_vec_free(ws, r_dev), 'vec_free')
    call nka_hip_check(nka_ex_destroy(sys), 'nka_ex_destroy')
  end subroutine

  subroutine read_options
    integer :: k, ios
    character(64) :: arg, val
    k = 1
    do while (k <= command_argument_count())
      call get_command_argument(k, arg)
      val = ''
      if (k < command_argument_count()) call get_command_argument(k+1, val)
      ios = 0
      select case (arg)
      case ('-n');        read(val,*,iostat=ios) nx
      case ('-a');        read(val,*,iostat=ios) a
      case ('--sweeps');  read(val,*,iostat=ios) nsweep
      case ('--omega');   read(val,*,iostat=ios) omega
      case ('--nka-vec'); read(val,*,iostat=ios) mvec
      case ('--flavor');  read(val,*,iostat=ios) flavor
      case ('--maxitr');  read(val,*,iostat=ios) maxitr
      case default
        write(*,'(a)') 'usage: nka_example_dev [-n N] [-a A] [--sweeps S] [--omega W] [--nka-vec M] [--flavor 0|1|2] [--maxitr K]'
        stop 1
      end select
      if (ios /= 0 .or. len_trim(val) == 0) then
        write(*,'(2a)') 'bad or missing value for ', trim(arg)
        stop 1
      end if
      k = k + 2
    end do
    if (nx < 3 .or. a <= 0.0_r8 .or. nsweep < 1 .or. omega <= 0.0_r8 .or. mvec < 0 .or. maxitr < 1) then
      write(*,'(a)') 'invalid option value'
      stop 1
    end if
  end subroutine

end program nka_example_dev
