!! nka_example_vec_dev -- BASELINE config 1 through the ABSTRACT-VECTOR flavour
!! with the whole solve resident on the GPU.
!!
!! The caller of the hot path in the reference's vector-flavour example
!! (src-F08-vector/nka_example.F90: system :66-179, solver :185-259, main :383-403)
!! works on grid_vector objects -- 2-D cell data with a ghost ring.  Here u and the
!! residual / correction r are hip_grid_vector objects in HBM
!! (hip_grid_vector_type.F90), the discrete system lives on the device
!! (include/nka_example_dev.h: nka_ex_residual_grid, nka_ex_pc_ssor_grid -- the
!! SSOR sweeps as anti-diagonal wavefronts, bit-identical to the lexicographic
!! loops), and the accelerator is the vector flavour of nka_type, which reaches the
!! device only through the hooks of class(vector):
!!
!!     r <- SSOR(residual(u)) ;  call accel%accel_update(r) ;  call u%update(-1, r)
!!
!! Per iteration only the 8-byte residual norm (r%norm2(), interior cells only)
!! and the scalars of the accelerator's hooks return to the host.  Options and
!! printed table are the reference's (:262-379, :233-255).

module example_system_dev

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use nka_hip_c
  use hip_grid_vector_type
  implicit none
  private

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
    integer(c_int) function nka_ex_residual_grid(sys, u, r) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: sys, u, r
    end function
    integer(c_int) function nka_ex_pc_ssor_grid(sys, nsweep, omega, r) bind(C)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), value :: sys, r
      integer(c_int32_t), value :: nsweep
      real(c_double), value :: omega
    end function
  end interface

  !! the discrete nonlinear system (reference `type system`, :73-84), coefficients in HBM
  type, public :: system_dev
    integer :: nx = 0, ny = 0
    type(c_ptr), private :: handle = c_null_ptr
  contains
    procedure :: init
    procedure :: residual
    procedure :: pc_ssor
    procedure :: release
  end type

contains

  subroutine init(this, a, nx, ny)
    class(system_dev), intent(inout) :: this
    real(r8), intent(in) :: a
    integer, intent(in) :: nx, ny
    call this%release
    call nka_hip_check(nka_ex_create(this%handle, int(nx, c_int32_t), int(ny, c_int32_t), a, 0_c_int32_t, c_null_ptr), &
                       'nka_ex_create')
    this%nx = nx
    this%ny = ny
  end subroutine

  subroutine release(this)
    class(system_dev), intent(inout) :: this
    if (c_associated(this%handle)) call nka_hip_check(nka_ex_destroy(this%handle), 'nka_ex_destroy')
    this%handle = c_null_ptr
  end subroutine

  !! r(1:nx,1:ny) <- residual of u (:103-120)
  subroutine residual(this, u, r)
    class(system_dev), intent(inout) :: this
    type(hip_grid_vector), intent(in) :: u
    type(hip_grid_vector), intent(inout) :: r
    call nka_hip_check(nka_ex_residual_grid(this%handle, u%base, r%base), 'nka_ex_residual_grid')
  end subroutine

  !! r <- nsweep SSOR sweeps applied to r, zero ghost ring (:147-179)
  subroutine pc_ssor(this, nsweep, omega, r)
    class(system_dev), intent(inout) :: this
    integer, intent(in) :: nsweep
    real(r8), intent(in) :: omega
    type(hip_grid_vector), intent(inout) :: r
    call nka_hip_check(nka_ex_pc_ssor_grid(this%handle, int(nsweep, c_int32_t), omega, r%base), 'nka_ex_pc_ssor_grid')
  end subroutine

end module example_system_dev


program nka_example_vec_dev

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use hip_block_vector_type, only: hip_block_vector_workspace
  use hip_grid_vector_type
  use example_system_dev
  use nka_type
  implicit none

  integer :: nx = 50, nsweep = 2, mvec = 0, maxitr = 999
  logical :: compact = .false.
  real(r8) :: a = 0.02_r8, omega = 1.4_r8

  call read_options
  call run

contains

  subroutine run
    type(system_dev) :: sys
    type(nka) :: accel
    type(hip_grid_vector) :: u, r
    type(c_ptr) :: ws
    real(r8) :: rnorm, rnorm0, red, rate
    integer :: itr
    real(r8), parameter :: TOL = 1.0e-6_r8

    ws = hip_block_vector_workspace(0)
    call sys%init(a, nx, nx)
    call r%init_grid(nx, nx, ws)
    call r%setval(0.0_r8)                 ! the reference leaves the ghosts of its work vector undefined
    if (mvec > 0) call accel%init(r, mvec, compact=compact)
    call u%init_grid(nx, nx, ws)
    call u%setval(0.0_r8)                 ! initial guess plus boundary data (:397)

    write(*,'(a4,a14,a13,a8)') 'Iter', 'Residual Norm', 'Reduction', 'Rate'
    call sys%residual(u, r)
    rnorm0 = r%norm2()
    write(*,'(i3,a,es14.6)') 0, ':', rnorm0
    do itr = 1, maxitr
      call sys%pc_ssor(nsweep, omega, r)
      if (mvec > 0) call accel%accel_update(r)       ! <-- the hot path, through the hooks of class(vector)
      call u%update(-1.0_r8, r)
      call sys%residual(u, r)
      rnorm = r%norm2()
      red = rnorm / rnorm0
      rate = red**(1.0_r8/itr)
      write(*,'(i3,a,es14.6,es13.3,f8.3)') itr, ':', rnorm, red, rate
      if (rnorm < TOL*rnorm0) exit
    end do
    call u%release
    call r%release
    call sys%release
  end subroutine

  subroutine read_options
    integer :: k, ios, ic
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
      case ('--maxitr');  read(val,*,iostat=ios) maxitr
      case ('--compact')
        read(val,*,iostat=ios) ic
        compact = ic /= 0
      case default
        write(*,'(a)') 'usage: nka_example_vec_dev [-n N] [-a A] [--sweeps S] [--omega W] [--nka-vec M] [--compact 0|1] [--maxitr K]'
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

end program nka_example_vec_dev
