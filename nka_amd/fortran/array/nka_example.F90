!! nka_example -- BASELINE config 1 driven through the Fortran front end.
!!
!! The caller of the hot path in the reference's example program
!! (src-F08/nka_example.F90): cell-centred finite volumes for
!!     -div((a+u) grad u) = 1   on the unit square,  u = 0 on the boundary,
!! solved by the fixed-point iteration  r <- SSOR(residual(u)); accelerate r;
!! u <- u - r.  Discretisation, sweep order and the stopping rule follow
!! src-F08/nka_example.F90:86-179, 226-256 so that the printed table can be
!! compared with the reference's reference_output; the code itself is written
!! for this repository (one module, explicit ghost-layer arrays).  The PDE and
!! SSOR parts run on the host (SSOR is a sequential sweep -- not the accelerated
!! path); only accel%accel_update(r) runs on the GPU.
!!
!! Options (as the reference, :280-353): -n N, -a A, --sweeps S, --omega W,
!! --nka-vec M.  Output format (:239-253) is unchanged.

module elliptic_problem

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  implicit none
  private
  public :: problem, problem_init, problem_residual, problem_ssor

  type :: problem
    integer :: nx = 0, ny = 0
    real(r8) :: a = 0.02_r8, hx = 0, hy = 0
    !! face coefficients west/east (cx), south/north (cy), centre (cc), source
    real(r8), allocatable :: cx(:,:), cy(:,:), cc(:,:), src(:,:)
  end type

contains

  subroutine problem_init(p, a, nx, ny)
    type(problem), intent(out) :: p
    real(r8), intent(in) :: a
    integer, intent(in) :: nx, ny
    p%a = a;  p%nx = nx;  p%ny = ny
    p%hx = 1.0_r8 / nx
    p%hy = 1.0_r8 / ny
    allocate(p%cx(nx+1,ny), p%cy(nx,ny+1), p%cc(nx,ny), p%src(nx,ny))
    p%src = 1.0_r8
  end subroutine

  !! Harmonic means of the cell conductivities a+u on the faces
  !! (src-F08/nka_example.F90:122-145), accumulated cell by cell so that the
  !! roundings are the reference's.
  subroutine assemble(p, u)
    type(problem), intent(inout) :: p
    real(r8), intent(in) :: u(0:,0:)
    integer :: i, j
    real(r8) :: rinv
    p%cx = 0.0_r8
    p%cy = 0.0_r8
    do j = 1, p%ny
      do i = 1, p%nx
        rinv = 1.0_r8 / (p%a + u(i,j))
        p%cx(i,j)   = p%cx(i,j)   + (rinv*p%hx**2)
        p%cx(i+1,j) = p%cx(i+1,j) + (rinv*p%hx**2)
        p%cy(i,j)   = p%cy(i,j)   + (rinv*p%hy**2)
        p%cy(i,j+1) = p%cy(i,j+1) + (rinv*p%hy**2)
      end do
    end do
    p%cx = 2.0_r8 / p%cx
    p%cy = 2.0_r8 / p%cy
    do j = 1, p%ny
      do i = 1, p%nx
        p%cc(i,j) = p%cx(i,j) + p%cx(i+1,j) + p%cy(i,j) + p%cy(i,j+1)
      end do
    end do
  end subroutine

  subroutine problem_residual(p, u, r)          ! :103-120
    type(problem), intent(inout) :: p
    real(r8), intent(in)  :: u(0:,0:)
    real(r8), intent(out) :: r(:,:)
    integer :: i, j
    call assemble(p, u)
    do j = 1, p%ny
      do i = 1, p%nx
        r(i,j) = p%cc(i,j)*u(i,j) - p%cx(i,j)*u(i-1,j) - p%cx(i+1,j)*u(i+1,j) &
                                  - p%cy(i,j)*u(i,j-1) - p%cy(i,j+1)*u(i,j+1) - p%src(i,j)
      end do
    end do
  end subroutine

  subroutine problem_ssor(p, nsweep, omega, r)  ! :147-179
    type(problem), intent(in) :: p
    integer, intent(in) :: nsweep
    real(r8), intent(in) :: omega
    real(r8), intent(inout) :: r(:,:)
    real(r8) :: z(0:p%nx+1,0:p%ny+1)
    integer :: s, i, j, pass, i0, i1, j0, j1, st
    z = 0.0_r8
    do s = 1, nsweep
      do pass = 1, 2                 ! forward, then backward
        if (pass == 1) then
          i0 = 1; i1 = p%nx; j0 = 1; j1 = p%ny; st = 1
        else
          i0 = p%nx; i1 = 1; j0 = p%ny; j1 = 1; st = -1
        end if
        do j = j0, j1, st
          do i = i0, i1, st
            z(i,j) = (1 - omega)*z(i,j) + omega*(r(i,j) &
                       + p%cx(i,j)*z(i-1,j) + p%cx(i+1,j)*z(i+1,j)  &
                       + p%cy(i,j)*z(i,j-1) + p%cy(i,j+1)*z(i,j+1))/p%cc(i,j)
          end do
        end do
      end do
    end do
    r = z(1:p%nx,1:p%ny)
  end subroutine

end module elliptic_problem


program nka_example

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use elliptic_problem
  use nka_type
  implicit none

  integer :: nx = 50, nsweep = 2, mvec = 0, flavor = NKA_HIP_FLAVOR_DEFAULT
  real(r8) :: a = 0.02_r8, omega = 1.4_r8

  call read_options
  call run

contains

  subroutine run
    type(problem) :: prob
    type(nka) :: accel
    real(r8), allocatable, target :: r1(:)
    real(r8), allocatable :: u(:,:)
    real(r8), pointer :: r(:,:)
    real(r8) :: rnorm, rnorm0, red, rate
    integer :: itr
    integer, parameter :: MAXITR = 999
    real(r8), parameter :: TOL = 1.0e-6_r8

    call problem_init(prob, a, nx, nx)
    if (mvec > 0) call accel%init(nx*nx, mvec, flavor=flavor)
    allocate(u(0:nx+1,0:nx+1), r1(nx*nx))
    u = 0.0_r8
    r(1:nx,1:nx) => r1

    write(*,'(a4,a14,a13,a8)') 'Iter', 'Residual Norm', 'Reduction', 'Rate'
    call problem_residual(prob, u, r)
    rnorm0 = norm2(r1)
    write(*,'(i3,a,es14.6)') 0, ':', rnorm0
    do itr = 1, MAXITR
      call problem_ssor(prob, nsweep, omega, r)
      if (mvec > 0) call accel%accel_update(r1)       ! <-- the hot path, on the GPU
      u(1:nx,1:nx) = u(1:nx,1:nx) - r
      call problem_residual(prob, u, r)
      rnorm = norm2(r1)
      red = rnorm / rnorm0
      rate = red**(1.0_r8/itr)
      write(*,'(i3,a,es14.6,es13.3,f8.3)') itr, ':', rnorm, red, rate
      if (rnorm < TOL*rnorm0) exit
    end do
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
      case default
        write(*,'(a)') 'usage: nka_example [-n N] [-a A] [--sweeps S] [--omega W] [--nka-vec M] [--flavor 0|1|2]'
        stop 1
      end select
      if (ios /= 0 .or. len_trim(val) == 0) then
        write(*,'(2a)') 'bad or missing value for ', trim(arg)
        stop 1
      end if
      k = k + 2
    end do
    if (nx < 3 .or. a <= 0.0_r8 .or. nsweep < 1 .or. omega <= 0.0_r8 .or. mvec < 0) then
      write(*,'(a)') 'invalid option value'
      stop 1
    end if
  end subroutine

end program nka_example
