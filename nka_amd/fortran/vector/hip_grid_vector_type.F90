!! hip_grid_vector_type -- the DEVICE-RESIDENT counterpart of the reference's
!! example vector, grid_vector (src-F08-vector/grid_vector_type.F90:44-197):
!! cell-centred data on an NX x NY grid with a ring of ghost cells, i = 0..NX+1,
!! j = 0..NY+1.  Same semantics --
!!   * clone, copy, setval, scale and the four updates act on ALL (NX+2)(NY+2)
!!     values, ghosts included (:104-165);
!!   * dot and norm2 sum the interior i = 1..NX, j = 1..NY only (:170-197) --
!! but not the same layout.  In HBM the NX*NY interior values come first,
!! x(i,j) at offset (i-1) + (j-1)*NX, and the ring is packed behind them: row
!! j = 0 (i = 0..NX+1), row j = NY+1, column i = 0 (j = 1..NY), column i = NX+1.
!! A reduction is then a dense, 16-byte aligned prefix of the allocation, so the
!! type is simply a one-field hip_block_vector with an unreduced tail of
!! 2(NX+2) + 2NY values: every hook, including the fused stage hooks the
!! accelerator calls, is inherited, and the three passes of an update run over the
!! interior at full bandwidth with no per-element ghost test.  The stencil kernels
!! that need neighbours (include/nka_example_dev.h: nka_ex_residual_grid,
!! nka_ex_pc_ssor_grid) read the same layout.
!!
!! set_array / get_array move a host array shaped like the reference's
!! `array(0:nx+1,0:ny+1)` to and from this layout.

module hip_grid_vector_type

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use vector_class
  use nka_hip_c
  use hip_block_vector_type
  implicit none
  private

  type, extends(hip_block_vector), public :: hip_grid_vector
    integer :: nx = 0, ny = 0
  contains
    procedure :: clone1 => grid_clone1
    procedure :: clone2 => grid_clone2
    procedure :: init_grid
    procedure :: set_array
    procedure :: get_array
  end type

contains

  !! grid_vector%init(nx, ny) (grid_vector_type.F90:74-80); values undefined
  subroutine init_grid(this, nx, ny, ws)
    class(hip_grid_vector), intent(inout) :: this
    integer, intent(in) :: nx, ny
    type(c_ptr), intent(in) :: ws
    if (nx < 1 .or. ny < 1) error stop 'hip_grid_vector%init_grid: nx, ny must be positive'
    call this%init(1, int(nx, c_int64_t) * ny, ws, ntail=2_c_int64_t*(nx+2) + 2_c_int64_t*ny)
    this%nx = nx
    this%ny = ny
  end subroutine

  !! clones keep the dynamic type: the NVI wrappers of class(vector) insist on
  !! same_type_as (vector_class.F90, reference :157,167,180)
  subroutine grid_clone1(this, clone)
    class(hip_grid_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone
    allocate(hip_grid_vector :: clone)
    select type (clone)
    type is (hip_grid_vector)
      call clone%init_grid(this%nx, this%ny, this%ws)
    end select
  end subroutine

  subroutine grid_clone2(this, clone, n)
    class(hip_grid_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone(:)
    integer, intent(in) :: n
    integer :: k
    allocate(hip_grid_vector :: clone(n))
    select type (clone)
    type is (hip_grid_vector)
      do k = 1, n
        call clone(k)%init_grid(this%nx, this%ny, this%ws)
      end do
    end select
  end subroutine

  !! host array(0:nx+1,0:ny+1) -> device
  subroutine set_array(this, array)
    class(hip_grid_vector), intent(inout) :: this
    real(r8), intent(in) :: array(0:,0:)
    real(r8), allocatable :: buf(:)
    integer(c_int64_t) :: p
    integer :: i, j
    if (size(array,1) /= this%nx+2 .or. size(array,2) /= this%ny+2) error stop 'hip_grid_vector%set_array: wrong shape'
    allocate(buf(this%ntot))
    p = 0
    do j = 1, this%ny
      do i = 1, this%nx
        p = p + 1
        buf(p) = array(i,j)
      end do
    end do
    do i = 0, this%nx+1
      buf(p+1+i) = array(i,0)
      buf(p+1+(this%nx+2)+i) = array(i,this%ny+1)
    end do
    p = p + 2*(this%nx+2)
    do j = 1, this%ny
      buf(p+j) = array(0,j)
      buf(p+this%ny+j) = array(this%nx+1,j)
    end do
    call nka_hip_check(nka_hip_vec_h2d(this%ws, this%ntot, this%base, buf), 'vec_h2d')
  end subroutine

  !! device -> host array(0:nx+1,0:ny+1)
  subroutine get_array(this, array)
    class(hip_grid_vector), intent(in) :: this
    real(r8), intent(out) :: array(0:,0:)
    real(r8), allocatable :: buf(:)
    integer(c_int64_t) :: p
    integer :: i, j
    if (size(array,1) /= this%nx+2 .or. size(array,2) /= this%ny+2) error stop 'hip_grid_vector%get_array: wrong shape'
    allocate(buf(this%ntot))
    call nka_hip_check(nka_hip_vec_d2h(this%ws, this%ntot, buf, this%base), 'vec_d2h')
    p = 0
    do j = 1, this%ny
      do i = 1, this%nx
        p = p + 1
        array(i,j) = buf(p)
      end do
    end do
    do i = 0, this%nx+1
      array(i,0) = buf(p+1+i)
      array(i,this%ny+1) = buf(p+1+(this%nx+2)+i)
    end do
    p = p + 2*(this%nx+2)
    do j = 1, this%ny
      array(0,j) = buf(p+j)
      array(this%nx+1,j) = buf(p+this%ny+j)
    end do
  end subroutine

end module hip_grid_vector_type
