!! hip_block_vector_type -- a concrete, DEVICE-RESIDENT vector for the
!! abstract-vector flavour (the MI355X counterpart of the reference's example
!! grid_vector, src-F08-vector/grid_vector_type.F90:44-197).
!!
!! A block vector of NFIELD fields of NPER doubles each (BASELINE config 5: 4
!! fields x 1e7), every field a separate array in HBM.  Each deferred procedure
!! of class(vector) is one HIP kernel per field through the C ABI
!! (nka_hip_vec_*, nka_amd/csrc/vec_ops.hip); the elementwise results are rounded
!! like the Fortran expressions of grid_vector (a*x + b*y + z evaluated left to
!! right, no fused multiply-add); dot products are deterministic two-stage
!! reductions, summed over the fields in order.

module hip_block_vector_type

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use vector_class
  use nka_hip_c
  implicit none
  private

  type, extends(vector), public :: hip_block_vector
    integer :: nfield = 0
    integer(c_int64_t) :: nper = 0
    type(c_ptr), allocatable :: field(:)        ! device pointers, one per field
    type(c_ptr) :: ws = c_null_ptr              ! shared workspace (stream, reduction scratch); not owned
  contains
    procedure :: clone1
    procedure :: clone2
    procedure :: copy_
    procedure :: setval
    procedure :: scale
    procedure :: update1_
    procedure :: update2_
    procedure :: update3_
    procedure :: update4_
    procedure :: dot_
    procedure :: norm2 => norm2_
    !! specific to this type
    procedure :: init
    procedure :: release
    procedure :: set_field          ! host -> device
    procedure :: get_field          ! device -> host
  end type

  public :: hip_block_vector_workspace

contains

  !! One workspace (device ordinal, stream, scratch) shared by all vectors.
  function hip_block_vector_workspace(device) result(ws)
    integer, intent(in) :: device
    type(c_ptr) :: ws
    call nka_hip_check(nka_hip_vec_workspace_create(ws, int(device, c_int32_t), c_null_ptr), 'vec_workspace_create')
  end function

  subroutine init(this, nfield, nper, ws)
    class(hip_block_vector), intent(inout) :: this
    integer, intent(in) :: nfield
    integer(c_int64_t), intent(in) :: nper
    type(c_ptr), intent(in) :: ws
    integer :: k
    call this%release
    this%nfield = nfield
    this%nper = nper
    this%ws = ws
    allocate(this%field(nfield))
    do k = 1, nfield
      call nka_hip_check(nka_hip_vec_alloc(ws, nper, this%field(k)), 'vec_alloc')
    end do
  end subroutine

  !! Device memory is released explicitly (clones made by the accelerator live
  !! as long as the accelerator object).
  subroutine release(this)
    class(hip_block_vector), intent(inout) :: this
    integer :: k
    if (allocated(this%field)) then
      do k = 1, size(this%field)
        if (c_associated(this%field(k))) call nka_hip_check(nka_hip_vec_free(this%ws, this%field(k)), 'vec_free')
      end do
      deallocate(this%field)
    end if
    this%nfield = 0
  end subroutine

  subroutine set_field(this, k, array)
    class(hip_block_vector), intent(inout) :: this
    integer, intent(in) :: k
    real(r8), intent(in), contiguous :: array(:)
    if (size(array, kind=c_int64_t) /= this%nper) error stop 'hip_block_vector%set_field: wrong size'
    call nka_hip_check(nka_hip_vec_h2d(this%ws, this%nper, this%field(k), array), 'vec_h2d')
  end subroutine

  subroutine get_field(this, k, array)
    class(hip_block_vector), intent(in) :: this
    integer, intent(in) :: k
    real(r8), intent(out), contiguous :: array(:)
    if (size(array, kind=c_int64_t) /= this%nper) error stop 'hip_block_vector%get_field: wrong size'
    call nka_hip_check(nka_hip_vec_d2h(this%ws, this%nper, array, this%field(k)), 'vec_d2h')
  end subroutine

  !! clone: same structure, NEW device storage, values undefined (vector_class.F90:93-101)
  subroutine clone1(this, clone)
    class(hip_block_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone
    allocate(hip_block_vector :: clone)
    select type (clone)
    type is (hip_block_vector)
      call clone%init(this%nfield, this%nper, this%ws)
    end select
  end subroutine

  subroutine clone2(this, clone, n)
    class(hip_block_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone(:)
    integer, intent(in) :: n
    integer :: k
    allocate(hip_block_vector :: clone(n))
    select type (clone)
    type is (hip_block_vector)
      do k = 1, n
        call clone(k)%init(this%nfield, this%nper, this%ws)
      end do
    end select
  end subroutine

  subroutine copy_(dest, src)
    class(hip_block_vector), intent(inout) :: dest
    class(vector), intent(in) :: src
    integer :: k
    select type (src)
    class is (hip_block_vector)
      do k = 1, dest%nfield
        call nka_hip_check(nka_hip_vec_copy(dest%ws, dest%nper, dest%field(k), src%field(k)), 'vec_copy')
      end do
    end select
  end subroutine

  subroutine setval(this, val)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: val
    integer :: k
    do k = 1, this%nfield
      call nka_hip_check(nka_hip_vec_setval(this%ws, this%nper, this%field(k), val), 'vec_setval')
    end do
  end subroutine

  subroutine scale(this, a)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    integer :: k
    do k = 1, this%nfield
      call nka_hip_check(nka_hip_vec_scale(this%ws, this%nper, this%field(k), a), 'vec_scale')
    end do
  end subroutine

  subroutine update1_(this, a, x)               ! this <- a*x + this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    integer :: k
    select type (x)
    class is (hip_block_vector)
      do k = 1, this%nfield
        call nka_hip_check(nka_hip_vec_update1(this%ws, this%nper, this%field(k), a, x%field(k)), 'vec_update1')
      end do
    end select
  end subroutine

  subroutine update2_(this, a, x, b)            ! this <- a*x + b*this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x
    integer :: k
    select type (x)
    class is (hip_block_vector)
      do k = 1, this%nfield
        call nka_hip_check(nka_hip_vec_update2(this%ws, this%nper, this%field(k), a, x%field(k), b), 'vec_update2')
      end do
    end select
  end subroutine

  subroutine update3_(this, a, x, b, y)         ! this <- a*x + b*y + this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x, y
    integer :: k
    select type (x)
    class is (hip_block_vector)
      select type (y)
      class is (hip_block_vector)
        do k = 1, this%nfield
          call nka_hip_check(nka_hip_vec_update3(this%ws, this%nper, this%field(k), a, x%field(k), b, y%field(k)), &
                             'vec_update3')
        end do
      end select
    end select
  end subroutine

  subroutine update4_(this, a, x, b, y, c)      ! this <- a*x + b*y + c*this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b, c
    class(vector), intent(in) :: x, y
    integer :: k
    select type (x)
    class is (hip_block_vector)
      select type (y)
      class is (hip_block_vector)
        do k = 1, this%nfield
          call nka_hip_check(nka_hip_vec_update4(this%ws, this%nper, this%field(k), a, x%field(k), b, y%field(k), c), &
                             'vec_update4')
        end do
      end select
    end select
  end subroutine

  function dot_(x, y) result(val)
    class(hip_block_vector), intent(in) :: x
    class(vector), intent(in) :: y
    real(r8) :: val, part
    integer :: k
    val = 0.0_r8
    select type (y)
    class is (hip_block_vector)
      do k = 1, x%nfield
        call nka_hip_check(nka_hip_vec_dot(x%ws, x%nper, x%field(k), y%field(k), part), 'vec_dot')
        val = val + part
      end do
    end select
  end function

  function norm2_(this) result(val)
    class(hip_block_vector), intent(in) :: this
    real(r8) :: val, part
    integer :: k
    val = 0.0_r8
    do k = 1, this%nfield
      call nka_hip_check(nka_hip_vec_dot(this%ws, this%nper, this%field(k), this%field(k), part), 'vec_dot')
      val = val + part
    end do
    val = sqrt(val)
  end function

end module hip_block_vector_type
