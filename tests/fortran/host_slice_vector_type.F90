!! host_slice_vector_type -- TEST INFRASTRUCTURE: a CPU vector type for the sharded tests of
!! the abstract-vector flavour that need no GPU.  It is what a user of the reference writes
!! for a distributed run (src-F08-vector/README.md:16-22): each rank holds a contiguous slice
!! and the two reduction methods, dot_ and norm2, are parallel-aware -- here through the
!! test-only host all-reduce tests/c/shm_allreduce.c where a real code calls MPI_Allreduce.
!! Only the ELEVEN deferred procedures are supplied: the accelerator reaches them through the
!! default bodies of the batched and stage hooks of vector_class, i.e. in the reference's order.

module host_slice_vector_type

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use vector_class
  implicit none
  private

  type, extends(vector), public :: host_slice_vector
    real(r8), pointer :: x(:) => null()       ! this rank's slice (pointer: views handed to slice_of)
    type(c_ptr) :: comm = c_null_ptr           ! shm_ar context shared by all vectors of the rank
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
    procedure :: init
  end type

  interface
    function shm_allreduce(ctx, vals, count) bind(C) result(rc)
      import :: c_ptr, c_double, c_int32_t, c_int
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: vals(*)
      integer(c_int32_t), value :: count
      integer(c_int) :: rc
    end function
  end interface

contains

  subroutine init(this, nloc, comm)
    class(host_slice_vector), intent(out) :: this
    integer, intent(in) :: nloc
    type(c_ptr), intent(in) :: comm
    allocate(this%x(nloc))
    this%x = 0.0_r8
    this%comm = comm
  end subroutine

  real(r8) function global_sum(this, s)
    class(host_slice_vector), intent(in) :: this
    real(r8), intent(in) :: s
    real(r8) :: v(1)
    v(1) = s
    if (shm_allreduce(this%comm, v, 1_c_int32_t) /= 0) error stop 'host_slice_vector: all-reduce failed'
    global_sum = v(1)
  end function

  subroutine clone1(this, clone)
    class(host_slice_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone
    allocate(host_slice_vector :: clone)
    select type (clone)
    type is (host_slice_vector)
      call clone%init(size(this%x), this%comm)
    end select
  end subroutine

  subroutine clone2(this, clone, n)
    class(host_slice_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone(:)
    integer, intent(in) :: n
    integer :: k
    allocate(host_slice_vector :: clone(n))
    select type (clone)
    type is (host_slice_vector)
      do k = 1, n
        call clone(k)%init(size(this%x), this%comm)
      end do
    end select
  end subroutine

  !! the slice behind a class(vector) argument of this type (the NVI wrappers of vector_class have
  !! already checked same_type_as)
  function slice_of(v) result(p)
    class(vector), intent(in), target :: v
    real(r8), pointer :: p(:)
    p => null()
    select type (v)
    class is (host_slice_vector)
      p => v%x
    end select
    if (.not. associated(p)) error stop 'host_slice_vector: foreign vector type'
  end function

  subroutine copy_(dest, src)
    class(host_slice_vector), intent(inout) :: dest
    class(vector), intent(in) :: src
    real(r8), pointer :: s(:)
    s => slice_of(src)
    dest%x = s
  end subroutine

  subroutine setval(this, val)
    class(host_slice_vector), intent(inout) :: this
    real(r8), intent(in) :: val
    this%x = val
  end subroutine

  subroutine scale(this, a)
    class(host_slice_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    integer :: i
    do i = 1, size(this%x)
      this%x(i) = a * this%x(i)
    end do
  end subroutine

  !! the four update forms of vector_class, element by element, each with the association of the
  !! reference's expression (a*x + this ; a*x + b*this ; a*x + b*y + this ; a*x + b*y + c*this)
  subroutine update1_(this, a, x)
    class(host_slice_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    real(r8), pointer :: xs(:)
    integer :: i
    xs => slice_of(x)
    do i = 1, size(this%x)
      this%x(i) = a * xs(i) + this%x(i)
    end do
  end subroutine

  subroutine update2_(this, a, x, b)
    class(host_slice_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x
    real(r8), pointer :: xs(:)
    integer :: i
    xs => slice_of(x)
    do i = 1, size(this%x)
      this%x(i) = a * xs(i) + b * this%x(i)
    end do
  end subroutine

  subroutine update3_(this, a, x, b, y)
    class(host_slice_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x, y
    real(r8), pointer :: xs(:), ys(:)
    integer :: i
    xs => slice_of(x)
    ys => slice_of(y)
    do i = 1, size(this%x)
      this%x(i) = a * xs(i) + b * ys(i) + this%x(i)
    end do
  end subroutine

  subroutine update4_(this, a, x, b, y, c)
    class(host_slice_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b, c
    class(vector), intent(in) :: x, y
    real(r8), pointer :: xs(:), ys(:)
    integer :: i
    xs => slice_of(x)
    ys => slice_of(y)
    do i = 1, size(this%x)
      this%x(i) = a * xs(i) + b * ys(i) + c * this%x(i)
    end do
  end subroutine

  !! parallel-aware: the local partial sum, then the sum over the ranks
  function dot_(x, y) result(val)
    class(host_slice_vector), intent(in) :: x
    class(vector), intent(in) :: y
    real(r8) :: val
    real(r8), pointer :: ys(:)
    ys => slice_of(y)
    val = global_sum(x, dot_product(x%x, ys))
  end function

  function norm2_(this) result(val)
    class(host_slice_vector), intent(in) :: this
    real(r8) :: val
    val = sqrt(global_sum(this, dot_product(this%x, this%x)))
  end function

end module host_slice_vector_type
