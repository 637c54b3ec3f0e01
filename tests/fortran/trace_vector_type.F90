!! trace_vector_type -- TEST INFRASTRUCTURE (no GPU): a CPU vector type that supplies ONLY the eleven
!! deferred procedures of the reference's abstract vector class (src-F08-vector/vector_class.F90:93-108)
!! and writes one line per hook call to a trace file: the hook, the serial numbers of the vectors
!! it was handed (numbered in order of creation) and the bit patterns of every scalar that goes in
!! or comes out.  The same source compiles against the reference's vector_class / nka_type and
!! against this repository's; the two traces of one driver run are then compared line by line
!! (tests/test_hook_trace.py): a user type that overrides nothing must see the reference's calls,
!! in the reference's order, with the reference's coefficients.

module trace_vector_type

  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use vector_class
  implicit none
  private

  integer, save :: serial = 0          ! vectors created so far
  integer, save, public :: trace_unit = -1

  type, extends(vector), public :: trace_vector
    real(r8), allocatable :: x(:)
    integer :: id = 0
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

contains

  subroutine init(this, n)
    class(trace_vector), intent(out) :: this
    integer, intent(in) :: n
    allocate(this%x(n))
    this%x = 0.0_r8
    serial = serial + 1
    this%id = serial
  end subroutine

  integer function id_of(v)
    class(vector), intent(in) :: v
    id_of = -1
    select type (v)
    class is (trace_vector)
      id_of = v%id
    end select
  end function

  function bits(a) result(h)
    real(r8), intent(in) :: a
    character(16) :: h
    write(h,'(z16.16)') transfer(a, 1_i8)
  end function

  subroutine clone1(this, clone)
    class(trace_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone
    allocate(trace_vector :: clone)
    select type (clone)
    type is (trace_vector)
      call clone%init(size(this%x))
      write(trace_unit,'(a,2(1x,i0))') 'clone1', this%id, clone%id
    end select
  end subroutine

  subroutine clone2(this, clone, n)
    class(trace_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone(:)
    integer, intent(in) :: n
    integer :: k
    allocate(trace_vector :: clone(n))
    select type (clone)
    type is (trace_vector)
      do k = 1, n
        call clone(k)%init(size(this%x))
      end do
      write(trace_unit,'(a,4(1x,i0))') 'clone2', this%id, n, clone(1)%id, clone(n)%id
    end select
  end subroutine

  subroutine copy_(dest, src)
    class(trace_vector), intent(inout) :: dest
    class(vector), intent(in) :: src
    write(trace_unit,'(a,2(1x,i0))') 'copy', dest%id, id_of(src)
    select type (src)
    class is (trace_vector)
      dest%x = src%x
    end select
  end subroutine

  subroutine setval(this, val)
    class(trace_vector), intent(inout) :: this
    real(r8), intent(in) :: val
    write(trace_unit,'(a,1x,i0,1x,a)') 'setval', this%id, bits(val)
    this%x = val
  end subroutine

  subroutine scale(this, a)
    class(trace_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    write(trace_unit,'(a,1x,i0,1x,a)') 'scale', this%id, bits(a)
    this%x = a * this%x
  end subroutine

  subroutine update1_(this, a, x)
    class(trace_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    write(trace_unit,'(a,2(1x,i0),1x,a)') 'update1', this%id, id_of(x), bits(a)
    select type (x)
    class is (trace_vector)
      this%x = a * x%x + this%x
    end select
  end subroutine

  subroutine update2_(this, a, x, b)
    class(trace_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x
    write(trace_unit,'(a,2(1x,i0),2(1x,a))') 'update2', this%id, id_of(x), bits(a), bits(b)
    select type (x)
    class is (trace_vector)
      this%x = a * x%x + b * this%x
    end select
  end subroutine

  subroutine update3_(this, a, x, b, y)
    class(trace_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x, y
    write(trace_unit,'(a,3(1x,i0),2(1x,a))') 'update3', this%id, id_of(x), id_of(y), bits(a), bits(b)
    select type (x)
    class is (trace_vector)
      select type (y)
      class is (trace_vector)
        this%x = a * x%x + b * y%x + this%x
      end select
    end select
  end subroutine

  subroutine update4_(this, a, x, b, y, c)
    class(trace_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b, c
    class(vector), intent(in) :: x, y
    write(trace_unit,'(a,3(1x,i0),3(1x,a))') 'update4', this%id, id_of(x), id_of(y), bits(a), bits(b), bits(c)
    select type (x)
    class is (trace_vector)
      select type (y)
      class is (trace_vector)
        this%x = a * x%x + b * y%x + c * this%x
      end select
    end select
  end subroutine

  function dot_(x, y) result(val)
    class(trace_vector), intent(in) :: x
    class(vector), intent(in) :: y
    real(r8) :: val
    integer :: i
    val = 0.0_r8
    select type (y)
    class is (trace_vector)
      do i = 1, size(x%x)
        val = val + x%x(i) * y%x(i)
      end do
    end select
    write(trace_unit,'(a,2(1x,i0),1x,a)') 'dot', x%id, id_of(y), bits(val)
  end function

  function norm2_(this) result(val)
    class(trace_vector), intent(in) :: this
    real(r8) :: val
    integer :: i
    val = 0.0_r8
    do i = 1, size(this%x)
      val = val + this%x(i) * this%x(i)
    end do
    val = sqrt(val)
    write(trace_unit,'(a,1x,i0,1x,a)') 'norm2', this%id, bits(val)
  end function

end module trace_vector_type
