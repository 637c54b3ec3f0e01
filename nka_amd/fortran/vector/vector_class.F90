!! vector_class -- the abstract vector interface of the abstract-vector flavour.
!!
!! Plugin interface kept intact from the reference (src-F08-vector/
!! vector_class.F90:90-228): a user type extends `vector` and supplies the eleven
!! deferred procedures
!!   clone1, clone2, setval, scale, norm2, copy_, update1_ .. update4_, dot_
!! and the accelerator only ever calls the public generics
!!   clone, copy, setval, scale, update (1-4 coefficient forms), dot, norm2.
!! The non-deferred wrappers check that the operands have the same dynamic type
!! (error stop otherwise, :157,167,180,194,210,226) and short-circuit zero
!! coefficients exactly like the reference (:176,189,203-206,219-222).
!!
!! Additions (SURVEY.md 8 f1), NOT deferred, so existing user types keep
!! compiling: dot_many and update_many have default bodies that loop over the
!! deferred hooks; a device vector may override them with fused kernels.
!! dot_pair_many computes BOTH inner-product rows an update needs (the new
!! difference against the stored w's, and f against them) in one pass.
!!
!! STAGE hooks, also NOT deferred: update_norm2, scale_dot_pair_many,
!! update_many_keep, axpy_many_keep name the three groups of statements of
!! accel_update between which a value reduced over the whole vector is needed
!! (F08V:237-238 | 255-264 + 347 | 336 + 374 + 382).  Their default bodies ARE
!! those statements, hook call by hook call, so the accelerator written against
!! them performs exactly the reference's sequence on any user type; a device
!! vector overrides each with ONE fused kernel (8n(12+3m) bytes per update).

module vector_class

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  implicit none
  private
  !! The DEFAULT bodies of the reduction-bearing batched / stage hooks under names of their own: a type that overrides
  !! them can still run the reference's sequence of deferred hook calls on request (hip_block_vector does, when its
  !! workspace sums in the reference's order) -- Fortran has no way to invoke an overridden binding of an ABSTRACT parent.
  public :: vector_default_dot_many, vector_default_dot_pair_many, vector_default_update_norm2, &
            vector_default_scale_dot_pair_many

  type, abstract, public :: vector
  contains
    generic :: clone => clone1, clone2
    procedure(clone1_if), deferred :: clone1
    procedure(clone2_if), deferred :: clone2
    procedure :: copy
    procedure(setval_if), deferred :: setval
    procedure(scale_if), deferred :: scale
    generic :: update => update1, update2, update3, update4
    procedure, private :: update1, update2, update3, update4
    procedure :: dot
    procedure(norm2_if), deferred :: norm2
    !! the hooks behind the wrappers (called only with same-type operands)
    procedure(copy_if), deferred :: copy_
    procedure(update1_if), deferred :: update1_
    procedure(update2_if), deferred :: update2_
    procedure(update3_if), deferred :: update3_
    procedure(update4_if), deferred :: update4_
    procedure(dot_if), deferred :: dot_
    !! optional batched forms (overridable, default = loops over the hooks)
    procedure :: dot_many
    procedure :: dot_pair_many
    procedure :: update_many
    procedure :: axpy_many
    !! optional stage hooks (overridable, default = the reference's hook sequence)
    procedure :: update_norm2
    procedure :: update_norm2_dots
    procedure :: scale_dot_pair_many
    procedure :: update_many_keep
    procedure :: axpy_many_keep
  end type

  abstract interface
    subroutine clone1_if(this, clone)
      import :: vector
      class(vector), intent(in) :: this
      class(vector), allocatable, intent(out) :: clone
    end subroutine
    subroutine clone2_if(this, clone, n)
      import :: vector
      class(vector), intent(in) :: this
      class(vector), allocatable, intent(out) :: clone(:)
      integer, intent(in) :: n
    end subroutine
    subroutine setval_if(this, val)
      import :: vector, r8
      class(vector), intent(inout) :: this
      real(r8), intent(in) :: val
    end subroutine
    subroutine scale_if(this, a)
      import :: vector, r8
      class(vector), intent(inout) :: this
      real(r8), intent(in) :: a
    end subroutine
    function norm2_if(this) result(val)
      import :: vector, r8
      class(vector), intent(in) :: this
      real(r8) :: val
    end function
    subroutine copy_if(dest, src)
      import :: vector
      class(vector), intent(inout) :: dest
      class(vector), intent(in) :: src
    end subroutine
    subroutine update1_if(this, a, x)
      import :: vector, r8
      class(vector), intent(inout) :: this
      real(r8), intent(in) :: a
      class(vector), intent(in) :: x
    end subroutine
    subroutine update2_if(this, a, x, b)
      import :: vector, r8
      class(vector), intent(inout) :: this
      real(r8), intent(in) :: a, b
      class(vector), intent(in) :: x
    end subroutine
    subroutine update3_if(this, a, x, b, y)
      import :: vector, r8
      class(vector), intent(inout) :: this
      real(r8), intent(in) :: a, b
      class(vector), intent(in) :: x, y
    end subroutine
    subroutine update4_if(this, a, x, b, y, c)
      import :: vector, r8
      class(vector), intent(inout) :: this
      real(r8), intent(in) :: a, b, c
      class(vector), intent(in) :: x, y
    end subroutine
    function dot_if(x, y) result(val)
      import :: vector, r8
      class(vector), intent(in) :: x, y
      real(r8) :: val
    end function
  end interface

contains

  subroutine need_same_type(a, b, who)
    class(vector), intent(in) :: a, b
    character(*), intent(in) :: who
    if (.not. same_type_as(a, b)) then
      write(*,'(2a)') 'incompatible arguments to VECTOR%', who
      error stop 1
    end if
  end subroutine

  !! dest <- src
  recursive subroutine copy(dest, src)
    class(vector), intent(inout) :: dest
    class(vector), intent(in) :: src
    call need_same_type(dest, src, 'COPY')
    call dest%copy_(src)
  end subroutine

  recursive function dot(x, y) result(val)
    class(vector), intent(in) :: x, y
    real(r8) :: val
    call need_same_type(x, y, 'DOT')
    val = x%dot_(y)
  end function

  !! this <- a*x + this
  recursive subroutine update1(this, a, x)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    if (a == 0.0_r8) return
    call need_same_type(this, x, 'UPDATE')
    call this%update1_(a, x)
  end subroutine

  !! this <- a*x + b*this
  recursive subroutine update2(this, a, x, b)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x
    if (a == 0.0_r8) then
      call this%scale(b)
      return
    end if
    call need_same_type(this, x, 'UPDATE')
    call this%update2_(a, x, b)
  end subroutine

  !! this <- a*x + b*y + this
  recursive subroutine update3(this, a, x, b, y)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x, y
    if (a == 0.0_r8) then
      call update1(this, b, y)
    else if (b == 0.0_r8) then
      call update1(this, a, x)
    else
      call need_same_type(this, x, 'UPDATE')
      call need_same_type(this, y, 'UPDATE')
      call this%update3_(a, x, b, y)
    end if
  end subroutine

  !! this <- a*x + b*y + c*this
  recursive subroutine update4(this, a, x, b, y, c)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a, b, c
    class(vector), intent(in) :: x, y
    if (a == 0.0_r8) then
      call update2(this, b, y, c)
    else if (b == 0.0_r8) then
      call update2(this, a, x, c)
    else
      call need_same_type(this, x, 'UPDATE')
      call need_same_type(this, y, 'UPDATE')
      call this%update4_(a, x, b, y, c)
    end if
  end subroutine

  !! vals(j) = <this, ys(idx(j))>, j = 1..size(idx).  Default: one dot per entry.
  subroutine dot_many(this, ys, idx, vals)
    class(vector), intent(in) :: this
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals(:)
    integer :: j
    do j = 1, size(idx)
      vals(j) = this%dot(ys(idx(j)))
    end do
  end subroutine

  !! Two rows against the same vectors: vals_this(j) = <this, ys(idx(j))>,
  !! vals_other(j) = <other, ys(idx(j))>, cross = <other, this>.  Default: the
  !! individual dot() calls.
  subroutine dot_pair_many(this, other, ys, idx, vals_this, vals_other, cross)
    class(vector), intent(in) :: this, other
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_other(:), cross
    call this%dot_many(ys, idx, vals_this)
    cross = other%dot(this)
    call other%dot_many(ys, idx, vals_other)
  end subroutine

  !! this <- this + sum_j ( a(j)*xs(idx(j)) + b(j)*ys(idx(j)) ), applied in order j = 1..size(idx).
  subroutine update_many(this, a, xs, b, ys, idx)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a(:), b(:)
    class(vector), intent(in) :: xs(:), ys(:)
    integer, intent(in) :: idx(:)
    integer :: j
    do j = 1, size(idx)
      call this%update(a(j), xs(idx(j)), b(j), ys(idx(j)))
    end do
  end subroutine

  !! this <- this + sum_j a(j)*xs(idx(j)), applied in order j = 1..size(idx).
  subroutine axpy_many(this, a, xs, idx)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a(:)
    class(vector), intent(in) :: xs(:)
    integer, intent(in) :: idx(:)
    integer :: j
    do j = 1, size(idx)
      call this%update(a(j), xs(idx(j)))
    end do
  end subroutine

  !! s = || a*x + this ||_2 (F08V:237-238).  `stored` tells the caller whether `this`
  !! now HOLDS a*x + this (.true.: this default, the reference's update + norm2) or was
  !! left untouched (.false.: an override may leave the update to the next stage,
  !! scale_dot_pair_many, which is then called with pre_a = a and applies it there --
  !! one pass over `this` instead of two).
  function update_norm2(this, a, x, stored) result(s)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    logical, intent(out) :: stored
    real(r8) :: s
    call this%update(a, x)
    s = this%norm2()
    stored = .true.
  end function

  !! update_norm2 with an offer: s = || a*x + this ||_2 exactly as update_norm2 (this default body IS
  !! update_norm2; `fused` = .false.).  A type that can take, in the SAME pure-read pass over `this`, x
  !! and the ys, the raw inner products of d = a*x + this (nothing stored, `this` left untouched) answers
  !! fused = .true. and returns
  !!   vals_this(j) = <d, ys(idx(j))>, vals_x(j) = <x, ys(idx(j))>, cross = <x, d>        (d NOT normalised)
  !! The accelerator then scales the d-sums by 1/s itself -- the Gram row of the normalised pair as
  !! fl(<d,w_k>/s) rather than the sum of fl(d_i/s)*w_k,i: last-bit differences -- skips
  !! scale_dot_pair_many and hands the whole pending normalisation (pend_pre_a = a, pend_a = 1/s) to the
  !! combine stage, as after update_norm2 with stored = .false. and scale_dot_pair_many with
  !! scaled = .false.: two reductions per update instead of three.
  function update_norm2_dots(this, a, x, ys, idx, vals_this, vals_x, cross, stored, fused) result(s)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_x(:), cross
    logical, intent(out) :: stored, fused
    real(r8) :: s
    vals_this = 0.0_r8
    vals_x = 0.0_r8
    cross = 0.0_r8
    fused = .false.
    s = this%update_norm2(a, x, stored)
  end function

  !! Normalise the new pair and take both inner-product rows (F08V:255-264, 347):
  !!   [pre_a present: this <- pre_a*f + this, the update update_norm2 did not store ;]
  !!   v <- a*v ; this <- a*this ; [subtract: v <- v - this] ;
  !!   vals_this(j) = <this, ys(idx(j))> ; and, if the type takes them in the same pass (f_row = .true.),
  !!   vals_f(j) = <f, ys(idx(j))>, cross = <f, this>.
  !! `scaled` (optional, like `stored` of update_norm2): .true. = this default, `this` and v now HOLD
  !! the normalised pair; an override may answer .false. after a PURE-READ pass that only formed
  !! a*(pre_a*f + this) in registers for the inner products -- the caller then passes the pending
  !! normalisation (pend_a, pend_pre_a, pend_subtract[, pend_w]) to the combine stage, whose
  !! override applies it to entry 1 of its lists while it combines (a type that answers .false.
  !! must override update_many_keep / axpy_many_keep accordingly).
  subroutine scale_dot_pair_many(this, v, a, subtract, f, ys, idx, vals_this, vals_f, cross, pre_a, scaled, f_row)
    class(vector), intent(inout) :: this, v
    real(r8), intent(in) :: a
    logical, intent(in) :: subtract
    class(vector), intent(in) :: f
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_f(:), cross
    real(r8), intent(in), optional :: pre_a
    logical, intent(out), optional :: scaled, f_row
    !! THIS DEFAULT BODY is what a user type that overrides nothing runs: exactly the reference's calls at this
    !! point of accel_update, in its order -- scale v, scale w, the Gram row <w1,w_k> (F08V:255-264) -- and NOT the
    !! projection row: the reference asks for <f,w_j> only after the drop decisions and after w_new <- f
    !! (F08V:336-347), and so does the accelerator when f_row comes back .false. (see nka_type: accel_update).
    !! An override that streams the ys anyway takes both rows in its one pass and answers f_row = .true..
    if (present(pre_a)) call this%update(pre_a, f)
    call v%scale(a)
    call this%scale(a)
    if (subtract) call v%update(-1.0_r8, this)
    call this%dot_many(ys, idx, vals_this)
    vals_f = 0.0_r8
    cross = 0.0_r8
    if (present(scaled)) scaled = .true.
    if (present(f_row)) f_row = .false.
  end subroutine

  !! xs(keep_in) <- this ; this <- this + sum_j (a(j)*xs(idx(j)) + b(j)*ys(idx(j))) in
  !! order ; ys(keep_out) <- this  (F08V:336, 374, 382).
  !! The two ring stores are named by their INDEX into the arrays the stage already receives
  !! (and may therefore modify: intent(inout)), never as separate dummies aliasing an element of
  !! an intent(in) array -- Fortran's argument-aliasing rules would make that undefined for a
  !! vector type that lives in host memory.  keep_in / keep_out never occur in idx.
  !! pend_a present: xs(idx(1)), ys(idx(1)) are the RAW new pair that a pure-read scale_dot_pair_many
  !! left untouched; the override normalises them on the way (see there) -- entry idx(1) of xs and
  !! ys is then MODIFIED.  Never passed to a type whose scale_dot_pair_many stores (this default).
  subroutine update_many_keep(this, a, xs, b, ys, idx, keep_in, keep_out, pend_a, pend_pre_a, pend_subtract)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a(:), b(:)
    class(vector), intent(inout) :: xs(:), ys(:)
    integer, intent(in) :: idx(:)
    integer, intent(in) :: keep_in, keep_out
    real(r8), intent(in), optional :: pend_a, pend_pre_a
    logical, intent(in), optional :: pend_subtract
    if (present(pend_a) .or. present(pend_pre_a) .or. present(pend_subtract)) &
      error stop 'VECTOR%UPDATE_MANY_KEEP: a pending normalisation needs an override that applies it'
    call xs(keep_in)%copy(this)
    call this%update_many(a, xs, b, ys, idx)
    call ys(keep_out)%copy(this)
  end subroutine

  !! ws(keep_in) <- this ; this <- this + sum_j a(j)*xs(idx(j)) in order ; xs(keep_out) <- this.
  !! pend_a[, pend_pre_a]: the pending normalisation, compact storage -- xs(idx(1)) is the raw v of the
  !! new pair, ws(idx(1)) its raw w; both are MODIFIED by an override that applies it; see update_many_keep.
  subroutine axpy_many_keep(this, a, xs, idx, ws, keep_in, keep_out, pend_a, pend_pre_a)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a(:)
    class(vector), intent(inout) :: xs(:), ws(:)
    integer, intent(in) :: idx(:)
    integer, intent(in) :: keep_in, keep_out
    real(r8), intent(in), optional :: pend_a, pend_pre_a
    if (present(pend_a) .or. present(pend_pre_a)) &
      error stop 'VECTOR%AXPY_MANY_KEEP: a pending normalisation needs an override that applies it'
    call ws(keep_in)%copy(this)
    call this%axpy_many(a, xs, idx)
    call xs(keep_out)%copy(this)
  end subroutine


  subroutine vector_default_dot_many(this, ys, idx, vals)
    class(vector), intent(in) :: this
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals(:)
    call dot_many(this, ys, idx, vals)
  end subroutine

  subroutine vector_default_dot_pair_many(this, other, ys, idx, vals_this, vals_other, cross)
    class(vector), intent(in) :: this, other
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_other(:), cross
    call dot_pair_many(this, other, ys, idx, vals_this, vals_other, cross)
  end subroutine

  function vector_default_update_norm2(this, a, x, stored) result(s)
    class(vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    logical, intent(out) :: stored
    real(r8) :: s
    s = update_norm2(this, a, x, stored)
  end function

  subroutine vector_default_scale_dot_pair_many(this, v, a, subtract, f, ys, idx, vals_this, vals_f, cross, pre_a, scaled, f_row)
    class(vector), intent(inout) :: this, v
    real(r8), intent(in) :: a
    logical, intent(in) :: subtract
    class(vector), intent(in) :: f
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_f(:), cross
    real(r8), intent(in), optional :: pre_a
    logical, intent(out), optional :: scaled, f_row
    call scale_dot_pair_many(this, v, a, subtract, f, ys, idx, vals_this, vals_f, cross, pre_a, scaled, f_row)
  end subroutine

end module vector_class
