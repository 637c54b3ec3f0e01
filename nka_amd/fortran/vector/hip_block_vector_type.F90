!! hip_block_vector_type -- a concrete, DEVICE-RESIDENT vector for the
!! abstract-vector flavour (the MI355X counterpart of the reference's example
!! grid_vector, src-F08-vector/grid_vector_type.F90:44-197).
!!
!! A block vector of NFIELD fields of NPER doubles each (BASELINE config 5: 4
!! fields x 1e7).  The fields are consecutive views of ONE allocation in HBM, so
!! each deferred procedure of class(vector) is a single HIP kernel over all
!! fields through the C ABI (nka_hip_vec_*, nka_amd/csrc/vec_ops.hip), not one
!! launch (and, for dot/norm2, one host round trip) per field.  The elementwise
!! results are rounded like the Fortran expressions of grid_vector (a*x + b*y + z
!! evaluated left to right, no fused multiply-add); dot products are
!! deterministic two-stage reductions.
!!
!! UNREDUCED TAIL.  A vector may carry NTAIL extra values behind its fields that
!! every elementwise procedure covers and NO reduction sees -- the ghost / halo
!! values of the reference's grid_vector (grid_vector_type.F90:104-165 operate on
!! the whole array, :170-197 sum the interior only).  Keeping them BEHIND the
!! reduced values (instead of interleaved, as a ring around a 2-D array) lets every
!! reduction and every fused stage kernel run over one dense, aligned prefix; the
!! tail (O(sqrt(n)) values for a ghost ring) is finished by the plain elementwise
!! kernels.  hip_grid_vector_type.F90 builds the 2-D grid vector on this.
!!
!! PARALLEL-AWARE REDUCTIONS.  The vector flavour of the reference is distributed
!! through the vector class: "the implementation of the vector base class
!! reduction methods will necessarily be parallel-aware"
!! (src-F08-vector/README.md:16-22).  Here every rank holds a contiguous slice of
!! each field (plus, in the unreduced tail, whatever halo copies it needs) and the
!! vectors of a rank share one workspace; install on that workspace
!!   call hip_block_vector_use_rccl(ws, id128, nranks, rank)     built-in: RCCL over xGMI
!!   call hip_block_vector_set_allreduce(ws, fn, ctx)            device-side sum, stream-ordered
!!   call hip_block_vector_set_host_allreduce(ws, fn, ctx)       host-side sum (e.g. MPI_Allreduce)
!! and EVERY sum a reduction returns -- dot_, norm2 (before its square root),
!! dot_many, dot_pair_many, and the 1 resp. 2L+1 sums of the fused stages -- is
!! summed over the ranks before the accelerator sees it: one collective per
!! stage, three per update.  The accelerator object itself (module nka_type of
!! the vector flavour) needs no change, exactly as in the reference.

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
    integer(c_int64_t) :: ntot = 0              ! nfield*nper + the unreduced tail
    integer(c_int64_t) :: nred = 0              ! nfield*nper: the leading values that reductions see
    type(c_ptr) :: base = c_null_ptr            ! device pointer to ntot doubles; field k starts at (k-1)*nper
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
    !! fused overrides of the optional batched hooks (one kernel for many dots / axpys)
    procedure :: dot_many => dot_many_fused
    procedure :: dot_pair_many => dot_pair_many_fused
    procedure :: update_many => update_many_fused
    procedure :: axpy_many => axpy_many_fused
    !! fused stage hooks: each group of statements of an update as ONE kernel
    procedure :: update_norm2 => update_norm2_fused
    procedure :: update_norm2_dots => update_norm2_dots_fused
    procedure :: scale_dot_pair_many => scale_dot_pair_many_fused
    procedure :: update_many_keep => update_many_keep_fused
    procedure :: axpy_many_keep => axpy_many_keep_fused
    !! specific to this type
    procedure :: init
    procedure :: release
    procedure :: set_field          ! host -> device
    procedure :: get_field          ! device -> host
  end type

  public :: hip_block_vector_workspace
  public :: hip_block_vector_use_rccl, hip_block_vector_set_allreduce, hip_block_vector_set_host_allreduce
  public :: hip_block_vector_allreduce_now
  public :: hip_block_vector_set_sum_order, NKA_HIP_SUMS_REFERENCE_ORDER, NKA_HIP_SUMS_BLOCKED, NKA_HIP_SUMS_BLOCKED_ROUNDED

contains

  !! One workspace (device ordinal, stream, scratch) shared by all vectors.
  function hip_block_vector_workspace(device) result(ws)
    integer, intent(in) :: device
    type(c_ptr) :: ws
    call nka_hip_check(nka_hip_vec_workspace_create(ws, int(device, c_int32_t), c_null_ptr), 'vec_workspace_create')
  end function

  !! Built-in reduction hook: one RCCL all-reduce per reduction on the workspace stream.
  !! id128 comes from nka_hip_comm_unique_id on one rank and reaches the others by any means.
  subroutine hip_block_vector_use_rccl(ws, id128, nranks, rank)
    type(c_ptr), intent(in) :: ws
    character(kind=c_char), intent(in) :: id128(128)
    integer, intent(in) :: nranks, rank
    call nka_hip_check(nka_hip_vec_comm_init_rank(ws, id128, int(nranks, c_int32_t), int(rank, c_int32_t)), &
                       'hip_block_vector_use_rccl')
  end subroutine

  !! fn: bind(C) integer(c_int) function fn(ctx, buf, count, stream) summing `count` doubles at
  !! DEVICE address buf over all ranks in place, ordered on the hipStream_t (nka_hip_allreduce_fn)
  subroutine hip_block_vector_set_allreduce(ws, fn, ctx)
    type(c_ptr), intent(in) :: ws, ctx
    type(c_funptr), intent(in) :: fn
    call nka_hip_check(nka_hip_vec_set_allreduce(ws, fn, ctx), 'hip_block_vector_set_allreduce')
  end subroutine

  !! fn: bind(C) integer(c_int) function fn(ctx, vals, count) summing `count` doubles in HOST
  !! memory over all ranks in place (nka_hip_host_allreduce_fn), e.g. a wrapper of MPI_Allreduce
  subroutine hip_block_vector_set_host_allreduce(ws, fn, ctx)
    type(c_ptr), intent(in) :: ws, ctx
    type(c_funptr), intent(in) :: fn
    call nka_hip_check(nka_hip_vec_set_host_allreduce(ws, fn, ctx), 'hip_block_vector_set_host_allreduce')
  end subroutine

  !! the installed hooks, once, on host values: lets a launcher prove the communicator
  subroutine hip_block_vector_allreduce_now(ws, vals)
    type(c_ptr), intent(in) :: ws
    real(r8), intent(inout), contiguous :: vals(:)
    call nka_hip_check(nka_hip_vec_allreduce_now(ws, vals, size(vals, kind=c_int32_t)), 'hip_block_vector_allreduce_now')
  end subroutine

  subroutine init(this, nfield, nper, ws, ntail)
    class(hip_block_vector), intent(inout) :: this
    integer, intent(in) :: nfield
    integer(c_int64_t), intent(in) :: nper
    type(c_ptr), intent(in) :: ws
    integer(c_int64_t), intent(in), optional :: ntail   ! values behind the fields that reductions skip
    call this%release
    this%nfield = nfield
    this%nper = nper
    this%nred = nfield * nper
    this%ntot = this%nred
    if (present(ntail)) then
      if (ntail < 0) error stop 'hip_block_vector%init: negative tail'
      this%ntot = this%nred + ntail
    end if
    this%ws = ws
    call nka_hip_check(nka_hip_vec_alloc(ws, this%ntot, this%base), 'vec_alloc')
  end subroutine

  !! Device memory is released explicitly (clones made by the accelerator live
  !! as long as the accelerator object).
  subroutine release(this)
    class(hip_block_vector), intent(inout) :: this
    if (c_associated(this%base)) call nka_hip_check(nka_hip_vec_free(this%ws, this%base), 'vec_free')
    this%base = c_null_ptr
    this%nfield = 0
    this%ntot = 0
    this%nred = 0
  end subroutine

  !! device address of the unreduced tail of `this`
  type(c_ptr) function tail_ptr(this)
    class(hip_block_vector), intent(in) :: this
    integer(c_intptr_t) :: addr
    addr = transfer(this%base, addr) + int(this%nred, c_intptr_t) * 8_c_intptr_t
    tail_ptr = transfer(addr, tail_ptr)
  end function

  !! device address of field k
  type(c_ptr) function field_ptr(this, k)
    class(hip_block_vector), intent(in) :: this
    integer, intent(in) :: k
    integer(c_intptr_t) :: addr
    if (k < 1 .or. k > this%nfield) error stop 'hip_block_vector: field index out of range'
    addr = transfer(this%base, addr) + int(k-1, c_intptr_t) * int(this%nper, c_intptr_t) * 8_c_intptr_t
    field_ptr = transfer(addr, field_ptr)
  end function

  subroutine set_field(this, k, array)
    class(hip_block_vector), intent(inout) :: this
    integer, intent(in) :: k
    real(r8), intent(in), contiguous :: array(:)
    if (size(array, kind=c_int64_t) /= this%nper) error stop 'hip_block_vector%set_field: wrong size'
    call nka_hip_check(nka_hip_vec_h2d(this%ws, this%nper, field_ptr(this, k), array), 'vec_h2d')
  end subroutine

  subroutine get_field(this, k, array)
    class(hip_block_vector), intent(in) :: this
    integer, intent(in) :: k
    real(r8), intent(out), contiguous :: array(:)
    if (size(array, kind=c_int64_t) /= this%nper) error stop 'hip_block_vector%get_field: wrong size'
    call nka_hip_check(nka_hip_vec_d2h(this%ws, this%nper, array, field_ptr(this, k)), 'vec_d2h')
  end subroutine

  !! clone: same structure, NEW device storage, values undefined (vector_class.F90:93-101)
  subroutine clone1(this, clone)
    class(hip_block_vector), intent(in) :: this
    class(vector), allocatable, intent(out) :: clone
    allocate(hip_block_vector :: clone)
    select type (clone)
    type is (hip_block_vector)
      call clone%init(this%nfield, this%nper, this%ws, this%ntot - this%nred)
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
        call clone(k)%init(this%nfield, this%nper, this%ws, this%ntot - this%nred)
      end do
    end select
  end subroutine

  subroutine copy_(dest, src)
    class(hip_block_vector), intent(inout) :: dest
    class(vector), intent(in) :: src
    select type (src)
    class is (hip_block_vector)
      call nka_hip_check(nka_hip_vec_copy(dest%ws, dest%ntot, dest%base, src%base), 'vec_copy')
    end select
  end subroutine

  subroutine setval(this, val)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: val
    call nka_hip_check(nka_hip_vec_setval(this%ws, this%ntot, this%base, val), 'vec_setval')
  end subroutine

  subroutine scale(this, a)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    call nka_hip_check(nka_hip_vec_scale(this%ws, this%ntot, this%base, a), 'vec_scale')
  end subroutine

  subroutine update1_(this, a, x)               ! this <- a*x + this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    select type (x)
    class is (hip_block_vector)
      call nka_hip_check(nka_hip_vec_update1(this%ws, this%ntot, this%base, a, x%base), 'vec_update1')
    end select
  end subroutine

  subroutine update2_(this, a, x, b)            ! this <- a*x + b*this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x
    select type (x)
    class is (hip_block_vector)
      call nka_hip_check(nka_hip_vec_update2(this%ws, this%ntot, this%base, a, x%base, b), 'vec_update2')
    end select
  end subroutine

  subroutine update3_(this, a, x, b, y)         ! this <- a*x + b*y + this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b
    class(vector), intent(in) :: x, y
    select type (x)
    class is (hip_block_vector)
      select type (y)
      class is (hip_block_vector)
        call nka_hip_check(nka_hip_vec_update3(this%ws, this%ntot, this%base, a, x%base, b, y%base), 'vec_update3')
      end select
    end select
  end subroutine

  subroutine update4_(this, a, x, b, y, c)      ! this <- a*x + b*y + c*this
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a, b, c
    class(vector), intent(in) :: x, y
    select type (x)
    class is (hip_block_vector)
      select type (y)
      class is (hip_block_vector)
        call nka_hip_check(nka_hip_vec_update4(this%ws, this%ntot, this%base, a, x%base, b, y%base, c), 'vec_update4')
      end select
    end select
  end subroutine

  function dot_(x, y) result(val)
    class(hip_block_vector), intent(in) :: x
    class(vector), intent(in) :: y
    real(r8) :: val
    val = 0.0_r8
    select type (y)
    class is (hip_block_vector)
      call nka_hip_check(nka_hip_vec_dot(x%ws, x%nred, x%base, y%base, val), 'vec_dot')
    end select
  end function

  function norm2_(this) result(val)
    class(hip_block_vector), intent(in) :: this
    real(r8) :: val
    call nka_hip_check(nka_hip_vec_norm2(this%ws, this%nred, this%base, val), 'vec_norm2')
  end function

  !! vals(j) = <this, ys(idx(j))>: `this` is read once while the ys stream past.
  subroutine dot_many_fused(this, ys, idx, vals)
    class(hip_block_vector), intent(in) :: this
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals(:)
    type(c_ptr) :: ptrs(size(idx))
    integer :: j
    if (size(idx) == 0) return
    if (reference_order(this)) then            ! the reference's own loop of dot() calls, each summed in its order
      call vector_default_dot_many(this, ys, idx, vals)
      return
    end if
    select type (ys)
    class is (hip_block_vector)
      do j = 1, size(idx)
        ptrs(j) = ys(idx(j))%base
      end do
    class default
      error stop 'incompatible arguments to VECTOR%DOT_MANY'
    end select
    call nka_hip_check(nka_hip_vec_dot_many(this%ws, this%nred, this%base, ptrs, size(idx, kind=c_int32_t), vals), &
                       'vec_dot_many')
  end subroutine

  subroutine dot_pair_many_fused(this, other, ys, idx, vals_this, vals_other, cross)
    class(hip_block_vector), intent(in) :: this
    class(vector), intent(in) :: other
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_other(:), cross
    type(c_ptr) :: ptrs(max(size(idx),1))
    integer :: j
    if (reference_order(this)) then
      call vector_default_dot_pair_many(this, other, ys, idx, vals_this, vals_other, cross)
      return
    end if
    select type (other)
    class is (hip_block_vector)
      select type (ys)
      class is (hip_block_vector)
        do j = 1, size(idx)
          ptrs(j) = ys(idx(j))%base
        end do
        call nka_hip_check(nka_hip_vec_dot_pair_many(this%ws, this%nred, this%base, other%base, ptrs, &
                           size(idx, kind=c_int32_t), vals_this, vals_other, cross), 'vec_dot_pair_many')
      class default
        error stop 'incompatible arguments to VECTOR%DOT_PAIR_MANY'
      end select
    class default
      error stop 'incompatible arguments to VECTOR%DOT_PAIR_MANY'
    end select
  end subroutine

  !! this <- (a(j)*xs(idx(j)) + b(j)*ys(idx(j))) + this, j in order: the rounding of
  !! successive update3_ calls, with `this` read and written once.
  subroutine update_many_fused(this, a, xs, b, ys, idx)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a(:), b(:)
    class(vector), intent(in) :: xs(:), ys(:)
    integer, intent(in) :: idx(:)
    type(c_ptr) :: xp(size(idx)), yp(size(idx))
    integer :: j
    if (size(idx) == 0) return
    select type (xs)
    class is (hip_block_vector)
      select type (ys)
      class is (hip_block_vector)
        do j = 1, size(idx)
          xp(j) = xs(idx(j))%base
          yp(j) = ys(idx(j))%base
        end do
      class default
        error stop 'incompatible arguments to VECTOR%UPDATE_MANY'
      end select
    class default
      error stop 'incompatible arguments to VECTOR%UPDATE_MANY'
    end select
    call nka_hip_check(nka_hip_vec_update_many(this%ws, this%ntot, this%base, a, xp, b, yp, &
                                               size(idx, kind=c_int32_t)), 'vec_update_many')
  end subroutine

  !! this <- a(j)*xs(idx(j)) + this, j in order (the rounding of successive update1_ calls)
  subroutine axpy_many_fused(this, a, xs, idx)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a(:)
    class(vector), intent(in) :: xs(:)
    integer, intent(in) :: idx(:)
    type(c_ptr) :: xp(size(idx))
    integer :: j
    if (size(idx) == 0) return
    select type (xs)
    class is (hip_block_vector)
      do j = 1, size(idx)
        xp(j) = xs(idx(j))%base
      end do
    class default
      error stop 'incompatible arguments to VECTOR%AXPY_MANY'
    end select
    call nka_hip_check(nka_hip_vec_axpy_many(this%ws, this%ntot, this%base, a, xp, size(idx, kind=c_int32_t)), &
                       'vec_axpy_many')
  end subroutine

  !! || a*x + this || in one pure-read pass (R 2n); the update itself is left to the
  !! next stage (stored = .false.), which reads `this` and x = f anyway.
  function update_norm2_fused(this, a, x, stored) result(s)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    logical, intent(out) :: stored
    real(r8) :: s
    s = 0.0_r8
    stored = .false.
    if (reference_order(this)) then            ! update, then norm2: F08V:237-238 as the reference issues them
      s = vector_default_update_norm2(this, a, x, stored)
      return
    end if
    select type (x)
    class is (hip_block_vector)
      call nka_hip_check(nka_hip_vec_update_norm2(this%ws, this%nred, this%base, a, x%base, 0_c_int32_t, s), &
                         'vec_update_norm2')
    class default
      error stop 'incompatible arguments to VECTOR%UPDATE_NORM2'
    end select
  end function

  !! The norm AND both inner-product rows in one pure-read pass (R (2+L)n): the raw sums of
  !! d = a*x + this; nothing is stored, the accelerator scales by 1/s and the combine stage
  !! normalises the pair.  (NKA_HIP_VEC_FUSE_NORM=0, or the deferral switched off: the separate stages.)
  function update_norm2_dots_fused(this, a, x, ys, idx, vals_this, vals_x, cross, stored, fused) result(s)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a
    class(vector), intent(in) :: x
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_x(:), cross
    logical, intent(out) :: stored, fused
    real(r8) :: s, dd
    type(c_ptr) :: ptrs(max(size(idx),1))
    integer :: j
    vals_this = 0.0_r8
    vals_x = 0.0_r8
    cross = 0.0_r8
    fused = defer_scale_enabled() .and. fuse_norm_enabled() .and. .not. reference_order(this) &     ! (any list length since round 5:
            .and. nka_hip_vec_get_sum_order(this%ws) /= NKA_HIP_SUMS_BLOCKED_ROUNDED                ! (rounded: the norm stays a pass of its own)
                                                                  ! the library runs balanced groups of at most 24 vectors)
    if (.not. fused) then
      s = update_norm2_fused(this, a, x, stored)
      return
    end if
    stored = .false.
    select type (x)
    class is (hip_block_vector)
      select type (ys)
      class is (hip_block_vector)
        do j = 1, size(idx)
          ptrs(j) = ys(idx(j))%base
        end do
        call nka_hip_check(nka_hip_vec_diff_norm_dot_pair_many(this%ws, this%nred, this%base, a, x%base, ptrs, &
                           size(idx, kind=c_int32_t), dd, vals_this, vals_x, cross), 'vec_diff_norm_dot_pair_many')
        s = sqrt(dd)
        return
      end select
    end select
    error stop 'incompatible arguments to VECTOR%UPDATE_NORM2_DOTS'
  end function

  !! [apply the deferred update,] scale both members of the new pair and take both
  !! inner-product rows while the stored vectors stream past once (R (3+L)n, W 2n).
  subroutine scale_dot_pair_many_fused(this, v, a, subtract, f, ys, idx, vals_this, vals_f, cross, pre_a, scaled, f_row)
    class(hip_block_vector), intent(inout) :: this
    class(vector), intent(inout) :: v
    real(r8), intent(in) :: a
    logical, intent(in) :: subtract
    class(vector), intent(in) :: f
    class(vector), intent(in) :: ys(:)
    integer, intent(in) :: idx(:)
    real(r8), intent(out) :: vals_this(:), vals_f(:), cross
    real(r8), intent(in), optional :: pre_a
    logical, intent(out), optional :: scaled, f_row
    type(c_ptr) :: ptrs(max(size(idx),1))
    logical :: defer
    integer :: j
    integer(c_int32_t) :: pre
    integer(c_int64_t) :: nt
    real(r8) :: pa
    pre = 0
    pa = 0.0_r8
    if (present(pre_a)) then
      pre = 1
      pa = pre_a
    end if
    if (reference_order(this)) then            ! scale v, scale w, the Gram row by dot(): F08V:255-264 as the reference issues them
      call vector_default_scale_dot_pair_many(this, v, a, subtract, f, ys, idx, vals_this, vals_f, cross, pre_a, scaled, f_row)
      return
    end if
    if (present(f_row)) f_row = .true.          ! both inner-product rows come out of the one pass
    select type (v)
    class is (hip_block_vector)
      select type (f)
      class is (hip_block_vector)
        select type (ys)
        class is (hip_block_vector)
          do j = 1, size(idx)
            ptrs(j) = ys(idx(j))%base
          end do
          !! Asked whether it stored: a PURE-READ pass (R (2+L)n, no
          !! store stream); the pair is normalised by the combine stage, which reads it anyway.
          defer = present(scaled) .and. defer_scale_enabled()
          if (present(scaled)) scaled = .not. defer
          if (defer) then
            call nka_hip_check(nka_hip_vec_dot_pair_many_scaled(this%ws, this%nred, this%base, a, pre, pa, f%base, ptrs, &
                               size(idx, kind=c_int32_t), vals_this, vals_f, cross), 'vec_dot_pair_many_scaled')
            return
          end if
          call nka_hip_check(nka_hip_vec_scale_dot_pair_many(this%ws, this%nred, this%base, v%base, a, &
                             merge(1_c_int32_t, 0_c_int32_t, subtract), pre, pa, f%base, ptrs, &
                             size(idx, kind=c_int32_t), vals_this, vals_f, cross), 'vec_scale_dot_pair_many')
          nt = this%ntot - this%nred
          if (nt > 0) then   ! the elementwise statements of the stage on the unreduced tail, hook by hook
            if (pre /= 0) call nka_hip_check(nka_hip_vec_update1(this%ws, nt, tail_ptr(this), pa, tail_ptr(f)), 'vec_update1')
            call nka_hip_check(nka_hip_vec_scale(this%ws, nt, tail_ptr(this), a), 'vec_scale')
            call nka_hip_check(nka_hip_vec_scale(this%ws, nt, tail_ptr(v), a), 'vec_scale')
            if (subtract) call nka_hip_check(nka_hip_vec_update1(this%ws, nt, tail_ptr(v), -1.0_r8, tail_ptr(this)), &
                                             'vec_update1')
          end if
          return
        end select
      end select
    end select
    error stop 'incompatible arguments to VECTOR%SCALE_DOT_PAIR_MANY'
  end subroutine

  !! keep_in <- this ; combine ; keep_out <- this, `this` read once and written once
  !! (R (1+2k)n, W 3n).
  subroutine update_many_keep_fused(this, a, xs, b, ys, idx, keep_in, keep_out, pend_a, pend_pre_a, pend_subtract)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a(:), b(:)
    class(vector), intent(inout) :: xs(:), ys(:)
    integer, intent(in) :: idx(:)
    integer, intent(in) :: keep_in, keep_out     ! xs(keep_in) <- this before, ys(keep_out) <- this after the combine
    real(r8), intent(in), optional :: pend_a, pend_pre_a
    logical, intent(in), optional :: pend_subtract
    type(c_ptr) :: xp(max(size(idx),1)), yp(max(size(idx),1)), kin, kout, kin_tail, kout_tail
    integer :: j
    integer(c_int32_t) :: ppre, psub
    real(r8) :: ppa
    ppre = 0
    ppa = 0.0_r8
    psub = 0
    if (present(pend_pre_a)) then
      ppre = 1
      ppa = pend_pre_a
    end if
    if (present(pend_subtract)) psub = merge(1_c_int32_t, 0_c_int32_t, pend_subtract)
    select type (xs)
    class is (hip_block_vector)
      select type (ys)
      class is (hip_block_vector)
        kin = xs(keep_in)%base
        kout = ys(keep_out)%base
        kin_tail = tail_ptr(xs(keep_in))
        kout_tail = tail_ptr(ys(keep_out))
        do j = 1, size(idx)
          xp(j) = xs(idx(j))%base
          yp(j) = ys(idx(j))%base
        end do
        if (present(pend_a)) then      ! entry 1 is the raw new pair: normalised on the way
          call nka_hip_check(nka_hip_vec_update_many_keep_pend(this%ws, this%nred, this%base, a, xp, b, yp, &
                             size(idx, kind=c_int32_t), kin, kout, pend_a, ppre, ppa, psub), &
                             'vec_update_many_keep_pend')
        else
          call nka_hip_check(nka_hip_vec_update_many_keep(this%ws, this%nred, this%base, a, xp, b, yp, &
                             size(idx, kind=c_int32_t), kin, kout), 'vec_update_many_keep')
        end if
        if (this%ntot > this%nred) then   ! the same statements on the unreduced tail
          do j = 1, size(idx)
            xp(j) = tail_ptr(xs(idx(j)))
            yp(j) = tail_ptr(ys(idx(j)))
          end do
          if (present(pend_a)) then
            call nka_hip_check(nka_hip_vec_update_many_keep_pend(this%ws, this%ntot - this%nred, tail_ptr(this), a, xp, &
                               b, yp, size(idx, kind=c_int32_t), kin_tail, kout_tail, pend_a, ppre, &
                               ppa, psub), 'vec_update_many_keep_pend')
          else
            call nka_hip_check(nka_hip_vec_update_many_keep(this%ws, this%ntot - this%nred, tail_ptr(this), a, xp, b, &
                               yp, size(idx, kind=c_int32_t), kin_tail, kout_tail), &
                               'vec_update_many_keep')
          end if
        end if
        return
      end select
    end select
    error stop 'incompatible arguments to VECTOR%UPDATE_MANY_KEEP'
  end subroutine

  subroutine axpy_many_keep_fused(this, a, xs, idx, ws, keep_in, keep_out, pend_a, pend_pre_a)
    class(hip_block_vector), intent(inout) :: this
    real(r8), intent(in) :: a(:)
    class(vector), intent(inout) :: xs(:), ws(:)
    integer, intent(in) :: idx(:)
    integer, intent(in) :: keep_in, keep_out     ! ws(keep_in) <- this before, xs(keep_out) <- this after the combine
    real(r8), intent(in), optional :: pend_a, pend_pre_a
    type(c_ptr) :: xp(max(size(idx),1)), pw, pwt, kin, kout, kin_tail, kout_tail
    integer :: j
    integer(c_int32_t) :: ppre
    real(r8) :: ppa
    ppre = 0
    ppa = 0.0_r8
    if (present(pend_pre_a)) then
      ppre = 1
      ppa = pend_pre_a
    end if
    select type (xs)
    class is (hip_block_vector)
      select type (ws)
      class is (hip_block_vector)
        kin = ws(keep_in)%base
        kout = xs(keep_out)%base
        kin_tail = tail_ptr(ws(keep_in))
        kout_tail = tail_ptr(xs(keep_out))
        pw = c_null_ptr
        pwt = c_null_ptr
        if (present(pend_a)) then              ! entry 1 is the raw v of the new pair, ws(idx(1)) its raw w
          pw = ws(idx(1))%base
          pwt = tail_ptr(ws(idx(1)))
        end if
        do j = 1, size(idx)
          xp(j) = xs(idx(j))%base
        end do
        if (present(pend_a)) then
          call nka_hip_check(nka_hip_vec_axpy_many_keep_pend(this%ws, this%nred, this%base, a, xp, &
                             size(idx, kind=c_int32_t), kin, kout, pw, pend_a, ppre, ppa), &
                             'vec_axpy_many_keep_pend')
        else
          call nka_hip_check(nka_hip_vec_axpy_many_keep(this%ws, this%nred, this%base, a, xp, &
                             size(idx, kind=c_int32_t), kin, kout), 'vec_axpy_many_keep')
        end if
        if (this%ntot > this%nred) then   ! the same statements on the unreduced tail
          do j = 1, size(idx)
            xp(j) = tail_ptr(xs(idx(j)))
          end do
          if (present(pend_a)) then
            call nka_hip_check(nka_hip_vec_axpy_many_keep_pend(this%ws, this%ntot - this%nred, tail_ptr(this), a, xp, &
                               size(idx, kind=c_int32_t), kin_tail, kout_tail, pwt, pend_a, ppre, ppa), &
                               'vec_axpy_many_keep_pend')
          else
            call nka_hip_check(nka_hip_vec_axpy_many_keep(this%ws, this%ntot - this%nred, tail_ptr(this), a, xp, &
                               size(idx, kind=c_int32_t), kin_tail, kout_tail), 'vec_axpy_many_keep')
          end if
        end if
        return
      end select
    end select
    error stop 'incompatible arguments to VECTOR%AXPY_MANY_KEEP'
  end subroutine

  !! The workspace sums in the reference's order (hip_block_vector_set_sum_order): the reduction-bearing batched / stage
  !! hooks then run their DEFAULT bodies -- the reference's own sequence of deferred hook calls (vector_class.F90) -- and
  !! dot() / norm2() sum element after element, so that the accelerator returns the reference's bits.
  logical function reference_order(this)
    class(hip_block_vector), intent(in) :: this
    reference_order = nka_hip_vec_get_sum_order(this%ws) == NKA_HIP_SUMS_REFERENCE_ORDER
  end function

  !! call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER | NKA_HIP_SUMS_BLOCKED | NKA_HIP_SUMS_BLOCKED_ROUNDED):
  !! nka_hip_vec_set_sum_order for every vector that shares the workspace (include/nka_hip.h).  _BLOCKED_ROUNDED keeps the norm stage
  !! a pass of its own, so that the Gram row is summed on the rounded pair (what NKA_HIP_VEC_FUSE_NORM=0 did for the tests).
  subroutine hip_block_vector_set_sum_order(ws, order)
    type(c_ptr), intent(in) :: ws
    integer, intent(in) :: order
    call nka_hip_check(nka_hip_vec_set_sum_order(ws, int(order, c_int32_t)), 'vec_set_sum_order')
  end subroutine

  !! NKA_HIP_VEC_FUSE_NORM=0: the norm stage stays a pass of its own (test / A/B aid; read once)
  logical function fuse_norm_enabled()
    logical, save :: known = .false., on = .true.
    character(len=8) :: val
    integer :: stat
    if (.not. known) then
      call get_environment_variable('NKA_HIP_VEC_FUSE_NORM', val, status=stat)
      if (stat == 0) on = .not. (val(1:1) == '0')
      known = .true.
    end if
    fuse_norm_enabled = on
  end function

  !! NKA_HIP_VEC_DEFER_SCALE=0: scale_dot_pair_many always stores (A/B aid; read once)
  logical function defer_scale_enabled()
    logical, save :: known = .false., on = .true.
    character(len=8) :: val
    integer :: stat
    if (.not. known) then
      call get_environment_variable('NKA_HIP_VEC_DEFER_SCALE', val, status=stat)
      if (stat == 0) on = .not. (val(1:1) == '0')
      known = .true.
    end if
    defer_scale_enabled = on
  end function

end module hip_block_vector_type
