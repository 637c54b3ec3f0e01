!! nka_type (array flavour) -- Fortran host side of the MI355X NKA accelerator.
!!
!! Drop-in for module nka_type of the reference (src-F08/nka_type.F90:148-181):
!! the same derived type name, the same type-bound procedure names and argument
!! meaning --
!!   init(vlen, mvec)  set_vec_tol(vtol)  accel_update(f)  restart()  relax()
!!   num_vec()  max_vec()  vec_len()  vec_tol()  defined()
!! -- but the object is only a handle: every vector, the Gram/Cholesky matrix
!! and the linked lists live on the GPU, and each call goes through the C ABI of
!! libnka_hip.so (module nka_hip_c, iso_c_binding).
!!
!! Differences a caller can see:
!!  * accel_update(f) with a host array keeps the reference signature
!!    (real(r8), intent(inout) :: f(:), F08:252) and copies f to the device and
!!    back; accel_update_dev(f_dev) takes device memory (type(c_ptr)) and is the
!!    entry a GPU-resident solver uses (no PCIe traffic, asynchronous).
!!  * the FAST distribution hook is set_allreduce / use_rccl: the local partial
!!    sums are already on the device, so what a sharded run needs is their global
!!    SUM (one call per update, 2+2*mvec doubles), not a dot product.
!!    set_dot_prod(dot_prod) (F08:209-219) is kept for source compatibility,
!!    same abstract interface: the update then evaluates the reference's own
!!    sequence of dot_prod calls on HOST copies of the operands (2+L vectors cross
!!    PCIe per update; the scalar step, the combine and the stores stay on the
!!    device) -- a slow path for callers that cannot change, see
!!    nka_hip_set_host_dot in include/nka_hip.h.
!!  * b = a is a DEEP copy, as for the reference's type with its allocatable
!!    components (F08:154-168): the defined assignment below clones the device
!!    object (nka_hip_clone: device-to-device copies of v, w, lists, factor) and
!!    carries a user dot product over; the two objects then evolve separately.
!!  * init takes optional flavor / device / stream arguments.  Without `flavor` the
!!    object runs the build's DEFAULT: compact storage (the v slot of a normalised
!!    pair keeps v - w, the combine is f + c*(v - w), the src-C statement), which
!!    differs from F08:397 in the association of the combine only (last bits; held
!!    to the compiled src-F08 reference in tests/) and moves 8n(9+L+k) instead of
!!    8n(8+L+2k) bytes per update.  flavor=NKA_HIP_FLAVOR_F08 or the environment
!!    variable NKA_HIP_FLAVOR=f08 selects the F08 statement bit for bit.

module nka_type

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use, intrinsic :: iso_c_binding
  use nka_hip_c
  implicit none
  private

  !! user dot product, the reference's interface (F08:216-219: assumed-shape x, y)
  abstract interface
    function dp(x, y)
      import :: r8
      real(r8), intent(in) :: x(:), y(:)
      real(r8) :: dp
    end function
  end interface

  !! what the C-side trampoline needs to find the user's procedure again
  type :: dp_holder
    procedure(dp), pointer, nopass :: fn => null()
  end type

  type, public :: nka
    private
    type(c_ptr) :: handle = c_null_ptr
    type(dp_holder), pointer :: user_dp => null()
  contains
    procedure :: init
    procedure :: set_vec_tol
    procedure :: set_dot_prod
    procedure :: set_allreduce
    procedure :: use_rccl
    procedure :: p2p_export
    procedure :: p2p_attach
    procedure :: p2p_detach
    procedure :: vec_len
    procedure :: num_vec
    procedure :: max_vec
    procedure :: vec_tol
    procedure :: accel_update
    procedure :: accel_update_dev
    procedure :: accel_update_swap
    procedure :: list_bound
    procedure :: set_sum_order
    procedure :: set_shard
    procedure :: relax
    procedure :: restart
    procedure :: defined
    procedure :: flavor => get_flavor
    procedure :: set_timing
    procedure :: get_timing
    procedure, private :: deep_copy
    generic :: assignment(=) => deep_copy
    final :: nka_delete
  end type nka

  public :: NKA_HIP_FLAVOR_F08, NKA_HIP_FLAVOR_F08_VECTOR, NKA_HIP_FLAVOR_C, NKA_HIP_FLAVOR_DEFAULT
  public :: NKA_HIP_SUMS_AUTO, NKA_HIP_SUMS_REFERENCE_ORDER, NKA_HIP_SUMS_BLOCKED, NKA_HIP_SUMS_BLOCKED_ROUNDED

contains

  !! call a%init(vlen, mvec)                                  F08:185-200
  subroutine init(this, vlen, mvec, flavor, device, stream)
    class(nka), intent(inout) :: this
    integer, intent(in) :: vlen, mvec
    integer, intent(in), optional :: flavor, device
    type(c_ptr), intent(in), optional :: stream
    integer(c_int32_t) :: fl, dev
    type(c_ptr) :: st
    call nka_delete_handle(this)       ! intent(out) semantics: a re-init starts afresh
    fl = NKA_HIP_FLAVOR_DEFAULT        ! NKA_HIP_FLAVOR, else compact storage (include/nka_hip.h)
    dev = 0
    st = c_null_ptr
    if (present(flavor)) fl = int(flavor, c_int32_t)
    if (present(device)) dev = int(device, c_int32_t)
    if (present(stream)) st = stream
    call nka_hip_check(nka_hip_create(this%handle, int(vlen, c_int64_t), int(mvec, c_int32_t), &
                                      0.01_c_double, fl, dev, st), 'nka%init')
  end subroutine

  subroutine nka_delete_handle(this)
    class(nka), intent(inout) :: this
    integer(c_int) :: rc
    if (c_associated(this%handle)) rc = nka_hip_destroy(this%handle)
    this%handle = c_null_ptr
    if (associated(this%user_dp)) deallocate(this%user_dp)
  end subroutine

  !! b = a.  The reference type has allocatable components and copies deeply
  !! (F08:154-168); this one is a device handle, so the defined assignment clones
  !! the device object: afterwards lhs and rhs are independent accelerators.
  subroutine deep_copy(lhs, rhs)
    class(nka), intent(inout) :: lhs
    class(nka), intent(in) :: rhs
    if (c_associated(lhs%handle, rhs%handle)) return           ! a = a
    call nka_delete_handle(lhs)
    if (.not. c_associated(rhs%handle)) return                 ! copy of an object that was never initialised
    call nka_hip_check(nka_hip_clone(rhs%handle, lhs%handle), 'nka assignment (deep copy)')
    if (associated(rhs%user_dp)) then                          ! the dp pointer component travels too (F08:161)
      allocate(lhs%user_dp)
      lhs%user_dp%fn => rhs%user_dp%fn
      call nka_hip_check(nka_hip_set_host_dot(lhs%handle, c_funloc(host_dot_trampoline), c_loc(lhs%user_dp)), &
                         'nka assignment (deep copy)')
    end if
  end subroutine

  !! call a%set_dot_prod(dot_prod)                             F08:209-214
  !! dot_prod must return the GLOBAL dot product of its arguments (F08:58-64).
  subroutine set_dot_prod(this, dot_prod)
    class(nka), intent(inout) :: this
    procedure(dp), pointer, intent(in) :: dot_prod
    if (.not.associated(dot_prod)) error stop 'nka%set_dot_prod: dot_prod is not associated'   ! F08:212
    if (.not.associated(this%user_dp)) allocate(this%user_dp)
    this%user_dp%fn => dot_prod
    call nka_hip_check(nka_hip_set_host_dot(this%handle, c_funloc(host_dot_trampoline), c_loc(this%user_dp)), &
                       'nka%set_dot_prod')
  end subroutine

  !! nka_hip_host_dot_fn: C calls this with host copies of the operands.
  function host_dot_trampoline(ctx, n, x, y) bind(C) result(d)
    type(c_ptr), value :: ctx
    integer(c_int64_t), value :: n
    real(c_double), intent(in) :: x(*), y(*)
    real(c_double) :: d
    type(dp_holder), pointer :: h
    call c_f_pointer(ctx, h)
    d = h%fn(x(1:n), y(1:n))
  end function

  subroutine nka_delete(this)
    type(nka), intent(inout) :: this
    call nka_delete_handle(this)
  end subroutine

  !! call a%set_vec_tol(vtol)                                 F08:202-207
  subroutine set_vec_tol(this, vtol)
    class(nka), intent(inout) :: this
    real(r8), intent(in) :: vtol
    call nka_hip_check(nka_hip_set_vec_tol(this%handle, vtol), 'nka%set_vec_tol')
  end subroutine

  !! Replaces set_dot_prod (F08:209-214).  fn is a bind(C) function
  !!   integer(c_int) function fn(ctx, buf, count, stream)
  !! that sums `count` doubles at device address buf over all ranks, in place,
  !! ordered on the given hipStream_t.
  subroutine set_allreduce(this, fn, ctx)
    class(nka), intent(inout) :: this
    type(c_funptr), intent(in) :: fn
    type(c_ptr), intent(in) :: ctx
    call nka_hip_check(nka_hip_set_allreduce(this%handle, fn, ctx), 'nka%set_allreduce')
  end subroutine

  !! Built-in hook: the RCCL all-reduces of an update over xGMI (two small ones with the default sums, one in the fast mode).
  subroutine use_rccl(this, id128, nranks, rank)
    class(nka), intent(inout) :: this
    character(kind=c_char), intent(in) :: id128(128)
    integer, intent(in) :: nranks, rank
    call nka_hip_check(nka_hip_comm_init_rank(this%handle, id128, int(nranks, c_int32_t), &
                                              int(rank, c_int32_t)), 'nka%use_rccl')
  end subroutine

  !! Opt-in: the peer-to-peer exchange in place of the all-reduce kernel (include/nka_hip.h, nka_hip_p2p_*).  Collective:
  !! every rank calls p2p_export(nranks, mine), the caller gathers the 64-byte handles in rank order by any means (an
  !! MPI_Allgather of 64 characters), every rank calls p2p_attach(all, nranks, rank) with rank counted from 0.
  subroutine p2p_export(this, nranks, handle64)
    class(nka), intent(inout) :: this
    integer, intent(in) :: nranks
    character(kind=c_char), intent(out) :: handle64(64)
    call nka_hip_check(nka_hip_p2p_export(this%handle, int(nranks, c_int32_t), handle64), 'nka%p2p_export')
  end subroutine

  subroutine p2p_attach(this, handles, nranks, rank)
    class(nka), intent(inout) :: this
    character(kind=c_char), intent(in) :: handles(:)          ! 64*nranks characters
    integer, intent(in) :: nranks, rank
    if (size(handles) < 64*nranks) call nka_hip_check(-1_c_int, 'nka%p2p_attach: need 64*nranks characters')
    call nka_hip_check(nka_hip_p2p_attach(this%handle, handles, int(nranks, c_int32_t), int(rank, c_int32_t)), 'nka%p2p_attach')
  end subroutine

  subroutine p2p_detach(this)
    class(nka), intent(inout) :: this
    call nka_hip_check(nka_hip_p2p_detach(this%handle), 'nka%p2p_detach')
  end subroutine

  integer function vec_len(this)                              ! F08:238-241
    class(nka), intent(in) :: this
    vec_len = int(nka_hip_vec_len(this%handle))
  end function

  integer function num_vec(this)                              ! F08:221-231
    class(nka), intent(in) :: this
    num_vec = nka_hip_num_vec(this%handle)
    if (num_vec < 0) call nka_hip_check(int(num_vec, c_int), 'nka%num_vec')
  end function

  integer function max_vec(this)                              ! F08:233-236
    class(nka), intent(in) :: this
    max_vec = nka_hip_max_vec(this%handle)
  end function

  real(r8) function vec_tol(this)                             ! F08:243-246
    class(nka), intent(in) :: this
    vec_tol = nka_hip_vec_tol(this%handle)
  end function

  !! call a%accel_update(f), host array (reference signature, F08:249-253)
  subroutine accel_update(this, f)
    class(nka), intent(inout) :: this
    real(r8), intent(inout), contiguous :: f(:)
    if (size(f) /= vec_len(this)) error stop 'nka%accel_update: size(f) /= vec_len()'   ! F08:258
    call nka_hip_check(nka_hip_accel_update_host(this%handle, f), 'nka%accel_update')
  end subroutine

  !! The same on device memory: f_dev points to vec_len() doubles on the GPU.
  subroutine accel_update_dev(this, f_dev)
    class(nka), intent(inout) :: this
    type(c_ptr), intent(in) :: f_dev
    call nka_hip_check(nka_hip_accel_update(this%handle, f_dev), 'nka%accel_update_dev')
  end subroutine

  !! Out-of-place form of accel_update_dev (nka_hip_accel_update_swap, include/nka_hip.h): F_IO enters with the
  !! device buffer that holds f -- the accelerator KEEPS it (it becomes the storage of w of the new pair) -- and
  !! returns with a free buffer for the caller's next input; F_ACC is the accelerated f, to be read only, valid
  !! until the next update.  Two of the five store streams of the combine pass less; same bits.
  subroutine accel_update_swap(this, f_io, f_acc)
    class(nka), intent(inout) :: this
    type(c_ptr), intent(inout) :: f_io
    type(c_ptr), intent(out) :: f_acc
    call nka_hip_check(nka_hip_accel_update_swap(this%handle, f_io, f_acc), 'nka%accel_update_swap')
  end subroutine accel_update_swap

  !! Upper bound on the list length (pending pair included) as the host knows it without synchronising: its own
  !! count tightened by the device's list word (nka_hip_list_bound).
  integer function list_bound(this)
    class(nka), intent(in) :: this
    list_bound = nka_hip_list_bound(this%handle)
  end function list_bound

  !! How the inner products are summed (nka_hip_set_sum_order, include/nka_hip.h): NKA_HIP_SUMS_REFERENCE_ORDER = every sum
  !! as the reference forms it (an update then returns the reference's bits at any n; single rank; slow beyond a few
  !! thousand elements), NKA_HIP_SUMS_BLOCKED_ROUNDED = the fast passes with the norm first and the Gram row on the rounded
  !! w1' (the closest to the reference the fast passes get), NKA_HIP_SUMS_AUTO (default) = reference order where it costs
  !! nothing (n <= 64), _BLOCKED_ROUNDED otherwise (since round 6), NKA_HIP_SUMS_BLOCKED = the opt-in single-pass fast mode
  !! (raw-sum Gram row: 5-9 % faster, one exchange per update; also selected by the environment variable NKA_HIP_SUMS=blocked).
  subroutine set_sum_order(this, order)
    class(nka), intent(inout) :: this
    integer, intent(in) :: order
    call nka_hip_check(nka_hip_set_sum_order(this%handle, int(order, c_int32_t)), 'nka%set_sum_order')
  end subroutine set_sum_order

  !! Sharded runs with reference-order sums: slice `rank` (0-based) of `nranks`, slices in rank order (nka_hip_set_shard;
  !! use_rccl tells the handle by itself).  The ranks then continue one another's running sums and the N-rank run returns
  !! the bits of the single-rank reference.
  subroutine set_shard(this, rank, nranks)
    class(nka), intent(inout) :: this
    integer, intent(in) :: rank, nranks
    call nka_hip_check(nka_hip_set_shard(this%handle, int(rank, c_int32_t), int(nranks, c_int32_t)), 'nka%set_shard')
  end subroutine set_shard

  subroutine restart(this)                                    ! F08:422-436
    class(nka), intent(inout) :: this
    call nka_hip_check(nka_hip_restart(this%handle), 'nka%restart')
  end subroutine

  subroutine relax(this)                                      ! F08:439-457
    class(nka), intent(inout) :: this
    call nka_hip_check(nka_hip_relax(this%handle), 'nka%relax')
  end subroutine

  logical function defined(this)                              ! F08:460-524
    class(nka), intent(in) :: this
    defined = .false.
    if (c_associated(this%handle)) defined = (nka_hip_defined(this%handle) == 1)
  end function

  !! which NKA_HIP_FLAVOR_* this object runs (the default resolved)
  integer function get_flavor(this)
    class(nka), intent(in) :: this
    get_flavor = nka_hip_flavor(this%handle)
  end function

  subroutine set_timing(this, capacity)
    class(nka), intent(inout) :: this
    integer, intent(in) :: capacity
    call nka_hip_check(nka_hip_set_timing(this%handle, int(capacity, c_int32_t)), 'nka%set_timing')
  end subroutine

  subroutine get_timing(this, back, ms)
    class(nka), intent(inout) :: this
    integer, intent(in) :: back
    real, intent(out) :: ms(4)
    real(c_float) :: t(4)
    call nka_hip_check(nka_hip_get_timing(this%handle, int(back, c_int32_t), t), 'nka%get_timing')
    ms = t
  end subroutine

end module nka_type
