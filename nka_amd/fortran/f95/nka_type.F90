!! nka_type (Fortran 95 procedural flavour) -- SURVEY.md 8 row f3.
!!
!! The reference's original procedural API (src-F95/nka_type.F90:205-207):
!!   nka_init(this,vlen,mvec)  nka_delete(this)  nka_set_vec_tol(this,vtol)
!!   nka_accel_update(this,f)  nka_relax(this)   nka_restart(this)
!!   nka_num_vec  nka_max_vec  nka_vec_len  nka_vec_tol  nka_real_kind  nka_defined
!! as thin wrappers over the C ABI of libnka_hip.so (module nka_hip_c).  The type
!! is a handle; every vector and the scalar step live on the GPU.
!!
!! The reference's optional per-call dummy procedure DP (src-F95/nka_type.F90:
!! 284-291) is kept with its interface: when present, the inner products of THAT
!! call are evaluated by DP on host copies of the operands, in the reference's
!! own order (slow compatibility path, nka_hip_set_host_dot in include/nka_hip.h).
!! The fast distribution hook is nka_set_allreduce (global sum of the device
!! partial sums).  nka_accel_update_dev takes device memory (type(c_ptr)).

module nka_type

  use, intrinsic :: iso_c_binding
  use nka_hip_c
  implicit none
  private

  integer, parameter :: r8 = selected_real_kind(15)

  type, public :: nka
    private
    type(c_ptr) :: handle = c_null_ptr
  end type nka

  !! the dummy DP of the nka_accel_update call in progress (the object is not
  !! thread safe in the reference either, SURVEY.md 8b)
  abstract interface
    pure function dp_iface(x, y)
      integer, parameter :: r8 = selected_real_kind(15)
      real(r8), intent(in) :: x(:), y(:)
      real(r8) :: dp_iface
    end function
  end interface
  procedure(dp_iface), pointer, save :: call_dp => null()

  public :: nka_init, nka_delete, nka_set_vec_tol, nka_defined
  public :: nka_vec_len, nka_num_vec, nka_max_vec, nka_vec_tol, nka_real_kind
  public :: nka_accel_update, nka_accel_update_dev, nka_relax, nka_restart, nka_set_allreduce
  public :: nka_set_sum_order, NKA_HIP_SUMS_AUTO, NKA_HIP_SUMS_REFERENCE_ORDER, NKA_HIP_SUMS_BLOCKED

contains

  subroutine nka_init(this, vlen, mvec)                       ! src-F95 :211-225
    type(nka), intent(inout) :: this
    integer, intent(in) :: vlen, mvec
    call nka_delete(this)
    call nka_hip_check(nka_hip_create(this%handle, int(vlen, c_int64_t), int(mvec, c_int32_t), 0.01_c_double, &
                                      NKA_HIP_FLAVOR_DEFAULT, 0_c_int32_t, c_null_ptr), 'nka_init')
  end subroutine

  subroutine nka_delete(this)                                 ! :266-275
    type(nka), intent(inout) :: this
    integer(c_int) :: rc
    if (c_associated(this%handle)) rc = nka_hip_destroy(this%handle)
    this%handle = c_null_ptr
  end subroutine

  subroutine nka_set_vec_tol(this, vtol)                      ! :227-232
    type(nka), intent(inout) :: this
    real(r8), intent(in) :: vtol
    call nka_hip_check(nka_hip_set_vec_tol(this%handle, vtol), 'nka_set_vec_tol')
  end subroutine

  !! Beside the reference's API: how the inner products are summed (nka_hip_set_sum_order, include/nka_hip.h).
  !! NKA_HIP_SUMS_REFERENCE_ORDER: every sum as the reference forms it -- an update then returns the reference's bits at any n.
  subroutine nka_set_sum_order(this, order)
    type(nka), intent(inout) :: this
    integer, intent(in) :: order
    call nka_hip_check(nka_hip_set_sum_order(this%handle, int(order, c_int32_t)), 'nka_set_sum_order')
  end subroutine

  subroutine nka_set_allreduce(this, fn, ctx)
    type(nka), intent(inout) :: this
    type(c_funptr), intent(in) :: fn
    type(c_ptr), intent(in) :: ctx
    call nka_hip_check(nka_hip_set_allreduce(this%handle, fn, ctx), 'nka_set_allreduce')
  end subroutine

  subroutine nka_accel_update(this, f, dp)                    ! :278-473 (host array)
    type(nka), intent(inout) :: this
    real(r8), intent(inout) :: f(:)
    !! Optional dot product to use instead of the device sums (:284-291).
    interface
      pure function dp(x, y)
        integer, parameter :: r8 = selected_real_kind(15)
        real(r8), intent(in) :: x(:), y(:)
        real(r8) :: dp
      end function dp
    end interface
    optional :: dp
    real(r8), allocatable :: tmp(:)
    integer(c_int) :: rc
    if (size(f) /= nka_vec_len(this)) stop 'nka_accel_update: size(f) /= nka_vec_len(this)'
    if (present(dp)) then
      call_dp => dp
      call nka_hip_check(nka_hip_set_host_dot(this%handle, c_funloc(per_call_dot), c_null_ptr), 'nka_accel_update')
    end if
    if (is_contiguous(f)) then
      rc = nka_hip_accel_update_host(this%handle, f)
    else
      tmp = f
      rc = nka_hip_accel_update_host(this%handle, tmp)
      f = tmp
    end if
    if (present(dp)) then
      call nka_hip_check(nka_hip_set_host_dot(this%handle, c_null_funptr, c_null_ptr), 'nka_accel_update')
      call_dp => null()
    end if
    call nka_hip_check(rc, 'nka_accel_update')
  end subroutine

  !! nka_hip_host_dot_fn for the duration of ONE nka_accel_update call
  function per_call_dot(ctx, n, x, y) bind(C) result(d)
    type(c_ptr), value :: ctx
    integer(c_int64_t), value :: n
    real(c_double), intent(in) :: x(*), y(*)
    real(c_double) :: d
    d = call_dp(x(1:n), y(1:n))
  end function

  subroutine nka_accel_update_dev(this, f_dev)
    type(nka), intent(inout) :: this
    type(c_ptr), intent(in) :: f_dev
    call nka_hip_check(nka_hip_accel_update(this%handle, f_dev), 'nka_accel_update_dev')
  end subroutine

  subroutine nka_relax(this)                                  ! :494-512
    type(nka), intent(inout) :: this
    call nka_hip_check(nka_hip_relax(this%handle), 'nka_relax')
  end subroutine

  subroutine nka_restart(this)                                ! :476-491
    type(nka), intent(inout) :: this
    call nka_hip_check(nka_hip_restart(this%handle), 'nka_restart')
  end subroutine

  integer function nka_num_vec(this)
    type(nka), intent(in) :: this
    nka_num_vec = nka_hip_num_vec(this%handle)
  end function

  integer function nka_max_vec(this)
    type(nka), intent(in) :: this
    nka_max_vec = nka_hip_max_vec(this%handle)
  end function

  integer function nka_vec_len(this)
    type(nka), intent(in) :: this
    nka_vec_len = int(nka_hip_vec_len(this%handle))
  end function

  real(r8) function nka_vec_tol(this)
    type(nka), intent(in) :: this
    nka_vec_tol = nka_hip_vec_tol(this%handle)
  end function

  integer function nka_real_kind(this)                        ! :261-264
    type(nka), intent(in) :: this
    nka_real_kind = r8
  end function

  logical function nka_defined(this)
    type(nka), intent(in) :: this
    nka_defined = .false.
    if (c_associated(this%handle)) nka_defined = (nka_hip_defined(this%handle) == 1)
  end function

end module nka_type
