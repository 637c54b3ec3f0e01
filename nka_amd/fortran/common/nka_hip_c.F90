!! nka_hip_c -- iso_c_binding interfaces to the C ABI of libnka_hip.so
!! (include/nka_hip.h).  The thin shim the Fortran host code calls the HIP
!! kernels through; no arithmetic happens on this side.

module nka_hip_c

  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: NKA_HIP_FLAVOR_F08 = 0, NKA_HIP_FLAVOR_F08_VECTOR = 1, NKA_HIP_FLAVOR_C = 2
  !! resolved by nka_hip_create: environment NKA_HIP_FLAVOR, else compact storage (include/nka_hip.h)
  integer(c_int), parameter :: NKA_HIP_FLAVOR_DEFAULT = -1
  integer(c_int), parameter :: NKA_HIP_SUMS_AUTO = 0, NKA_HIP_SUMS_REFERENCE_ORDER = 1, NKA_HIP_SUMS_BLOCKED = 2, &
                               NKA_HIP_SUMS_BLOCKED_ROUNDED = 3

  interface
    integer(c_int) function nka_hip_create(handle, vlen_local, mvec, vtol, flavor, device, stream) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
      type(c_ptr), intent(out) :: handle
      integer(c_int64_t), value :: vlen_local
      integer(c_int32_t), value :: mvec, flavor, device
      real(c_double), value :: vtol
      type(c_ptr), value :: stream
    end function
    integer(c_int) function nka_hip_destroy(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_clone(src, out) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: src
      type(c_ptr), intent(out) :: out
    end function
    integer(c_int) function nka_hip_accel_update(handle, f_dev) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, f_dev
    end function
    integer(c_int) function nka_hip_accel_update_swap(handle, f_io, f_acc) bind(C)   ! out of place: include/nka_hip.h
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      type(c_ptr), intent(inout) :: f_io      ! in: the caller's buffer with f (kept by the library); out: a free buffer
      type(c_ptr), intent(out) :: f_acc       ! the accelerated f, to be read only, valid until the next update
    end function
    integer(c_int) function nka_hip_list_bound(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_vec_set_sum_order(ws, order) bind(C)
      import :: c_int, c_ptr, c_int32_t
      type(c_ptr), value :: ws
      integer(c_int32_t), value :: order
    end function
    integer(c_int) function nka_hip_vec_get_sum_order(ws) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: ws
    end function
    integer(c_int) function nka_hip_set_sum_order(handle, order) bind(C)
      import :: c_int, c_ptr, c_int32_t
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: order      ! NKA_HIP_SUMS_AUTO / _REFERENCE_ORDER / _BLOCKED (include/nka_hip.h)
    end function
    integer(c_int) function nka_hip_set_shard(handle, rank, nranks) bind(C)
      import :: c_int, c_ptr, c_int32_t
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: rank, nranks   ! slice `rank` (0-based) of `nranks`, slices in rank order
    end function
    integer(c_int) function nka_hip_accel_update_host(handle, f_host) bind(C)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(inout) :: f_host(*)
    end function
    integer(c_int) function nka_hip_restart(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_relax(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_set_vec_tol(handle, vtol) bind(C)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), value :: vtol
    end function
    integer(c_int) function nka_hip_num_vec(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_max_vec(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_flavor(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int64_t) function nka_hip_vec_len(handle) bind(C)
      import :: c_int64_t, c_ptr
      type(c_ptr), value :: handle
    end function
    real(c_double) function nka_hip_vec_tol(handle) bind(C)
      import :: c_double, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_defined(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_set_allreduce(handle, fn, ctx) bind(C)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: handle, ctx
      type(c_funptr), value :: fn
    end function
    integer(c_int) function nka_hip_set_host_dot(handle, fn, ctx) bind(C)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: handle, ctx
      type(c_funptr), value :: fn
    end function
    integer(c_int) function nka_hip_state_digest(handle, digest) bind(C)
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: handle
      integer(c_int64_t), intent(out) :: digest
    end function
    integer(c_int) function nka_hip_comm_destroy(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_comm_unique_id(id128) bind(C)
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: id128(128)
    end function
    integer(c_int) function nka_hip_comm_init_rank(handle, id128, nranks, rank) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: id128(128)
      integer(c_int32_t), value :: nranks, rank
    end function
    ! peer-to-peer exchange (opt-in, include/nka_hip.h): handle64 = a hipIpcMemHandle_t; handles = nranks of them in rank order
    integer(c_int) function nka_hip_p2p_export(handle, nranks, handle64) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_char
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nranks
      character(kind=c_char), intent(out) :: handle64(64)
    end function
    integer(c_int) function nka_hip_p2p_attach(handle, handles, nranks, rank) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: handles(*)
      integer(c_int32_t), value :: nranks, rank
    end function
    integer(c_int) function nka_hip_p2p_detach(handle) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function nka_hip_set_timing(handle, capacity) bind(C)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: capacity
    end function
    integer(c_int) function nka_hip_get_timing(handle, back, ms) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_float
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: back
      real(c_float), intent(out) :: ms(4)
    end function
    type(c_ptr) function nka_hip_last_error() bind(C)
      import :: c_ptr
    end function

    !! vector primitives (the deferred procedures of the abstract vector class)
    integer(c_int) function nka_hip_vec_workspace_create(ws, device, stream) bind(C)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), intent(out) :: ws
      integer(c_int32_t), value :: device
      type(c_ptr), value :: stream
    end function
    integer(c_int) function nka_hip_vec_workspace_destroy(ws) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: ws
    end function
    !! parallel-aware reductions of the device vector types (include/nka_hip.h)
    integer(c_int) function nka_hip_vec_set_allreduce(ws, fn, ctx) bind(C)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: ws, ctx
      type(c_funptr), value :: fn
    end function
    integer(c_int) function nka_hip_vec_set_host_allreduce(ws, fn, ctx) bind(C)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: ws, ctx
      type(c_funptr), value :: fn
    end function
    integer(c_int) function nka_hip_vec_comm_init_rank(ws, id128, nranks, rank) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_char
      type(c_ptr), value :: ws
      character(kind=c_char), intent(in) :: id128(128)
      integer(c_int32_t), value :: nranks, rank
    end function
    integer(c_int) function nka_hip_vec_comm_destroy(ws) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: ws
    end function
    integer(c_int) function nka_hip_vec_allreduce_now(ws, host_vals, count) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: ws
      real(c_double), intent(inout) :: host_vals(*)
      integer(c_int32_t), value :: count
    end function
    integer(c_int) function nka_hip_vec_alloc(ws, n, dev) bind(C)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: ws
      integer(c_int64_t), value :: n
      type(c_ptr), intent(out) :: dev
    end function
    integer(c_int) function nka_hip_vec_free(ws, dev) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: ws, dev
    end function
    integer(c_int) function nka_hip_vec_copy(ws, n, dst, src) bind(C)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: ws, dst, src
      integer(c_int64_t), value :: n
    end function
    integer(c_int) function nka_hip_vec_setval(ws, n, x, val) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, x
      integer(c_int64_t), value :: n
      real(c_double), value :: val
    end function
    integer(c_int) function nka_hip_vec_scale(ws, n, x, a) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, x
      integer(c_int64_t), value :: n
      real(c_double), value :: a
    end function
    integer(c_int) function nka_hip_vec_update1(ws, n, z, a, x) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, x
      integer(c_int64_t), value :: n
      real(c_double), value :: a
    end function
    integer(c_int) function nka_hip_vec_update2(ws, n, z, a, x, b) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, x
      integer(c_int64_t), value :: n
      real(c_double), value :: a, b
    end function
    integer(c_int) function nka_hip_vec_update3(ws, n, z, a, x, b, y) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, x, y
      integer(c_int64_t), value :: n
      real(c_double), value :: a, b
    end function
    integer(c_int) function nka_hip_vec_update4(ws, n, z, a, x, b, y, c) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, x, y
      integer(c_int64_t), value :: n
      real(c_double), value :: a, b, c
    end function
    integer(c_int) function nka_hip_vec_dot(ws, n, x, y, res) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, x, y
      integer(c_int64_t), value :: n
      real(c_double), intent(out) :: res
    end function
    integer(c_int) function nka_hip_vec_norm2(ws, n, x, res) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, x
      integer(c_int64_t), value :: n
      real(c_double), intent(out) :: res
    end function
    integer(c_int) function nka_hip_vec_dot_many(ws, n, x, ys, count, vals) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, x
      integer(c_int64_t), value :: n
      type(c_ptr), intent(in) :: ys(*)            ! host array of device pointers
      integer(c_int32_t), value :: count
      real(c_double), intent(out) :: vals(*)
    end function
    integer(c_int) function nka_hip_vec_dot_pair_many(ws, n, x0, x1, ys, count, vals0, vals1, cross) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, x0, x1
      integer(c_int64_t), value :: n
      type(c_ptr), intent(in) :: ys(*)
      integer(c_int32_t), value :: count
      real(c_double), intent(out) :: vals0(*), vals1(*), cross
    end function
    integer(c_int) function nka_hip_vec_update_many(ws, n, z, a, xs, b, ys, count) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: a(*), b(*)
      type(c_ptr), intent(in) :: xs(*), ys(*)     ! host arrays of device pointers
      integer(c_int32_t), value :: count
    end function
    integer(c_int) function nka_hip_vec_axpy_many(ws, n, z, a, xs, count) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: a(*)
      type(c_ptr), intent(in) :: xs(*)
      integer(c_int32_t), value :: count
    end function
    integer(c_int) function nka_hip_vec_update_norm2(ws, n, z, a, x, store, res) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, x
      integer(c_int64_t), value :: n
      real(c_double), value :: a
      integer(c_int32_t), value :: store
      real(c_double), intent(out) :: res
    end function
    integer(c_int) function nka_hip_vec_scale_dot_pair_many(ws, n, w, v, a, subtract, pre, pre_a, f, ys, count, &
                                                            vals_w, vals_f, cross) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, w, v, f
      integer(c_int64_t), value :: n
      real(c_double), value :: a, pre_a
      integer(c_int32_t), value :: subtract, pre, count
      type(c_ptr), intent(in) :: ys(*)
      real(c_double), intent(out) :: vals_w(*), vals_f(*), cross
    end function
    integer(c_int) function nka_hip_vec_update_many_keep(ws, n, z, a, xs, b, ys, count, keep_in, keep_out) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, keep_in, keep_out
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: a(*), b(*)
      type(c_ptr), intent(in) :: xs(*), ys(*)
      integer(c_int32_t), value :: count
    end function
    integer(c_int) function nka_hip_vec_axpy_many_keep(ws, n, z, a, xs, count, keep_in, keep_out) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, keep_in, keep_out
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: a(*)
      type(c_ptr), intent(in) :: xs(*)
      integer(c_int32_t), value :: count
    end function
    integer(c_int) function nka_hip_vec_dot_pair_many_scaled(ws, n, w, a, pre, pre_a, f, ys, count, &
                                                             vals_w, vals_f, cross) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, w, f
      integer(c_int64_t), value :: n
      real(c_double), value :: a, pre_a
      integer(c_int32_t), value :: pre, count
      type(c_ptr), intent(in) :: ys(*)
      real(c_double), intent(out) :: vals_w(*), vals_f(*), cross
    end function
    integer(c_int) function nka_hip_vec_diff_norm_dot_pair_many(ws, n, z, a, x, ys, count, dd, vals_z, vals_x, cross) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, x
      integer(c_int64_t), value :: n
      real(c_double), value :: a
      integer(c_int32_t), value :: count
      type(c_ptr), intent(in) :: ys(*)
      real(c_double), intent(out) :: dd, vals_z(*), vals_x(*), cross
    end function
    integer(c_int) function nka_hip_vec_update_many_keep_pend(ws, n, z, a, xs, b, ys, count, keep_in, keep_out, &
                                                              pend_a, pend_pre, pend_pre_a, pend_subtract) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, keep_in, keep_out
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: a(*), b(*)
      type(c_ptr), intent(in) :: xs(*), ys(*)
      integer(c_int32_t), value :: count, pend_pre, pend_subtract
      real(c_double), value :: pend_a, pend_pre_a
    end function
    integer(c_int) function nka_hip_vec_axpy_many_keep_pend(ws, n, z, a, xs, count, keep_in, keep_out, &
                                                            pend_w, pend_a, pend_pre, pend_pre_a) bind(C)
      import :: c_int, c_int32_t, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, z, keep_in, keep_out, pend_w
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: a(*)
      type(c_ptr), intent(in) :: xs(*)
      integer(c_int32_t), value :: count, pend_pre
      real(c_double), value :: pend_a, pend_pre_a
    end function
    integer(c_int) function nka_hip_vec_h2d(ws, n, dst_dev, src_host) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, dst_dev
      integer(c_int64_t), value :: n
      real(c_double), intent(in) :: src_host(*)
    end function
    integer(c_int) function nka_hip_vec_d2h(ws, n, dst_host, src_dev) bind(C)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: ws, src_dev
      integer(c_int64_t), value :: n
      real(c_double), intent(out) :: dst_host(*)
    end function
  end interface

contains

  !! The reference has no status codes: a failed precondition stops the program
  !! (f90_assert / error stop).  A failing C-ABI call does the same here, with
  !! the library's message.
  subroutine nka_hip_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    write(*,'(3a,i0)') 'nka_hip: ', what, ' failed with status ', rc
    p = nka_hip_last_error()
    if (c_associated(p)) then
      call c_f_pointer(p, msg, [512])
      do i = 1, 512
        if (msg(i) == c_null_char) exit
      end do
      write(*,'(512a)') msg(1:i-1)
    end if
    error stop 1
  end subroutine

end module nka_hip_c
