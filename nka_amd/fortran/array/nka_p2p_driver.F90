!! nka_p2p_driver -- a SHARDED caller of the array flavour written the reference's way (module nka_type, type(nka),
!! accel_update on a host array: /root/reference/src-F08/nka_type.F90:58-64 "every process makes the same calls on its part of
!! the vector") that reduces through the opt-in PEER-TO-PEER EXCHANGE (include/nka_hip.h: nka_hip_p2p_*) instead of an
!! all-reduce: one process per rank,
!!
!!   nka_p2p_driver N MVEC NCALLS OUTFILE RANK WORLD SHMFILE
!!
!! The 64-byte hipIpc handles are gathered through tests/c/shm_allreduce.c (a file-mapped all-gather: this image has no MPI;
!! a real caller uses MPI_Allgather).  Inputs: the integer LCG of SURVEY.md 8(c), every 5th call a vector from a 3-dimensional
!! pool (dependence drops), relax() after call 7.  Written: lo, hi, then per call the global input, num_vec, the local result.

program nka_p2p_driver

  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use, intrinsic :: iso_c_binding
  use nka_type
  implicit none

  interface
    function shm_ar_open(path, world, rank) bind(C) result(ctx)
      import :: c_char, c_int, c_ptr
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), value :: world, rank
      type(c_ptr) :: ctx
    end function
    integer(c_int) function shm_allgather(ctx, mine, nbytes, all) bind(C)
      import :: c_int, c_int32_t, c_ptr, c_char
      type(c_ptr), value :: ctx
      character(kind=c_char), intent(in) :: mine(*)
      integer(c_int32_t), value :: nbytes
      character(kind=c_char), intent(out) :: all(*)
    end function
    integer(c_int) function shm_barrier(ctx) bind(C)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
  end interface

  character(256) :: arg, outfile, shmfile
  integer :: n, mvec, ncalls, rank, world, t, k, i, lun, lo, hi
  integer(i8) :: lcg_state = 1
  type(nka) :: accel
  type(c_ptr) :: shm
  character(kind=c_char) :: mine(64)
  character(kind=c_char), allocatable :: handles(:)
  real(r8), allocatable :: host(:), pool(:,:), coef(:), f(:)

  call get_command_argument(1, arg); read(arg,*) n
  call get_command_argument(2, arg); read(arg,*) mvec
  call get_command_argument(3, arg); read(arg,*) ncalls
  call get_command_argument(4, outfile)
  call get_command_argument(5, arg); read(arg,*) rank
  call get_command_argument(6, arg); read(arg,*) world
  call get_command_argument(7, shmfile)

  lo = int((int(rank, i8) * n) / world)             ! contiguous slices (nka_amd/dist.py:slice_bounds)
  hi = int((int(rank + 1, i8) * n) / world)
  if (rank > 0) lo = lo - mod(lo, 2)
  if (rank + 1 < world) hi = hi - mod(hi, 2)
  shm = shm_ar_open(trim(shmfile)//c_null_char, int(world, c_int), int(rank, c_int))
  if (.not. c_associated(shm)) error stop 'cannot map the exchange file'

  call accel%init(hi - lo, mvec)
  ! the peer-to-peer exchange, collectively: export, gather the handles in rank order, attach
  allocate(handles(64*world))
  call accel%p2p_export(world, mine)
  if (shm_allgather(shm, mine, 64_c_int32_t, handles) /= 0) error stop 'gathering the handles failed'
  call accel%p2p_attach(handles, world, rank)

  allocate(host(n), pool(n,3), coef(3), f(hi - lo))
  do k = 1, 3
    do i = 1, n
      pool(i,k) = lcg()
    end do
  end do
  open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
  write(lun) int(lo, i8), int(hi, i8)
  do t = 1, ncalls
    if (mod(t, 5) == 0) then
      do k = 1, 3
        coef(k) = lcg()
      end do
      host = coef(1)*pool(:,1) + coef(2)*pool(:,2) + coef(3)*pool(:,3)
    else
      do i = 1, n
        host(i) = lcg()
      end do
    end if
    write(lun) host
    f = host(lo+1:hi)
    call accel%accel_update(f)                      ! F08:249 -- host array in, accelerated correction out
    if (t == 7) call accel%relax
    write(lun) real(accel%num_vec(), r8)
    write(lun) f
  end do
  close(lun)
  if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
  if (shm_barrier(shm) /= 0) error stop 'a peer is missing at the end'    ! (nobody frees a mailbox a peer may still write into)
  call accel%p2p_detach
  write(*,'(a,i0,a,i0,a,i0)') 'nka_p2p_driver: rank ', rank, ' of ', world, ', final num_vec ', accel%num_vec()

contains

  real(r8) function lcg()
    lcg_state = mod(1103515245_i8*lcg_state + 12345_i8, 2147483648_i8)
    lcg = real(lcg_state, r8) / 1073741824.0_r8 - 1.0_r8
  end function

end program nka_p2p_driver
