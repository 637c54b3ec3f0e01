!! nka_bench -- steady-state accel_update throughput through the FORTRAN front end
!! (module nka_type -> iso_c_binding -> libnka_hip.so) on device-resident vectors.
!!
!!   nka_bench [N [MVEC [STEPS [FLAVOR]]]]      defaults 100000000 20 10 2 (C rounding)
!!
!! Inputs are independent uniform(-1,1) vectors generated on the host and copied
!! to HBM BEFORE the timed region; the timed region is STEPS calls of
!! accel%accel_update_dev(f_dev) bracketed by stream synchronisations
!! (num_vec() synchronises).  Reports updates/s and the contract's algorithmic
!! bandwidth 8n(11+3m)/t (SURVEY.md 8d).  bench.py is the driver's benchmark;
!! this program shows the same numbers are reached from Fortran host code.

program nka_bench

  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use, intrinsic :: iso_c_binding
  use nka_hip_c
  use nka_type
  implicit none

  integer(i8) :: n = 100000000_i8
  integer :: mvec = 20, steps = 10, flavor = NKA_HIP_FLAVOR_C
  character(64) :: arg
  type(nka) :: accel
  type(c_ptr) :: ws
  type(c_ptr), allocatable :: pool(:)
  real(r8), allocatable :: host(:)
  integer :: t, warm, ninp, nv
  integer(i8) :: c0, c1, rate
  real(r8) :: per

  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg,*) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg,*) mvec; end if
  if (command_argument_count() >= 3) then; call get_command_argument(3, arg); read(arg,*) steps; end if
  if (command_argument_count() >= 4) then; call get_command_argument(4, arg); read(arg,*) flavor; end if
  if (n > huge(1)) error stop 'nka_bench: the Fortran init takes a default integer length'

  warm = mvec + 3
  ninp = warm + steps
  call nka_hip_check(nka_hip_vec_workspace_create(ws, 0_c_int32_t, c_null_ptr), 'vec_workspace_create')
  call accel%init(int(n), mvec, flavor=flavor)
  allocate(pool(ninp), host(n))
  do t = 1, ninp
    call nka_hip_check(nka_hip_vec_alloc(ws, n, pool(t)), 'vec_alloc')
    call random_number(host)
    host = 2.0_r8*host - 1.0_r8
    call nka_hip_check(nka_hip_vec_h2d(ws, n, pool(t), host), 'vec_h2d')
  end do
  deallocate(host)

  do t = 1, warm
    call accel%accel_update_dev(pool(t))
  end do
  nv = accel%num_vec()                   ! synchronises
  if (nv /= mvec) error stop 'nka_bench: subspace not full after warm-up'

  call system_clock(c0, rate)
  do t = warm+1, ninp
    call accel%accel_update_dev(pool(t))
  end do
  nv = accel%num_vec()                   ! synchronises
  call system_clock(c1)
  per = real(c1 - c0, r8) / real(rate, r8) / steps

  write(*,'(a,i0,a,i0,a,i0,a,i0)') 'nka_bench (Fortran front end): n=', n, ' mvec=', mvec, ' steps=', steps, ' flavor=', flavor
  write(*,'(a,f10.3,a,f9.4)') 'updates/s ', 1.0_r8/per, '   ms/update ', 1e3_r8*per
  write(*,'(a,f10.1,a,f7.4)') 'algorithmic GB/s (8n(11+3m)/t) ', 8.0_r8*n*(11+3*mvec)/per/1e9_r8, &
                              '   fraction of 8 TB/s ', 8.0_r8*n*(11+3*mvec)/per/8.0e12_r8
  if (.not. accel%defined()) error stop 'accelerator not well defined'
end program nka_bench
