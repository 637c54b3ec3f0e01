!! nka_vector_driver -- exercises the abstract-vector flavour on the GPU.
!!
!!   nka_vector_driver check NFIELD NPER MVEC NCALLS OUTFILE [COMPACT 0|1]
!!       drives NKA (vector flavour) on a hip_block_vector with the integer-LCG
!!       inputs of SURVEY.md 8(c) (x <- (1103515245 x + 12345) mod 2^31, value
!!       x/2^30 - 1; every 5th call a vector from a 3-dimensional pool so that
!!       dependence drops occur) and writes, per call, num_vec and the returned
!!       vector to OUTFILE (stream, native real64) for tests/test_fortran_gpu.py
!!       to compare with the oracle's F08-vector flavour.
!!   nka_vector_driver checktile NFIELD NPER MVEC NCALLS OUTFILE COMPACT R
!!       the same at BASELINE size without a BASELINE-size oracle: the small LCG
!!       input x of length n0 = NFIELD*NPER is tiled R times into a block vector
!!       of NFIELD fields of NPER*R (X(i) = x(mod(i,n0))).  Every inner product is
!!       R times the small one, so with R = 4^k the returned vector is the tiled
!!       small result up to the rounding of the sums (tests/test_hip_fullsize.py).
!!       Written per call: x, num_vec, the FIRST tile of the result, and the
!!       2-norm of (result - tiled first tile), which must be exactly zero.
!!   nka_vector_driver checkgrid NX NY MVEC NCALLS OUTFILE [COMPACT 0|1]
!!       the same on a hip_grid_vector (NX x NY cells plus a ghost ring, the
!!       device counterpart of the reference's grid_vector): every value of the
!!       (NX+2) x (NY+2) array, ghosts included, comes from the LCG, so a
!!       reduction that saw a ghost would change every decision.  Written per call:
!!       the input array, num_vec, the returned array (natural layout, ghosts
!!       included) -- compared with the compiled reference on its own grid_vector.
!!   nka_vector_driver shard NFIELD NPER MVEC NCALLS OUTFILE COMPACT RANK WORLD SHMFILE [RCCL 0|1]
!!       the `check` run SHARDED over WORLD processes (SURVEY.md 8e; the reference's
!!       contract for this flavour: parallel-aware reductions in the vector class,
!!       src-F08-vector/README.md:16-22): rank RANK holds a contiguous slice of each
!!       of the NFIELD fields of (global) length NPER in a hip_block_vector whose
!!       workspace carries a HOST all-reduce hook (tests/c/shm_allreduce.c through a
!!       file mapped by all ranks -- where a real caller installs MPI_Allreduce) and,
!!       with RCCL 1, the built-in RCCL hook on a one-rank communicator in front of
!!       it (the device-side path).  Written per call: the global input, num_vec, the
!!       digest of this rank's replicated scalar state, this rank's slices of the result.
!!   nka_vector_driver script NFIELD NPER MVEC NOPS OUTFILE COMPACT SCRIPTFILE [RANK WORLD SHMFILE]
!!       replays NOPS operations from SCRIPTFILE (stream of real64: an operation code, then
!!       its payload -- 0 accel_update + the NFIELD*NPER input values, 1 relax, 2 restart,
!!       3 set_vec_tol + the tolerance) on a hip_block_vector and writes, per operation,
!!       num_vec and, after an update, the returned vector: random call sequences generated
!!       and checked against the oracle by tools/fuzz_gpu.py --vector.  With RANK WORLD SHMFILE the replay is
!!       SHARDED like `shard`: this rank holds the slice [lo, hi) of each field (the script carries the GLOBAL
!!       inputs), the workspace reduces through the host all-reduce of tests/c/shm_allreduce.c; written first:
!!       lo, hi; per update: this rank's slices of the result (tools/fuzz_gpu.py --vector-sharded).
!!   nka_vector_driver bench NFIELD NPER MVEC STEPS [COMPACT 0|1]
!!       BASELINE config 5 (4 x 1e7, mvec 20): steady-state updates/s of the
!!       hook-by-hook path, with the bytes it moves, 8n(12+8m) (SURVEY.md 8d).
!!   nka_vector_driver benchgrid NX NY MVEC STEPS [COMPACT 0|1]
!!       the same on a hip_grid_vector of NX x NY cells plus its ghost ring.

program nka_vector_driver

  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use, intrinsic :: iso_c_binding
  use vector_class
  use nka_hip_c, only: nka_hip_check, nka_hip_comm_unique_id
  use hip_block_vector_type
  use hip_grid_vector_type
  use nka_type
  implicit none

  character(256) :: mode, arg, outfile, shmfile, scriptfile
  integer :: nfield, mvec, ncalls, icompact = 0, rtile = 1, rank = 0, world = 1, irccl = 0
  logical :: compact, grid = .false., refsum = .false.   ! (compact argument + 10: sums in the reference's order)
  logical :: rounded = .false.                           ! (compact argument + 20: NKA_HIP_SUMS_BLOCKED_ROUNDED)
  integer(i8) :: nper
  integer(i8) :: lcg_state = 1

  !! tests/c/shm_allreduce.c (test infrastructure: a host all-reduce without MPI)
  interface
    function shm_ar_open(path, world, rank) bind(C) result(ctx)
      import :: c_char, c_int, c_ptr
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), value :: world, rank
      type(c_ptr) :: ctx
    end function
    function shm_allreduce(ctx, vals, count) bind(C) result(rc)
      import :: c_ptr, c_double, c_int32_t, c_int
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: vals(*)
      integer(c_int32_t), value :: count
      integer(c_int) :: rc
    end function
  end interface

  call get_command_argument(1, mode)
  call get_command_argument(2, arg); read(arg,*) nfield
  call get_command_argument(3, arg); read(arg,*) nper
  call get_command_argument(4, arg); read(arg,*) mvec
  call get_command_argument(5, arg); read(arg,*) ncalls
  select case (trim(mode))
  case ('check')
    call get_command_argument(6, outfile)
    if (command_argument_count() >= 7) then
      call get_command_argument(7, arg); read(arg,*) icompact
    end if
    refsum = icompact >= 10 .and. icompact < 20; rounded = icompact >= 20; compact = mod(icompact, 10) /= 0
    call run_check
  case ('script')
    call get_command_argument(6, outfile)
    call get_command_argument(7, arg); read(arg,*) icompact
    call get_command_argument(8, scriptfile)
    if (command_argument_count() >= 11) then
      call get_command_argument(9, arg); read(arg,*) rank
      call get_command_argument(10, arg); read(arg,*) world
      call get_command_argument(11, shmfile)
    end if
    refsum = icompact >= 10 .and. icompact < 20; rounded = icompact >= 20; compact = mod(icompact, 10) /= 0
    call run_script
  case ('shard')
    call get_command_argument(6, outfile)
    call get_command_argument(7, arg); read(arg,*) icompact
    call get_command_argument(8, arg); read(arg,*) rank
    call get_command_argument(9, arg); read(arg,*) world
    call get_command_argument(10, shmfile)
    if (command_argument_count() >= 11) then
      call get_command_argument(11, arg); read(arg,*) irccl
    end if
    refsum = icompact >= 10 .and. icompact < 20; rounded = icompact >= 20; compact = mod(icompact, 10) /= 0
    call run_shard
  case ('checkgrid')
    call get_command_argument(6, outfile)
    if (command_argument_count() >= 7) then
      call get_command_argument(7, arg); read(arg,*) icompact
    end if
    refsum = icompact >= 10 .and. icompact < 20; rounded = icompact >= 20; compact = mod(icompact, 10) /= 0
    call run_checkgrid
  case ('checktile')
    call get_command_argument(6, outfile)
    call get_command_argument(7, arg); read(arg,*) icompact
    call get_command_argument(8, arg); read(arg,*) rtile
    refsum = icompact >= 10 .and. icompact < 20; rounded = icompact >= 20; compact = mod(icompact, 10) /= 0
    call run_checktile
  case ('bench', 'benchgrid')
    grid = trim(mode) == 'benchgrid'
    if (command_argument_count() >= 6) then
      call get_command_argument(6, arg); read(arg,*) icompact
    end if
    refsum = icompact >= 10 .and. icompact < 20; rounded = icompact >= 20; compact = mod(icompact, 10) /= 0
    call run_bench
  case default
    error stop 'usage: nka_vector_driver check|bench NFIELD NPER MVEC NCALLS [OUTFILE]'
  end select

contains

  real(r8) function lcg()
    lcg_state = mod(1103515245_i8*lcg_state + 12345_i8, 2147483648_i8)
    lcg = real(lcg_state, r8) / 1073741824.0_r8 - 1.0_r8
  end function

  subroutine run_check
    type(hip_block_vector) :: f
    type(nka) :: accel
    type(c_ptr) :: ws
    real(r8), allocatable :: host(:), pool(:,:), coef(:)
    integer :: t, k, lun
    integer(i8) :: n, i
    n = nfield * nper
    ws = hip_block_vector_workspace(0)
    if (refsum) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER)
    if (rounded) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_BLOCKED_ROUNDED)
    call f%init(nfield, nper, ws)
    call accel%init(f, mvec, compact=compact)
    allocate(host(n), pool(n,3), coef(3))
    do k = 1, 3
      do i = 1, n
        pool(i,k) = lcg()
      end do
    end do
    open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
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
      write(lun) host                             ! the input, so the checker needs no second LCG
      do k = 1, nfield
        call f%set_field(k, host((k-1)*nper+1:k*nper))
      end do
      call accel%accel_update(f)
      do k = 1, nfield
        call f%get_field(k, host((k-1)*nper+1:k*nper))
      end do
      write(lun) real(accel%num_vec(), r8)
      write(lun) host
    end do
    close(lun)
    if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
    write(*,'(a,i0,a,i0)') 'check: wrote ', ncalls, ' calls, final num_vec ', accel%num_vec()
  end subroutine

  subroutine run_script
    type(hip_block_vector) :: f
    type(nka) :: accel
    type(c_ptr) :: ws, shm
    real(r8), allocatable :: host(:), loc(:)
    real(r8) :: code, vtol
    integer :: t, k, lun, lin
    integer(i8) :: n, lo, hi, nloc
    n = nfield * nper
    lo = 0
    hi = nper
    ws = hip_block_vector_workspace(0)
    if (refsum) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER)
    if (rounded) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_BLOCKED_ROUNDED)
    if (world > 1) then
      call slice_bounds(nper, rank, lo, hi)
      shm = shm_ar_open(trim(shmfile)//c_null_char, int(world, c_int), int(rank, c_int))
      if (.not. c_associated(shm)) error stop 'script: cannot map the all-reduce file'
      call hip_block_vector_set_host_allreduce(ws, c_funloc(shm_allreduce), shm)
    end if
    nloc = hi - lo
    call f%init(nfield, nloc, ws)
    call accel%init(f, mvec, compact=compact)
    allocate(host(n), loc(nfield*nloc))
    open(newunit=lin, file=trim(scriptfile), access='stream', form='unformatted', status='old', action='read')
    open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
    if (world > 1) write(lun) lo, hi
    do t = 1, ncalls
      read(lin) code
      select case (nint(code))
      case (0)
        read(lin) host
        do k = 1, nfield
          call f%set_field(k, host((k-1)*nper+lo+1:(k-1)*nper+hi))
        end do
        call accel%accel_update(f)
        do k = 1, nfield
          call f%get_field(k, loc((k-1)*nloc+1:k*nloc))
        end do
        write(lun) real(accel%num_vec(), r8)
        write(lun) loc
      case (1)
        call accel%relax
        write(lun) real(accel%num_vec(), r8)
      case (2)
        call accel%restart
        write(lun) real(accel%num_vec(), r8)
      case (3)
        read(lin) vtol
        call accel%set_vec_tol(vtol)
        write(lun) real(accel%num_vec(), r8)
      case default
        error stop 'script: unknown operation code'
      end select
    end do
    close(lun)
    close(lin)
    if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
    write(*,'(a,i0,a,i0)') 'script: ', ncalls, ' operations, final num_vec ', accel%num_vec()
  end subroutine

  !! contiguous slice [lo, hi) (0-based) of rank r of `world` over n values; lo, hi even
  !! inside the vector (nka_amd/dist.py:slice_bounds)
  subroutine slice_bounds(n, r, lo, hi)
    integer(i8), intent(in) :: n
    integer, intent(in) :: r
    integer(i8), intent(out) :: lo, hi
    lo = (int(r, i8) * n) / world
    hi = (int(r + 1, i8) * n) / world
    if (r > 0) lo = lo - mod(lo, 2_i8)
    if (r + 1 < world) hi = hi - mod(hi, 2_i8)
  end subroutine

  subroutine run_shard
    type(hip_block_vector) :: f
    type(nka) :: accel
    type(c_ptr) :: ws, shm
    real(r8), allocatable :: host(:), pool(:,:), coef(:), loc(:)
    real(r8) :: probe(3)
    character(kind=c_char) :: id128(128)
    integer :: t, k, lun
    integer(i8) :: n, i, lo, hi, nloc
    n = nfield * nper
    call slice_bounds(nper, rank, lo, hi)
    nloc = hi - lo
    ws = hip_block_vector_workspace(0)
    if (refsum) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER)
    if (rounded) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_BLOCKED_ROUNDED)
    shm = shm_ar_open(trim(shmfile)//c_null_char, int(world, c_int), int(rank, c_int))
    if (.not. c_associated(shm)) error stop 'shard: cannot map the all-reduce file'
    if (irccl /= 0) then            ! the device-side hook: RCCL on a one-rank communicator (this box has one GPU)
      call nka_hip_check(nka_hip_comm_unique_id(id128), 'comm_unique_id')
      call hip_block_vector_use_rccl(ws, id128, 1, 0)
    end if
    call hip_block_vector_set_host_allreduce(ws, c_funloc(shm_allreduce), shm)
    probe = real(rank + 1, r8) * [1.0_r8, 2.0_r8, 3.0_r8]       ! prove the communicator before the first update
    call hip_block_vector_allreduce_now(ws, probe)
    if (any(probe /= real(world*(world+1)/2, r8) * [1.0_r8, 2.0_r8, 3.0_r8])) error stop 'shard: all-reduce self-test failed'
    call f%init(nfield, nloc, ws)
    call accel%init(f, mvec, compact=compact)
    allocate(host(n), pool(n,3), coef(3), loc(nfield*nloc))
    do k = 1, 3
      do i = 1, n
        pool(i,k) = lcg()
      end do
    end do
    open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
    write(lun) lo, hi
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
      write(lun) host                             ! the GLOBAL input (every rank draws the same LCG sequence)
      do k = 1, nfield
        call f%set_field(k, host((k-1)*nper+lo+1:(k-1)*nper+hi))
      end do
      call accel%accel_update(f)
      if (t == 7) call accel%relax                ! collective, like every call of a sharded run (F08:58-64)
      do k = 1, nfield
        call f%get_field(k, loc((k-1)*nloc+1:k*nloc))
      end do
      write(lun) real(accel%num_vec(), r8)
      write(lun) accel%state_digest()
      write(lun) loc
    end do
    close(lun)
    if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
    write(*,'(a,i0,a,i0,a,i0,a,i0,a,i0)') 'shard: rank ', rank, ' of ', world, ' slice ', lo, ':', hi, ', final num_vec ', &
                                         accel%num_vec()
  end subroutine

  subroutine run_checkgrid
    type(hip_grid_vector) :: f
    type(nka) :: accel
    type(c_ptr) :: ws
    real(r8), allocatable :: host(:,:), pool(:,:,:), coef(:)
    integer :: t, k, lun, nx, ny, i, j
    nx = nfield
    ny = int(nper)
    ws = hip_block_vector_workspace(0)
    if (refsum) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER)
    if (rounded) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_BLOCKED_ROUNDED)
    call f%init_grid(nx, ny, ws)
    call accel%init(f, mvec, compact=compact)
    allocate(host(0:nx+1,0:ny+1), pool(0:nx+1,0:ny+1,3), coef(3))
    do k = 1, 3
      do j = 0, ny+1
        do i = 0, nx+1
          pool(i,j,k) = lcg()
        end do
      end do
    end do
    open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
    do t = 1, ncalls
      if (mod(t, 5) == 0) then
        do k = 1, 3
          coef(k) = lcg()
        end do
        host = coef(1)*pool(:,:,1) + coef(2)*pool(:,:,2) + coef(3)*pool(:,:,3)
      else
        do j = 0, ny+1
          do i = 0, nx+1
            host(i,j) = lcg()
          end do
        end do
      end if
      write(lun) host
      call f%set_array(host)
      call accel%accel_update(f)
      call f%get_array(host)
      write(lun) real(accel%num_vec(), r8)
      write(lun) host
    end do
    close(lun)
    if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
    write(*,'(a,i0,a,i0)') 'checkgrid: wrote ', ncalls, ' calls, final num_vec ', accel%num_vec()
  end subroutine

  subroutine run_checktile
    type(hip_block_vector) :: f
    type(nka) :: accel
    type(c_ptr) :: ws
    real(r8), allocatable :: small(:), big(:), pool(:,:), coef(:)
    real(r8) :: dev
    integer :: t, k, lun, r
    integer(i8) :: n0, nbig, nperbig, i
    n0 = nfield * nper
    nbig = n0 * rtile
    nperbig = nper * rtile
    ws = hip_block_vector_workspace(0)
    if (refsum) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER)
    if (rounded) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_BLOCKED_ROUNDED)
    call f%init(nfield, nperbig, ws)
    call accel%init(f, mvec, compact=compact)
    allocate(small(n0), big(nbig), pool(n0,3), coef(3))
    do k = 1, 3
      do i = 1, n0
        pool(i,k) = lcg()
      end do
    end do
    open(newunit=lun, file=trim(outfile), access='stream', form='unformatted', status='replace')
    do t = 1, ncalls
      if (mod(t, 7) == 0) then
        do k = 1, 3
          coef(k) = lcg()
        end do
        small = coef(1)*pool(:,1) + coef(2)*pool(:,2) + coef(3)*pool(:,3)
      else
        do i = 1, n0
          small(i) = lcg()
        end do
      end if
      write(lun) small
      do r = 0, rtile-1
        big(r*n0+1:(r+1)*n0) = small
      end do
      do k = 1, nfield
        call f%set_field(k, big((k-1)*nperbig+1:k*nperbig))
      end do
      call accel%accel_update(f)
      do k = 1, nfield
        call f%get_field(k, big((k-1)*nperbig+1:k*nperbig))
      end do
      dev = 0.0_r8
      do r = 1, rtile-1
        dev = dev + sum((big(r*n0+1:(r+1)*n0) - big(1:n0))**2)
      end do
      write(lun) real(accel%num_vec(), r8)
      write(lun) big(1:n0)
      write(lun) sqrt(dev)
    end do
    close(lun)
    if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
    write(*,'(a,i0,a,i0,a,i0)') 'checktile: n = ', nbig, ', wrote ', ncalls, ' calls, final num_vec ', accel%num_vec()
  end subroutine

  subroutine run_bench
    class(hip_block_vector), allocatable :: f
    class(hip_block_vector), allocatable :: inputs(:)
    type(nka) :: accel
    type(c_ptr) :: ws
    real(r8), allocatable :: host(:)
    integer :: t, k, warm, ninp
    integer(i8) :: n, c0, c1, rate
    real(r8) :: secs, per, dummy
    warm = mvec + 3
    ninp = warm + ncalls
    ws = hip_block_vector_workspace(0)
    if (refsum) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_REFERENCE_ORDER)
    if (rounded) call hip_block_vector_set_sum_order(ws, NKA_HIP_SUMS_BLOCKED_ROUNDED)
    if (grid) then            ! NX = nfield, NY = nper: one field of NX*NY cells plus the ghost ring
      allocate(hip_grid_vector :: f)
      allocate(hip_grid_vector :: inputs(ninp))
    else
      allocate(hip_block_vector :: f)
      allocate(hip_block_vector :: inputs(ninp))
    end if
    call make(f)
    n = f%nred
    call accel%init(f, mvec, compact=compact)
    !! independent inputs, resident on the device before the timed region
    allocate(host(f%nper))
    do t = 1, ninp
      call make(inputs(t))
      do k = 1, f%nfield
        call random_number(host)
        host = 2.0_r8*host - 1.0_r8
        call inputs(t)%set_field(k, host)
      end do
    end do
    do t = 1, warm
      call f%copy(inputs(t))
      call accel%accel_update(f)
    end do
    if (accel%num_vec() /= mvec) error stop 'bench: subspace not full after warm-up'
    dummy = f%norm2()                               ! drains the stream
    call system_clock(c0, rate)
    do t = warm+1, ninp
      call accel%accel_update(inputs(t))
    end do
    dummy = inputs(ninp)%norm2()
    call system_clock(c1)
    secs = real(c1 - c0, r8) / real(rate, r8)
    per = secs / ncalls
    if (grid) then
      write(*,'(a,i0,a,i0,a,i0,a,i0,a,l1)') 'abstract-vector path on a grid vector: ', nfield, ' x ', nper, &
          ' cells + ghost ring of ', f%ntot - f%nred, ' mvec=', mvec, ' compact=', compact
    else
      write(*,'(a,i0,a,i0,a,i0,a,l1)') 'abstract-vector path: fields=', nfield, ' n_per_field=', nper, ' mvec=', mvec, &
                                       ' compact=', compact
    end if
    write(*,'(a,f10.3,a,f10.3,a)') 'updates/s ', 1.0_r8/per, '   ms/update ', 1e3_r8*per, ''
    !! bytes per update: hook by hook 8n(12+8m); with the stage hooks of hip_block_vector (ONE pure-read
    !! pass for the norm and both inner-product rows, the combine normalising the new pair itself)
    !! 8n(8+3m), three words per element below the contract figure 8n(11+3m); compact option: 8n(9+2m)
    !! (NKA_HIP_VEC_FUSE_NORM=0: the norm stage as its own pass, two words more)
    if (compact) then
      write(*,'(a,f10.1,a,f10.1)') 'moved GB/s, stage hooks compact (8n(9+2m)) ', 8.0_r8*n*(9+2*mvec)/per/1e9_r8, &
                                   '   contract GB/s (8n(11+3m)) ', 8.0_r8*n*(11+3*mvec)/per/1e9_r8
    else
      write(*,'(a,f10.1,a,f10.1)') 'moved GB/s, stage hooks (8n(8+3m)) ', 8.0_r8*n*(8+3*mvec)/per/1e9_r8, &
                                   '   contract GB/s (8n(11+3m)) ', 8.0_r8*n*(11+3*mvec)/per/1e9_r8
    end if
    write(*,'(a,f8.4)') 'fraction of the 8 TB/s HBM roofline by contract bytes 8n(11+3m) ', &
                        8.0_r8*n*(11+3*mvec)/per/8.0e12_r8
  contains
    subroutine make(v)
      class(hip_block_vector), intent(inout) :: v
      select type (v)
      type is (hip_grid_vector)
        call v%init_grid(nfield, int(nper), ws)
        call v%setval(0.0_r8)
      class default
        call v%init(nfield, nper, ws)
      end select
    end subroutine
  end subroutine

end program nka_vector_driver
