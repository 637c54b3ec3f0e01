!! hook_trace_driver -- TEST INFRASTRUCTURE (no GPU): drives the vector flavour of the accelerator
!! with the tracing vector type (trace_vector_type.F90) through a call sequence that visits every
!! branch of accel_update -- first call, growth to capacity, capacity drops, dependence drops in
!! mid-list, a zero difference (s == 0 -> relax), relax(), restart(), a tighter set_vec_tol --
!! using only the methods the reference's nka type has, so that the SAME source builds against the
!! reference's modules and against this repository's (oracle/Makefile: hooktrace_ref / hooktrace_ours).
!!
!!   hook_trace_driver N MVEC NCALLS TRACEFILE
!!
!! Trace: the vector type's hook lines, and after every accel_update a line
!!   "call t num_vec k out <bits of sum(f)>".

program hook_trace_driver

  use, intrinsic :: iso_fortran_env, only: r8 => real64, i8 => int64
  use vector_class
  use trace_vector_type
  use nka_type
  implicit none

  character(256) :: arg, tracefile
  integer :: n, mvec, ncalls, t, k, i
  integer(i8) :: lcg_state = 1
  type(trace_vector) :: f
  type(nka) :: accel
  real(r8), allocatable :: host(:), pool(:,:), coef(:)
  character(16) :: h

  call get_command_argument(1, arg); read(arg,*) n
  call get_command_argument(2, arg); read(arg,*) mvec
  call get_command_argument(3, arg); read(arg,*) ncalls
  call get_command_argument(4, tracefile)

  open(newunit=trace_unit, file=trim(tracefile), status='replace', action='write')
  call f%init(n)
  call accel%init(f, mvec)
  allocate(host(n), pool(n,3), coef(3))
  do k = 1, 3
    do i = 1, n
      pool(i,k) = lcg()
    end do
  end do
  do t = 1, ncalls
    if (mod(t, 10) >= 5) then                ! five in a row from a 3-dimensional pool: their differences are
                                             ! dependent, the factorisation drops in mid-list
      do k = 1, 3
        coef(k) = lcg()
      end do
      host = coef(1)*pool(:,1) + coef(2)*pool(:,2) + coef(3)*pool(:,3)
    else if (t == 12) then                   ! the previous input again: s == 0
      continue
    else
      do i = 1, n
        host(i) = lcg()
      end do
    end if
    f%x = host
    call accel%accel_update(f)
    write(h,'(z16.16)') transfer(sum(f%x), 1_i8)
    write(trace_unit,'(a,1x,i0,1x,a,1x,i0,1x,a,1x,a)') 'call', t, 'num_vec', accel%num_vec(), 'out', h
    if (t == 7) then
      call accel%relax
      write(trace_unit,'(a)') 'relax'
    end if
    if (t == 28) then
      call accel%restart
      write(trace_unit,'(a)') 'restart'
    end if
    if (t == 24) then
      call accel%set_vec_tol(0.5_r8)
      write(trace_unit,'(a)') 'set_vec_tol'
    end if
  end do
  close(trace_unit)
  if (.not. accel%defined()) error stop 'accelerator not well defined after the run'
  write(*,'(a,i0)') 'hook_trace_driver: final num_vec ', accel%num_vec()

contains

  real(r8) function lcg()
    lcg_state = mod(1103515245_i8*lcg_state + 12345_i8, 2147483648_i8)
    lcg = real(lcg_state, r8) / 1073741824.0_r8 - 1.0_r8
  end function

end program hook_trace_driver
