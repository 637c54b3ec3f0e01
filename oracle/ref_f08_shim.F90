!! oracle/ref_f08_shim.F90 -- bind(C) handle API over the *compiled reference*
!! module nka_type of /root/reference/src-F08/nka_type.F90 (array flavour).
!! TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ from
!! the reference sources where they lie; nothing of the reference is copied.
!! Handles are small integers into a fixed pool so ctypes can drive several
!! accelerators at once.

module ref_f08_shim
  use, intrinsic :: iso_c_binding
  use nka_type, only: nka
  implicit none
  private
  integer, parameter :: NPOOL = 16
  type(nka), save, target :: pool(NPOOL)
  logical, save :: used(NPOOL) = .false.
contains

  integer(c_int) function ref_f08_init(vlen, mvec) bind(C, name='ref_f08_init')
    integer(c_int), value :: vlen, mvec
    integer :: h
    ref_f08_init = -1
    do h = 1, NPOOL
      if (.not. used(h)) then
        used(h) = .true.
        call pool(h)%init(int(vlen), int(mvec))
        ref_f08_init = h
        return
      end if
    end do
  end function

  subroutine ref_f08_delete(h) bind(C, name='ref_f08_delete')
    integer(c_int), value :: h
    if (h >= 1 .and. h <= NPOOL) then
      if (used(h)) call pool(h)%init(0, 1)  ! releases the big arrays
      used(h) = .false.
    end if
  end subroutine

  subroutine ref_f08_set_vec_tol(h, vtol) bind(C, name='ref_f08_set_vec_tol')
    integer(c_int), value :: h
    real(c_double), value :: vtol
    call pool(h)%set_vec_tol(vtol)
  end subroutine

  subroutine ref_f08_accel_update(h, f, n) bind(C, name='ref_f08_accel_update')
    integer(c_int), value :: h, n
    real(c_double), intent(inout) :: f(n)
    call pool(h)%accel_update(f)
  end subroutine

  subroutine ref_f08_restart(h) bind(C, name='ref_f08_restart')
    integer(c_int), value :: h
    call pool(h)%restart()
  end subroutine

  subroutine ref_f08_relax(h) bind(C, name='ref_f08_relax')
    integer(c_int), value :: h
    call pool(h)%relax()
  end subroutine

  integer(c_int) function ref_f08_num_vec(h) bind(C, name='ref_f08_num_vec')
    integer(c_int), value :: h
    ref_f08_num_vec = pool(h)%num_vec()
  end function

  integer(c_int) function ref_f08_max_vec(h) bind(C, name='ref_f08_max_vec')
    integer(c_int), value :: h
    ref_f08_max_vec = pool(h)%max_vec()
  end function

  integer(c_int) function ref_f08_vec_len(h) bind(C, name='ref_f08_vec_len')
    integer(c_int), value :: h
    ref_f08_vec_len = pool(h)%vec_len()
  end function

  real(c_double) function ref_f08_vec_tol(h) bind(C, name='ref_f08_vec_tol')
    integer(c_int), value :: h
    ref_f08_vec_tol = pool(h)%vec_tol()
  end function

  integer(c_int) function ref_f08_defined(h) bind(C, name='ref_f08_defined')
    integer(c_int), value :: h
    ref_f08_defined = merge(1, 0, pool(h)%defined())
  end function

end module ref_f08_shim
