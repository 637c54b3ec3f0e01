!! oracle/ref_f08vec_shim.F90 -- bind(C) handle API over the compiled reference
!! abstract-vector flavour: /root/reference/src-F08-vector/{vector_class,
!! nka_type,grid_vector_type}.F90.  TEST INFRASTRUCTURE ONLY (oracle/_ref/).
!! The vectors are the reference's own grid_vector (nx x ny interior plus a
!! ghost ring); f crosses the boundary as the full (nx+2)*(ny+2) array.

module ref_f08vec_shim
  use, intrinsic :: iso_c_binding
  use nka_type, only: nka
  use grid_vector_type, only: grid_vector
  implicit none
  private
  integer, parameter :: NPOOL = 8
  type(nka), save :: pool(NPOOL)
  type(grid_vector), save :: fvec(NPOOL)
  logical, save :: used(NPOOL) = .false.
contains

  integer(c_int) function ref_f08vec_init(nx, ny, mvec) bind(C, name='ref_f08vec_init')
    integer(c_int), value :: nx, ny, mvec
    integer :: h
    ref_f08vec_init = -1
    do h = 1, NPOOL
      if (.not. used(h)) then
        used(h) = .true.
        call fvec(h)%init(int(nx), int(ny))
        call fvec(h)%setval(0.0_c_double)
        call pool(h)%init(fvec(h), int(mvec))
        ref_f08vec_init = h
        return
      end if
    end do
  end function

  subroutine ref_f08vec_delete(h) bind(C, name='ref_f08vec_delete')
    integer(c_int), value :: h
    if (h >= 1 .and. h <= NPOOL) used(h) = .false.
  end subroutine

  subroutine ref_f08vec_set_vec_tol(h, vtol) bind(C, name='ref_f08vec_set_vec_tol')
    integer(c_int), value :: h
    real(c_double), value :: vtol
    call pool(h)%set_vec_tol(vtol)
  end subroutine

  subroutine ref_f08vec_accel_update(h, f, ntot) bind(C, name='ref_f08vec_accel_update')
    integer(c_int), value :: h, ntot
    real(c_double), intent(inout) :: f(ntot)
    integer :: nx2, ny2
    nx2 = fvec(h)%nx + 2
    ny2 = fvec(h)%ny + 2
    fvec(h)%array(0:,0:) = reshape(f, [nx2, ny2])
    call pool(h)%accel_update(fvec(h))
    f = reshape(fvec(h)%array, [ntot])
  end subroutine

  subroutine ref_f08vec_restart(h) bind(C, name='ref_f08vec_restart')
    integer(c_int), value :: h
    call pool(h)%restart()
  end subroutine

  subroutine ref_f08vec_relax(h) bind(C, name='ref_f08vec_relax')
    integer(c_int), value :: h
    call pool(h)%relax()
  end subroutine

  integer(c_int) function ref_f08vec_num_vec(h) bind(C, name='ref_f08vec_num_vec')
    integer(c_int), value :: h
    ref_f08vec_num_vec = pool(h)%num_vec()
  end function

end module ref_f08vec_shim
