!! nka_type (abstract-vector flavour) -- NKA on user-defined vector objects.
!!
!! Drop-in for module nka_type of src-F08-vector/nka_type.F90:148-171: same type
!! name and type-bound procedures
!!   init(vec, mvec)  set_vec_tol(vtol)  accel_update(f)  restart()  relax()
!!   num_vec()  max_vec()  vec_tol()  defined()
!! The algorithm can only touch the vectors through the hooks of class(vector),
!! so -- as in the reference -- the list bookkeeping and the (mvec+1)^2 Gram /
!! Cholesky matrix live on the host and every O(n) statement is a hook call:
!!   update(-1,f) ; norm2 ; scale(1/s) x2 ; dot x L        (F08V:237-262)
!!   copy ; dot x k ; update(-c,w,c,v) x k ; copy          (F08V:336-382)
!! These statements are issued through the optional STAGE hooks of vector_class
!! (update_norm2, scale_dot_pair_many, update_many_keep / axpy_many_keep) and its
!! batched hooks (dot_pair_many, dot_many, update_many), whose default bodies are
!! exactly the calls listed above -- a user type that does not override them sees
!! only the reference's deferred hooks, in the reference's order.
!! With a device-resident concrete vector (hip_block_vector_type) each hook is a
!! HIP kernel.  The hook sequence, and therefore every rounding of the stored
!! vectors, is the reference's; the scalar step restates F08V:269-368 (see
!! factor_with_drops / solve_normal_equations).

module nka_type

  use, intrinsic :: iso_fortran_env, only: r8 => real64
  use vector_class
  implicit none
  private

  type, public :: nka
    private
    logical :: subspace = .false., pending = .false.
    logical :: compact = .false.                  ! optional: keep v - w in the v slots (see init)
    integer :: mvec = 0
    real(r8) :: vtol = 0.01_r8                    ! default drop tolerance (F08V:152)
    class(vector), allocatable :: v(:), w(:)      ! mvec+1 slots each
    real(r8), allocatable :: h(:,:)               ! raw Gram entries / Cholesky factor
    integer :: first = 0, last = 0, free = 0      ! list heads; 0 terminates a list
    integer, allocatable :: next(:), prev(:)
  contains
    procedure :: init
    procedure :: set_vec_tol
    procedure :: num_vec
    procedure :: max_vec
    procedure :: vec_tol
    procedure :: accel_update
    procedure :: relax
    procedure :: restart
    procedure :: defined
    procedure :: state_digest
  end type nka

contains

  !! call a%init(vec, mvec): slots are clones of vec (F08V:175-188).
  !! compact (optional, default .false.): once a pair is normalised, keep the
  !! difference v - w in the v slot and combine with  f <- f + c*(v - w)  -- the
  !! rounding of the C reference (src-C/...c:423) instead of update3_(-c,w,c,v) --
  !! so the combine streams one vector per pair instead of two.
  subroutine init(this, vec, mvec, compact)
    class(nka), intent(out) :: this
    class(vector), intent(in) :: vec
    integer, intent(in) :: mvec
    logical, intent(in), optional :: compact
    if (mvec <= 0) error stop 'nka%init: mvec must be positive'
    this%mvec = mvec
    if (present(compact)) this%compact = compact
    call vec%clone(this%v, mvec+1)
    call vec%clone(this%w, mvec+1)
    allocate(this%h(mvec+1,mvec+1), this%next(mvec+1), this%prev(mvec+1))
    this%h = 0.0_r8
    this%prev = 0
    call this%restart
  end subroutine

  subroutine set_vec_tol(this, vtol)              ! F08V:190-195
    class(nka), intent(inout) :: this
    real(r8), intent(in) :: vtol
    if (.not.(vtol > 0.0_r8)) error stop 'nka%set_vec_tol: vtol must be positive'
    this%vtol = vtol
  end subroutine

  integer function num_vec(this)                  ! F08V:197-207
    class(nka), intent(in) :: this
    integer :: k
    num_vec = 0
    k = this%first
    do while (k /= 0)
      num_vec = num_vec + 1
      k = this%next(k)
    end do
    if (this%pending) num_vec = num_vec - 1
  end function

  integer function max_vec(this)
    class(nka), intent(in) :: this
    max_vec = this%mvec
  end function

  real(r8) function vec_tol(this)
    class(nka), intent(in) :: this
    vec_tol = this%vtol
  end function

  subroutine restart(this)                        ! F08V:400-414
    class(nka), intent(inout) :: this
    integer :: k
    this%subspace = .false.
    this%pending = .false.
    this%first = 0
    this%last = 0
    this%free = 1
    do k = 1, this%mvec
      this%next(k) = k + 1
    end do
    this%next(this%mvec+1) = 0
  end subroutine

  subroutine relax(this)                          ! F08V:417-435
    class(nka), intent(inout) :: this
    integer :: slot
    if (.not. this%pending) return
    slot = this%first
    this%first = this%next(slot)
    if (this%first == 0) then
      this%last = 0
    else
      this%prev(this%first) = 0
    end if
    call push_free(this, slot)
    this%pending = .false.
  end subroutine

  subroutine push_free(this, slot)
    class(nka), intent(inout) :: this
    integer, intent(in) :: slot
    this%next(slot) = this%free
    this%free = slot
  end subroutine

  !! Row-by-row Cholesky factorisation of the Gram matrix in list order with the
  !! capacity and dependence drops (F08V:269-321).  For list entries j newer than
  !! k, h(j,k) is the raw inner product and h(k,j) the factor entry.
  subroutine factor_with_drops(this)
    class(nka), intent(inout) :: this
    integer :: i, j, k, nvec, before, after
    real(r8) :: pivot, entry
    this%h(this%first,this%first) = 1.0_r8
    nvec = 1
    k = this%next(this%first)
    do while (k /= 0)
      nvec = nvec + 1
      if (nvec > this%mvec) then                  ! capacity: k is the last entry
        this%last = this%prev(k)
        this%next(this%last) = 0
        call push_free(this, k)
        exit
      end if
      pivot = 1.0_r8
      j = this%first
      do while (j /= k)
        entry = this%h(j,k)
        i = this%first
        do while (i /= j)
          entry = entry - this%h(k,i) * this%h(j,i)
          i = this%next(i)
        end do
        entry = entry / this%h(j,j)
        pivot = pivot - entry**2
        this%h(k,j) = entry
        j = this%next(j)
      end do
      if (pivot > this%vtol**2) then
        this%h(k,k) = sqrt(pivot)
        k = this%next(k)
      else                                        ! dependent on the newer vectors: unlink k
        before = this%prev(k)
        after = this%next(k)
        this%next(before) = after
        if (after == 0) then
          this%last = before
        else
          this%prev(after) = before
        end if
        call push_free(this, k)
        nvec = nvec - 1
        k = after
      end if
    end do
    this%subspace = .true.
    this%pending = .false.
  end subroutine

  !! c holds <f,w_j> by slot on entry, the least-squares coefficients on return
  !! (forward then backward substitution, F08V:344-368).
  subroutine solve_normal_equations(this, c)
    class(nka), intent(in) :: this
    real(r8), intent(inout) :: c(:)
    integer :: i, j
    real(r8) :: t
    j = this%first
    do while (j /= 0)
      t = c(j)
      i = this%first
      do while (i /= j)
        t = t - this%h(j,i) * c(i)
        i = this%next(i)
      end do
      c(j) = t / this%h(j,j)
      j = this%next(j)
    end do
    j = this%last
    do while (j /= 0)
      t = c(j)
      i = this%last
      do while (i /= j)
        t = t - this%h(i,j) * c(i)
        i = this%prev(i)
      end do
      c(j) = t / this%h(j,j)
      j = this%prev(j)
    end do
  end subroutine

  !! call a%accel_update(f)                                   F08V:219-397
  !! The O(n) statements are issued as THREE stage hooks of class(vector), placed
  !! where the algorithm needs a value reduced over the whole vector: the norm of
  !! the new difference, the two inner-product rows, and nothing after the
  !! combine.  Each stage's default body is the reference's own hook sequence.
  subroutine accel_update(this, f)
    class(nka), intent(inout) :: this
    class(vector), intent(inout) :: f
    real(r8) :: s, c(this%mvec+1), vals(this%mvec+1), bvals(this%mvec+1), cross
    integer :: k, slot, idx(this%mvec+1), nidx, j
    logical :: have_rows, have_f_row, stored, scaled, fused

    have_rows = .false.
    have_f_row = .false.
    stored = .true.
    scaled = .true.
    fused = .false.
    nidx = 0
    if (this%pending) then
      k = this%next(this%first)                              ! the older entries, in list order
      do while (k /= 0)
        nidx = nidx + 1
        idx(nidx) = k
        k = this%next(k)
      end do
      !! s = ||w1 - f|| ; w1 <- w1 - f now or in a later stage (F08V:237-238).  A vector type may take the RAW
      !! inner products of d = w1 - f in the same pass (fused): <d,w_k>, <f,w_k>, <f,d>.
      s = this%w(this%first)%update_norm2_dots(-1.0_r8, f, this%w, idx(1:nidx), vals(1:nidx), bvals(1:nidx), cross, &
                                               stored, fused)
      if (s == 0.0_r8) then                                  ! nothing to learn from a zero difference
        call this%relax
        have_f_row = fused                                   ! <f,w_k> of the list that remains: already taken
      end if
    end if

    if (this%pending) then
      !! v1 <- v1/s, w1 <- w1/s (F08V:255-256; compact: then v1 <- v1 - w1) and both
      !! inner-product rows of this update: the Gram row <w1,w_k> (F08V:260-264) and
      !! the projection row <f,w_k> (F08V:347) for every older entry, plus <f,w1>.
      !! f is not modified in between, so the values equal the reference's; rows of
      !! entries the factorisation then drops are simply not used.
      !! (`scaled` comes back .false. when the vector type took the rows in a pure-read pass and left
      !!  the normalisation itself to the combine stage below, which reads the pair anyway)
      if (fused) then                                        ! raw sums of d: scale them, nothing was stored
        scaled = .false.
        cross = (1.0_r8/s) * cross
        do j = 1, nidx
          vals(j) = (1.0_r8/s) * vals(j)
        end do
      else if (stored) then
        call this%w(this%first)%scale_dot_pair_many(this%v(this%first), 1.0_r8/s, this%compact, f, this%w, &
                                                    idx(1:nidx), vals(1:nidx), bvals(1:nidx), cross, scaled=scaled, &
                                                    f_row=have_rows)
      else                                                   ! the norm stage left w1 <- w1 - f to this one
        call this%w(this%first)%scale_dot_pair_many(this%v(this%first), 1.0_r8/s, this%compact, f, this%w, &
                                                    idx(1:nidx), vals(1:nidx), bvals(1:nidx), cross, pre_a=-1.0_r8, &
                                                    scaled=scaled, f_row=have_rows)
      end if
      if (fused) have_rows = .true.
      if (.not. have_rows .and. .not. scaled) &
        error stop 'nka%accel_update: a vector type that defers the normalisation must also take the projection row'
      do j = 1, nidx
        this%h(this%first,idx(j)) = vals(j)
      end do
      if (have_rows) then                                    ! the projection row came out of the same pass
        c(this%first) = cross
        do j = 1, nidx
          c(idx(j)) = bvals(j)
        end do
      end if
      call factor_with_drops(this)
    else if (have_f_row) then
      do j = 1, nidx
        c(idx(j)) = bvals(j)
      end do
      have_rows = .true.
    end if

    slot = this%free
    this%free = this%next(slot)

    if (this%subspace) then
      nidx = 0
      k = this%first
      do while (k /= 0)
        nidx = nidx + 1
        idx(nidx) = k
        k = this%next(k)
      end do
      if (.not. have_rows) then
        !! The reference's own sequence, call by call (F08V:336-382) -- what a vector type that overrides nothing
        !! sees, and what remains after relax() / s == 0 when no stage has taken the projection row:
        !! w_new <- f ; <f,w_j> for j = first ... last of the list as it stands AFTER the drops ; the
        !! substitutions ; f <- f - c w + c v for every k in list order ; v_new <- f.
        call this%w(slot)%copy(f)
        call f%dot_many(this%w, idx(1:nidx), vals(1:nidx))
        do j = 1, nidx
          c(idx(j)) = vals(j)
        end do
        call solve_normal_equations(this, c)
        do j = 1, nidx
          vals(j) = c(idx(j))
        end do
        if (this%compact) then
          call f%axpy_many(vals(1:nidx), this%v, idx(1:nidx))
        else
          call f%update_many(-vals(1:nidx), this%w, vals(1:nidx), this%v, idx(1:nidx))
        end if
        call this%v(slot)%copy(f)
      else
        call solve_normal_equations(this, c)
        do j = 1, nidx
          vals(j) = c(idx(j))
        end do
        !! w_new <- f (F08V:336) ; f <- f - c w + c v for every k in list order (F08V:374) ;
        !! v_new <- f (F08V:382): one stage; the two ring stores are named by slot index
        !! (a pair left un-normalised above is entry 1 = this%first of the lists: never dropped)
        if (this%compact) then                                 ! v slots hold v - w: f <- f + c*(v - w)
          if (scaled) then
            call f%axpy_many_keep(vals(1:nidx), this%v, idx(1:nidx), this%w, slot, slot)
          else if (stored) then
            call f%axpy_many_keep(vals(1:nidx), this%v, idx(1:nidx), this%w, slot, slot, pend_a=1.0_r8/s)
          else
            call f%axpy_many_keep(vals(1:nidx), this%v, idx(1:nidx), this%w, slot, slot, pend_a=1.0_r8/s, pend_pre_a=-1.0_r8)
          end if
        else
          if (scaled) then
            call f%update_many_keep(-vals(1:nidx), this%w, vals(1:nidx), this%v, idx(1:nidx), slot, slot)
          else if (stored) then
            call f%update_many_keep(-vals(1:nidx), this%w, vals(1:nidx), this%v, idx(1:nidx), slot, slot, &
                                    pend_a=1.0_r8/s, pend_subtract=.false.)
          else
            call f%update_many_keep(-vals(1:nidx), this%w, vals(1:nidx), this%v, idx(1:nidx), slot, slot, &
                                    pend_a=1.0_r8/s, pend_pre_a=-1.0_r8, pend_subtract=.false.)
          end if
        end if
      end if
    else
      call this%w(slot)%copy(f)                              ! no subspace yet: f is returned unchanged
      call this%v(slot)%copy(f)
    end if

    this%prev(slot) = 0
    this%next(slot) = this%first
    if (this%first == 0) then
      this%last = slot
    else
      this%prev(this%first) = slot
    end if
    this%first = slot
    this%pending = .true.
  end subroutine accel_update

  !! 64-bit FNV-1a digest of the scalar state this rank keeps on the host (flags, lists,
  !! the Gram / Cholesky matrix).  In a sharded run that state is replicated: every rank
  !! must report the same digest after the same calls (SURVEY.md 8e) -- the drop decisions
  !! are taken independently per rank on the all-reduced sums.  Not in the reference.
  function state_digest(this) result(d)
    use, intrinsic :: iso_fortran_env, only: int64
    class(nka), intent(in) :: this
    integer(int64) :: d
    integer :: i, j
    d = -3750763034362895579_int64                       ! 14695981039346656037 as a signed 64-bit pattern
    call mix(merge(1_int64, 0_int64, this%subspace))
    call mix(merge(1_int64, 0_int64, this%pending))
    call mix(int(this%first, int64)); call mix(int(this%last, int64)); call mix(int(this%free, int64))
    do i = 1, this%mvec+1
      call mix(int(this%next(i), int64))
      call mix(int(this%prev(i), int64))
    end do
    do j = 1, this%mvec+1
      do i = 1, this%mvec+1
        call mix(transfer(this%h(i,j), 1_int64))
      end do
    end do
  contains
    subroutine mix(word)
      integer(int64), intent(in) :: word
      integer :: b
      do b = 0, 7
        d = ieor(d, iand(ishft(word, -8*b), 255_int64))
        d = d * 1099511628211_int64                      ! wraps modulo 2^64
      end do
    end subroutine
  end function

  !! Structural invariants of the two lists (F08V:438-501).
  logical function defined(this)
    class(nka), intent(in) :: this
    logical, allocatable :: seen(:)
    integer :: n, k, steps
    defined = .false.
    if (this%mvec < 1) return
    if (.not.allocated(this%v) .or. .not.allocated(this%w) .or. .not.allocated(this%h)) return
    if (.not.allocated(this%next) .or. .not.allocated(this%prev)) return
    n = this%mvec + 1
    if (size(this%v) /= n .or. size(this%w) /= n) return
    if (any(shape(this%h) /= [n,n]) .or. size(this%next) /= n .or. size(this%prev) /= n) return
    if (.not.(this%vtol > 0.0_r8)) return
    if (any(this%next < 0) .or. any(this%next > n)) return
    if (this%first < 0 .or. this%first > n .or. this%free < 0 .or. this%free > n) return
    allocate(seen(n))
    seen = .false.
    if (this%first == 0) then
      if (this%last /= 0) return
    else
      if (this%prev(this%first) /= 0) return
      k = this%first
      steps = 0
      do
        if (seen(k)) return
        seen(k) = .true.
        steps = steps + 1
        if (this%next(k) == 0) exit
        if (this%prev(this%next(k)) /= k) return
        k = this%next(k)
        if (steps > n) return
      end do
      if (this%last /= k) return
    end if
    k = this%free
    do while (k /= 0)
      if (seen(k)) return
      seen(k) = .true.
      k = this%next(k)
    end do
    defined = all(seen)
  end function

end module nka_type
