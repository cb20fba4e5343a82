!=========================================================================================================
! DROP-IN module `sparse_matrix` for the UNMODIFIED Tomofast-x sources (build recipe: INTEGRATION.md 0).
!
! Replaces src/inversion/sparse_matrix.f90 of the reference: same module name, same public type t_sparse_matrix, same
! type-bound procedures with the same argument lists (sparse_matrix.f90:72-98), so that the reference's own
! joint_inverse_problem.F90, model.F90, damping.F90, admm_method.F90, cross_gradient.F90, clustering.F90,
! damping_gradient.F90, problem_joint_gravmag.F90 and its unit tests compile against it as they are.
!
! Two kinds of objects hide behind the one type, as in the reference's own use of it:
!   * the sensitivity matrix (joint_inverse_problem.F90:213): its kernels live on the GPU (read_sensitivity_kernel of the drop-in
!     module sensitivity_gravmag registers them here), every product is a HIP kernel of libtfx.so;
!   * a matrix assembled on the host with add / add_row / new_row (the constraint rows, unit-test matrices): kept as plain rows
!     here; lsqr_solve_sensit hands the constraint rows to the GPU, and a product asked of such a matrix (only the unit tests
!     do) uploads it to a scratch context first - there is NO host-side product.
! This file is the repository's own code; it `use`s the reference's global_typedefs / mpi_tools, which is why it only compiles
! where the reference sources are (the development container).
!=========================================================================================================
module sparse_matrix
  use iso_c_binding
  use global_typedefs
  use mpi_tools, only: exit_MPI
  use tfx_binding
  use tfx_reference_api, only: api_matrix => t_sparse_matrix, tfx_api_context, api_check, api_canonical_csr
  implicit none
  private

  type, public :: t_sparse_matrix
    private
    type(api_matrix), public :: h                    ! the host-side rows (tfx_reference_api's builder: every row has an offset)
    integer :: nl = 0, ncolumns = 0
    integer(kind=8) :: nnz_pred = 0
    logical :: have_rows = .false.
    integer :: uid = 0                               ! serial number (initialize): which matrix the scratch context holds
    ! device-resident kernels (the sensitivity matrix): block p of the joint system = kernel of problem p
    logical :: on_device = .false.
    logical :: loaded(2) = .false.
    integer :: slot(2) = -1                          ! tfx_select_problem slot
    integer :: nrows_p(2) = 0, ncols_p(2) = 0        ! size of the kernel
    integer :: row0(2) = 0, col0(2) = 0              ! where its block starts in the joint matrix (0-based)
    real(kind=CUSTOM_REAL), allocatable, public :: lsqr_var(:)
    integer, public :: tag = 0
  contains
    private
    procedure, public, pass :: initialize => sparse_matrix_initialize
    procedure, public, pass :: reset => sparse_matrix_reset
    procedure, public, pass :: finalize => sparse_matrix_finalize
    procedure, public, pass :: add => sparse_matrix_add
    procedure, public, pass :: add_row => sparse_matrix_add_row
    procedure, public, pass :: new_row => sparse_matrix_new_row
    procedure, public, pass :: add_empty_rows => sparse_matrix_add_empty_rows
    procedure, public, pass :: mult_vector => sparse_matrix_mult_vector
    procedure, public, pass :: add_mult_vector => sparse_matrix_add_mult_vector
    procedure, public, pass :: part_mult_vector => sparse_matrix_part_mult_vector
    procedure, public, pass :: trans_mult_vector => sparse_matrix_trans_mult_vector
    procedure, public, pass :: add_trans_mult_vector => sparse_matrix_add_trans_mult_vector
    procedure, public, pass :: normalize_columns => sparse_matrix_normalize_columns
    procedure, public, pass :: get_total_row_number => sparse_matrix_get_total_row_number
    procedure, public, pass :: get_current_row_number => sparse_matrix_get_current_row_number
    procedure, public, pass :: get_ncolumns => sparse_matrix_get_ncolumns
    procedure, public, pass :: get_number_elements => sparse_matrix_get_number_elements
    procedure, public, pass :: get_nnz => sparse_matrix_get_nnz
    ! not in the reference: how the other drop-in modules reach the device side
    procedure, public, pass :: register_device_kernel => sparse_matrix_register_device_kernel
    procedure, public, pass :: is_on_device => sparse_matrix_is_on_device
    procedure, public, pass :: device_block => sparse_matrix_device_block
  end type t_sparse_matrix

  type(c_ptr), save :: scratch_ctx = c_null_ptr      ! products of host-assembled matrices (unit tests)
  ! what the scratch context holds: the rows of ONE host-assembled matrix (identified by the serial number it got in initialize, its
  ! shape and its entry count); every procedure that changes rows clears it, so a matrix multiplied repeatedly is uploaded once
  integer, save :: uploaded_owner = 0, next_uid = 0
  integer(kind=8), save :: uploaded_nnz = -1
  integer, save :: uploaded_nl = -1, uploaded_ncolumns = -1, uploaded_row = -1

contains

  subroutine check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    call api_check(rc, where, 0)                     ! (the reference's convention: banner + abort, mpi_tools.F90:29-53)
  end subroutine check

  subroutine sparse_matrix_initialize(this, nl, ncolumns, nnz, myrank, nl_empty)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: nl, ncolumns, myrank
    integer, intent(in), optional :: nl_empty
    integer(kind=8), intent(in) :: nnz
    this%nl = nl
    this%ncolumns = ncolumns
    this%on_device = .false.
    this%loaded = .false.
    this%slot = -1
    ! The row storage is set aside when the first row arrives: the sensitivity matrix (joint_inverse_problem.F90:213, nnz = that of
    ! the kernels) never receives one here - read_sensitivity_kernel registers its kernels on the device.
    this%nnz_pred = nnz
    this%have_rows = .false.
    next_uid = next_uid + 1
    this%uid = next_uid
    uploaded_owner = 0
    if (present(nl_empty)) continue
  end subroutine sparse_matrix_initialize

  subroutine need_rows(this, myrank)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: myrank
    if (this%have_rows) return
    call this%h%initialize(this%nl, this%ncolumns, this%nnz_pred, myrank)
    this%have_rows = .true.
  end subroutine need_rows

  subroutine sparse_matrix_reset(this)
    class(t_sparse_matrix), intent(inout) :: this
    uploaded_owner = 0
    if (this%have_rows) call this%h%reset()
  end subroutine sparse_matrix_reset

  subroutine sparse_matrix_finalize(this, myrank)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: myrank
    if (this%on_device) return                       ! (the kernels were finished on the device when they were loaded)
    call need_rows(this, myrank)
    call this%h%finalize(myrank)                     ! sparse_matrix.f90:157-182: every row must have been closed
  end subroutine sparse_matrix_finalize

  subroutine sparse_matrix_add(this, value, column, myrank)
    class(t_sparse_matrix), intent(inout) :: this
    real(kind=CUSTOM_REAL), intent(in) :: value
    integer, intent(in) :: column, myrank
    call need_rows(this, myrank)
    uploaded_owner = 0
    call this%h%add(value, column, myrank)
  end subroutine sparse_matrix_add

  subroutine sparse_matrix_add_row(this, nel_add, values, columns, myrank)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: nel_add
    real(kind=MATRIX_PRECISION), intent(in) :: values(nel_add)
    integer, intent(in) :: columns(nel_add)
    integer, intent(in) :: myrank
    call need_rows(this, myrank)
    uploaded_owner = 0
    call this%h%add_row(nel_add, values, columns, myrank)
  end subroutine sparse_matrix_add_row

  subroutine sparse_matrix_new_row(this, myrank)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: myrank
    call need_rows(this, myrank)
    uploaded_owner = 0
    call this%h%new_row(myrank)
  end subroutine sparse_matrix_new_row

  subroutine sparse_matrix_add_empty_rows(this, nrows, myrank)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: nrows, myrank
    call need_rows(this, myrank)
    uploaded_owner = 0
    call this%h%add_empty_rows(nrows, myrank)
  end subroutine sparse_matrix_add_empty_rows

  ! ---- device side ------------------------------------------------------------------------------------
  ! read_sensitivity_kernel: the kernel of problem p sits in `slot`; its block of the joint matrix starts at (row0, col0)
  subroutine sparse_matrix_register_device_kernel(this, p, slot, nrows, ncols, row0, col0)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: p, slot, nrows, ncols, row0, col0
    this%on_device = .true.
    this%loaded(p) = .true.
    this%slot(p) = slot
    this%nrows_p(p) = nrows
    this%ncols_p(p) = ncols
    this%row0(p) = row0
    this%col0(p) = col0
  end subroutine sparse_matrix_register_device_kernel

  pure logical function sparse_matrix_is_on_device(this)
    class(t_sparse_matrix), intent(in) :: this
    sparse_matrix_is_on_device = this%on_device
  end function sparse_matrix_is_on_device

  pure subroutine sparse_matrix_device_block(this, p, loaded, slot, nrows, ncols, row0, col0)
    class(t_sparse_matrix), intent(in) :: this
    integer, intent(in) :: p
    logical, intent(out) :: loaded
    integer, intent(out) :: slot, nrows, ncols, row0, col0
    loaded = this%loaded(p); slot = this%slot(p); nrows = this%nrows_p(p); ncols = this%ncols_p(p)
    row0 = this%row0(p); col0 = this%col0(p)
  end subroutine sparse_matrix_device_block

  ! The context a product runs in: the API's context for the device-resident kernels; for a host-assembled matrix a scratch context
  ! that receives the rows first (tfx_matrix_upload_csr).
  function product_context(this) result(ctx)
    class(t_sparse_matrix), intent(in) :: this
    type(c_ptr) :: ctx
    integer(c_int64_t) :: nnz
    integer(c_int64_t), allocatable :: rowptr(:), rp(:)
    integer(c_int32_t), allocatable :: cols(:)
    real(c_float), allocatable :: vals(:)
    if (this%on_device) then
      ctx = tfx_api_context(0, 1)
      return
    endif
    if (.not. c_associated(scratch_ctx)) call check(tfx_create(0_c_int, c_null_ptr, scratch_ctx), 'tfx_create')
    ctx = scratch_ctx
    nnz = this%h%get_number_elements()
    if (uploaded_owner == this%uid .and. this%uid /= 0 .and. uploaded_nnz == nnz .and. uploaded_nl == this%nl .and. &
        uploaded_ncolumns == this%ncolumns .and. uploaded_row == this%h%get_current_row_number()) return      ! these rows are there already
    allocate(rowptr(this%nl + 1))
    rowptr = this%h%ijl(1:this%nl + 1)
    rowptr(this%h%get_current_row_number() + 2:) = nnz             ! (rows not closed yet are empty)
    ! add() takes any column order and repeated columns (sparse_matrix.f90:213-229); the device layout wants ascending, distinct ones
    call api_canonical_csr(this%nl, rowptr, this%h%ija, this%h%sa, rp, cols, vals)
    call check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem')
    call check(tfx_matrix_upload_csr(ctx, int(this%nl, c_int64_t), int(this%ncolumns, c_int64_t), rp, cols, vals), 'tfx_matrix_upload_csr')
    uploaded_owner = this%uid
    uploaded_nnz = nnz; uploaded_nl = this%nl; uploaded_ncolumns = this%ncolumns; uploaded_row = this%h%get_current_row_number()
  end function product_context

  ! b (+)= A x over the blocks of the joint matrix (device) or the uploaded rows (host-assembled)
  subroutine product(this, x, b, add, transposed)
    class(t_sparse_matrix), intent(in) :: this
    real(kind=CUSTOM_REAL), intent(in) :: x(*)
    real(kind=CUSTOM_REAL), intent(inout) :: b(*)
    integer, intent(in) :: add
    logical, intent(in) :: transposed
    type(c_ptr) :: ctx
    integer :: p
    if (.not. this%on_device) then
      if (this%get_number_elements() == 0) then
        if (add == 0) b(1:merge(this%ncolumns, this%nl, transposed)) = 0._CUSTOM_REAL
        return
      endif
      ctx = product_context(this)
      if (transposed) then
        call check(tfx_spmtv(ctx, x, b, int(add, c_int)), 'trans_mult_vector')
      else
        call check(tfx_spmv(ctx, x, b, int(add, c_int)), 'mult_vector')
      endif
      return
    endif
    ctx = product_context(this)
    if (add == 0) b(1:merge(this%ncolumns, this%nl, transposed)) = 0._CUSTOM_REAL
    do p = 1, 2
      if (.not. this%loaded(p)) cycle
      call check(tfx_select_problem(ctx, int(this%slot(p), c_int)), 'tfx_select_problem')
      if (transposed) then
        call check(tfx_spmtv(ctx, x(this%row0(p) + 1), b(this%col0(p) + 1), int(add, c_int)), 'trans_mult_vector')
      else
        call check(tfx_spmv(ctx, x(this%col0(p) + 1), b(this%row0(p) + 1), int(add, c_int)), 'mult_vector')
      endif
    enddo
    call check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem')
  end subroutine product

  subroutine sparse_matrix_mult_vector(this, x, b)                       ! sparse_matrix.f90:298-308
    class(t_sparse_matrix), intent(in) :: this
    real(kind=CUSTOM_REAL), intent(in) :: x(this%ncolumns)
    real(kind=CUSTOM_REAL), intent(out) :: b(this%nl)
    call product(this, x, b, 0, .false.)
  end subroutine sparse_matrix_mult_vector

  subroutine sparse_matrix_add_mult_vector(this, x, b)                   ! :313-329
    class(t_sparse_matrix), intent(in) :: this
    real(kind=CUSTOM_REAL), intent(in) :: x(this%ncolumns)
    real(kind=CUSTOM_REAL), intent(inout) :: b(this%nl)
    call product(this, x, b, 1, .false.)
  end subroutine sparse_matrix_add_mult_vector

  subroutine sparse_matrix_trans_mult_vector(this, x, b)                 ! :373-383
    class(t_sparse_matrix), intent(in) :: this
    real(kind=CUSTOM_REAL), intent(in) :: x(this%nl)
    real(kind=CUSTOM_REAL), intent(out) :: b(this%ncolumns)
    call product(this, x, b, 0, .true.)
  end subroutine sparse_matrix_trans_mult_vector

  subroutine sparse_matrix_add_trans_mult_vector(this, x, b)             ! :388-405
    class(t_sparse_matrix), intent(in) :: this
    real(kind=CUSTOM_REAL), intent(in) :: x(this%nl)
    real(kind=CUSTOM_REAL), intent(inout) :: b(this%ncolumns)
    call product(this, x, b, 1, .true.)
  end subroutine sparse_matrix_add_trans_mult_vector

  ! :335-367: rows [line_start, line_start + ndata) times x, columns shifted by param_shift = the block of one kernel
  subroutine sparse_matrix_part_mult_vector(this, nelements, x, ndata, b, line_start, param_shift, myrank)
    class(t_sparse_matrix), intent(in) :: this
    integer, intent(in) :: nelements, ndata
    real(kind=CUSTOM_REAL), intent(in) :: x(nelements)
    integer, intent(in) :: line_start, param_shift
    integer, intent(in) :: myrank
    real(kind=CUSTOM_REAL), intent(out) :: b(ndata)
    type(c_ptr) :: ctx
    integer :: p, line_end
    line_end = line_start + ndata - 1
    if (line_start < 1 .or. line_start > this%nl .or. line_end < 1 .or. line_end > this%nl) &
      call exit_MPI("Wrong line index in sparse_matrix_part_mult_vector!", myrank, 0)
    if (.not. this%on_device) call exit_MPI("part_mult_vector: the sensitivity kernel is not on the device!", myrank, 0)
    do p = 1, 2
      if (.not. this%loaded(p)) cycle
      if (this%row0(p) + 1 == line_start .and. this%col0(p) == param_shift) then
        if (this%nrows_p(p) /= ndata .or. this%ncols_p(p) /= nelements) &
          call exit_MPI("part_mult_vector: the block does not match the kernel on the device!", myrank, p)
        ctx = tfx_api_context(myrank, 1)
        call check(tfx_select_problem(ctx, int(this%slot(p), c_int)), 'tfx_select_problem')
        call check(tfx_spmv(ctx, x, b, 0_c_int), 'part_mult_vector')
        call check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem')
        return
      endif
    enddo
    call exit_MPI("part_mult_vector: no kernel starts at this line / parameter shift!", myrank, line_start)
  end subroutine sparse_matrix_part_mult_vector

  ! :414-443 - on the GPU (tfx_matrix_normalize_columns); a host-assembled matrix gets its scaled values back
  subroutine sparse_matrix_normalize_columns(this, column_norm)
    class(t_sparse_matrix), intent(inout) :: this
    real(kind=CUSTOM_REAL), intent(out) :: column_norm(this%ncolumns)
    type(c_ptr) :: ctx
    integer(c_int64_t), allocatable :: rowptr(:)
    integer(c_int32_t), allocatable :: cols(:)
    integer :: p
    if (this%on_device) then
      ctx = product_context(this)
      column_norm = 0._CUSTOM_REAL
      do p = 1, 2
        if (.not. this%loaded(p)) cycle
        call check(tfx_select_problem(ctx, int(this%slot(p), c_int)), 'tfx_select_problem')
        call check(tfx_matrix_normalize_columns(ctx, column_norm(this%col0(p) + 1:)), 'normalize_columns')
      enddo
      call check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem')
      return
    endif
    if (this%get_number_elements() == 0) then
      column_norm = 0._CUSTOM_REAL
      return
    endif
    ctx = product_context(this)
    call check(tfx_matrix_normalize_columns(ctx, column_norm), 'normalize_columns')
    ! the scaled values go back into the rows as the caller stored them: entry k of the host rows went into entry where(k) of the uploaded
    ! (sorted, merged) rows; of several entries of one column the first receives the scaled sum and the others zero (the row's
    ! products are unchanged)
    block
      integer(c_int64_t), allocatable :: rp_in(:), rp(:), where(:)
      integer(c_int32_t), allocatable :: cc(:)
      real(c_float), allocatable :: vv(:), dv(:)
      logical, allocatable :: taken(:)
      integer(c_int64_t) :: k, nnz
      nnz = this%h%get_number_elements()
      allocate(rp_in(this%nl + 1))
      rp_in = this%h%ijl(1:this%nl + 1)
      rp_in(this%h%get_current_row_number() + 2:) = nnz
      call api_canonical_csr(this%nl, rp_in, this%h%ija, this%h%sa, rp, cc, vv, where)
      allocate(rowptr(this%nl + 1), cols(max(rp(this%nl + 1), 1_c_int64_t)), dv(max(rp(this%nl + 1), 1_c_int64_t)), taken(max(rp(this%nl + 1), 1_c_int64_t)))
      call check(tfx_matrix_download_csr(ctx, rowptr, cols, dv), 'tfx_matrix_download_csr')
      taken = .false.
      do k = 1, nnz
        if (taken(where(k))) then
          this%h%sa(k) = 0._c_float
        else
          this%h%sa(k) = dv(where(k))
          taken(where(k)) = .true.
        endif
      enddo
    end block
    uploaded_owner = 0                      ! (the host rows were rewritten: the next product uploads them again)
  end subroutine sparse_matrix_normalize_columns

  pure function sparse_matrix_get_total_row_number(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer :: res
    res = this%nl
  end function sparse_matrix_get_total_row_number

  pure function sparse_matrix_get_current_row_number(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer :: res
    res = 0
    if (this%have_rows) res = this%h%get_current_row_number()
  end function sparse_matrix_get_current_row_number

  pure function sparse_matrix_get_ncolumns(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer :: res
    res = this%ncolumns
  end function sparse_matrix_get_ncolumns

  pure function sparse_matrix_get_number_elements(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer(kind=8) :: res
    res = 0
    if (this%have_rows) res = this%h%get_number_elements()
  end function sparse_matrix_get_number_elements

  pure function sparse_matrix_get_nnz(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer(kind=8) :: res
    res = this%nnz_pred
  end function sparse_matrix_get_nnz

end module sparse_matrix
