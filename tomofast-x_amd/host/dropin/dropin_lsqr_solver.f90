!=========================================================================================================
! DROP-IN module `lsqr_solver` for the UNMODIFIED Tomofast-x sources (build recipe: INTEGRATION.md 0).
! Replaces src/inversion/lsqr_solver2.F90: lsqr_solve_sensit (:47-63) and lsqr_solve (:321-329) with the reference's argument lists;
! the solve runs on the GPU (tfx_lsqr_solve of libtfx.so through tfx_reference_api's lsqr_solve_sensit).  The repository's own code.
!
! The reference's joint system always has TWO column blocks of nmodel_components * nelements unknowns, also when only one problem is
! solved (joint_inverse_problem.F90:213-214, :712-739); the device system has one block per LOADED kernel.  This module maps between
! the two: constraint columns of a loaded problem move to that kernel's device columns, the solution comes back at the reference's
! offsets, the unknowns of a problem that is not solved stay zero (their columns are empty in the reference too).
!=========================================================================================================
module lsqr_solver
  use iso_c_binding
  use global_typedefs
  use mpi_tools, only: exit_MPI
  use sparse_matrix
  use tfx_binding
  use tfx_reference_api, only: api_matrix => t_sparse_matrix, api_lsqr_solve_sensit => lsqr_solve_sensit, tfx_api_context, api_check, api_canonical_csr
  implicit none
  private

  public :: lsqr_solve
  public :: lsqr_solve_sensit

contains

  subroutine lsqr_solve_sensit(nlines, ncolumns, niter, rmin, gamma, target_misfit, &
                               matrix_sensit, matrix_cons, u, x, &
                               SOLVE_PROBLEM, nelements, nx, ny, nz, ncomponents, compression_type, &
                               WAVELET_DOMAIN, memory, myrank, nbproc)
    integer, intent(in) :: nlines, ncolumns, niter
    real(kind=CUSTOM_REAL), intent(in) :: rmin, gamma, target_misfit
    logical, intent(in) :: SOLVE_PROBLEM(2)
    integer, intent(in) :: nelements, nx, ny, nz, ncomponents, compression_type
    logical, intent(in) :: WAVELET_DOMAIN
    integer, intent(in) :: myrank, nbproc
    type(t_sparse_matrix), intent(in) :: matrix_sensit
    type(t_sparse_matrix), intent(in) :: matrix_cons
    real(kind=CUSTOM_REAL), intent(inout) :: x(ncolumns)
    real(kind=CUSTOM_REAL), intent(inout) :: u(nlines)
    real(kind=CUSTOM_REAL), intent(out) :: memory

    type(api_matrix) :: dev_sensit, dev_cons
    real(kind=CUSTOM_REAL), allocatable :: x_dev(:)
    logical :: loaded(2), identity
    integer :: slot(2), nr(2), nc(2), row0(2), col0(2), dev_col0(2)
    integer :: p, q, ncols_dev, nl_cons, i
    integer(kind=8) :: k, nel_cons

    ! Sanity check (lsqr_solver2.F90:84-89).
    if (matrix_sensit%get_total_row_number() + matrix_cons%get_total_row_number() /= nlines .or. &
        matrix_sensit%get_ncolumns() /= ncolumns .or. &
        matrix_cons%get_ncolumns() /= ncolumns) then
      call exit_MPI("Wrong matrix sizes in lsqr_solve_sensit! Exiting.", myrank, 0)
    endif
    if (.not. matrix_sensit%is_on_device()) call exit_MPI("lsqr_solve_sensit: the sensitivity kernel is not on the device!", myrank, 0)

    ! ---- the device system: one column block per loaded kernel, in slot order
    ncols_dev = 0
    dev_col0 = 0
    do p = 1, 2
      call matrix_sensit%device_block(p, loaded(p), slot(p), nr(p), nc(p), row0(p), col0(p))
    enddo
    do q = 0, 1
      do p = 1, 2
        if (loaded(p) .and. slot(p) == q) then
          dev_col0(p) = ncols_dev
          ncols_dev = ncols_dev + nc(p)
        endif
      enddo
    enddo
    identity = .true.
    do p = 1, 2
      if (loaded(p)) then
        if (dev_col0(p) /= col0(p)) identity = .false.
        if (SOLVE_PROBLEM(p) .neqv. loaded(p)) call exit_MPI("lsqr_solve_sensit: a solved problem has no kernel on the device!", myrank, p)
      endif
    enddo
    if (ncols_dev /= ncolumns) identity = .false.

    dev_sensit%on_device = .true.
    dev_sensit%nproblems = count(loaded)
    dev_sensit%nl_device = matrix_sensit%get_total_row_number()
    dev_sensit%ncolumns_device = ncols_dev

    ! ---- the constraint rows in device columns
    nl_cons = matrix_cons%get_total_row_number()
    nel_cons = matrix_cons%get_number_elements()
    call dev_cons%initialize(nl_cons, ncols_dev, max(nel_cons, 1_8), myrank)
    if (nel_cons > 0 .or. matrix_cons%get_current_row_number() > 0) then
      dev_cons%ijl(1:nl_cons + 1) = matrix_cons%h%ijl(1:nl_cons + 1)
      dev_cons%sa(1:nel_cons) = matrix_cons%h%sa(1:nel_cons)
      if (identity) then
        dev_cons%ija(1:nel_cons) = matrix_cons%h%ija(1:nel_cons)
      else
        do k = 1, nel_cons
          i = matrix_cons%h%ija(k)
          q = 0
          do p = 1, 2
            if (loaded(p) .and. i > col0(p) .and. i <= col0(p) + nc(p)) q = p
          enddo
          if (q == 0) call exit_MPI("lsqr_solve_sensit: a constraint acts on a problem that has no kernel!", myrank, i)
          dev_cons%ija(k) = i - col0(q) + dev_col0(q)
        enddo
      endif
      dev_cons%nel = nel_cons
    endif
    dev_cons%nl_current = nl_cons
    if (nel_cons == 0) dev_cons%ijl = 0

    allocate(x_dev(max(ncols_dev, 1)))
    call api_lsqr_solve_sensit(nlines, ncols_dev, niter, rmin, gamma, target_misfit, dev_sensit, dev_cons, u, x_dev, &
                               SOLVE_PROBLEM, nelements, nx, ny, nz, ncomponents, compression_type, WAVELET_DOMAIN, memory, &
                               myrank, nbproc)
    x = 0._CUSTOM_REAL                                                            ! lsqr_solver2.F90:120
    do p = 1, 2
      if (loaded(p)) x(col0(p) + 1:col0(p) + nc(p)) = x_dev(dev_col0(p) + 1:dev_col0(p) + nc(p))
    enddo
  end subroutine lsqr_solve_sensit

  ! lsqr_solver2.F90:321-440: min |A x - u| for a matrix assembled on the host (the reference's unit tests, tests_lsqr.f90): the rows
  ! are uploaded as a kernel of a scratch problem and the same device LSQR runs without constraint blocks.
  subroutine lsqr_solve(nlines, nelements, niter, rmin, gamma, matrix, u, x, myrank)
    integer, intent(in) :: nlines, nelements, niter
    real(kind=CUSTOM_REAL), intent(in) :: rmin, gamma
    integer, intent(in) :: myrank
    type(t_sparse_matrix), intent(in) :: matrix
    real(kind=CUSTOM_REAL), intent(inout) :: x(nelements)
    real(kind=CUSTOM_REAL), intent(inout) :: u(nlines)
    type(c_ptr), save :: ctx = c_null_ptr
    type(c_ptr) :: none(1)
    integer(c_int64_t), allocatable :: rowptr(:), rp(:)
    integer(c_int32_t), allocatable :: cols(:)
    real(c_float), allocatable :: vals(:)
    integer(c_int) :: iters
    real(c_double) :: r
    integer(kind=8) :: nel

    if (myrank == 0) print *, 'Entered subroutine lsqr_solve, gamma =', gamma
    if (matrix%get_total_row_number() /= nlines .or. matrix%get_ncolumns() /= nelements) &
      call exit_MPI("Wrong matrix size in lsqr_solve! Exiting.", myrank, 0)                   ! :341-345
    if (matrix%is_on_device()) call exit_MPI("lsqr_solve: use lsqr_solve_sensit for the device-resident kernel.", myrank, 0)
    if (.not. c_associated(ctx)) call api_check(tfx_create(0_c_int, c_null_ptr, ctx), 'tfx_create', myrank)
    nel = matrix%get_number_elements()
    x = 0._CUSTOM_REAL
    if (nel == 0) return
    allocate(rowptr(nlines + 1))
    rowptr = matrix%h%ijl(1:nlines + 1)
    ! rows as add() built them (any column order, repeated columns: sparse_matrix.f90:213-229) -> ascending, distinct columns per row
    call api_canonical_csr(nlines, rowptr, matrix%h%ija, matrix%h%sa, rp, cols, vals)
    call api_check(tfx_matrix_upload_csr(ctx, int(nlines, c_int64_t), int(nelements, c_int64_t), rp, cols, vals), 'lsqr_solve (upload)', myrank)
    none = c_null_ptr
    call api_check(tfx_lsqr_solve(ctx, int(niter, c_int), rmin, gamma, 0.d0, u, 0_c_int, none, none, x, iters, r), 'lsqr_solve', myrank)
    if (myrank == 0) print *, 'End of subroutine lsqr_solve, r =', r, ' iter =', iters
  end subroutine lsqr_solve

end module lsqr_solver
