! Golden-vector driver (OUR code): builds two t_sparse_matrix objects (S and C) through the reference's
! public build API (src/inversion/sparse_matrix.f90:213-293,157-183), then records
!   S.x (mult_vector :298), S^T.y (trans_mult_vector :373), and the solution of
!   lsqr_solve_sensit (src/inversion/lsqr_solver2.F90:47-308) for several iteration counts.
! Input stream file (big-endian):
!   int32: nl_s, nl_c, ncols, nnz_s, nnz_c, nruns
!   S: int32 rowcount(nl_s), int32 cols(nnz_s) (1-based), real4 vals(nnz_s)
!   C: int32 rowcount(nl_c), int32 cols(nnz_c), real4 vals(nnz_c)
!   real8 xin(ncols), real8 yin(nl_s), real8 b(nl_s+nl_c)
!   per run: int32 niter, real8 rmin, real8 gamma
! Output: real8 Sx(nl_s), real8 STy(ncols), then per run real8 x(ncols)
program gold_lsqr
  use mpi
  use global_typedefs
  use sparse_matrix
  use lsqr_solver
  implicit none
  integer :: nl_s, nl_c, ncols, nnz_s, nnz_c, nruns, ierr, i, irun, niter, p
  integer, allocatable :: rc_s(:), cols_s(:), rc_c(:), cols_c(:)
  real(kind=4), allocatable :: vals_s(:), vals_c(:)
  real(kind=8), allocatable :: xin(:), yin(:), b(:), u(:), x(:), sx(:), sty(:)
  real(kind=8) :: rmin, gamma, memory
  type(t_sparse_matrix) :: S, C
  character(len=512) :: fin, fout
  logical :: SOLVE_PROBLEM(2)

  call MPI_Init(ierr)
  read(*, '(a)') fin
  read(*, '(a)') fout
  open(21, file=trim(fin), form='unformatted', access='stream', status='old', action='read')
  read(21) nl_s, nl_c, ncols, nnz_s, nnz_c, nruns
  allocate(rc_s(nl_s), cols_s(nnz_s), vals_s(nnz_s), rc_c(nl_c), cols_c(nnz_c), vals_c(nnz_c))
  allocate(xin(ncols), yin(nl_s), b(nl_s + nl_c), u(nl_s + nl_c), x(ncols), sx(nl_s), sty(ncols))
  read(21) rc_s, cols_s, vals_s
  read(21) rc_c, cols_c, vals_c
  read(21) xin, yin, b

  call S%initialize(nl_s, ncols, int(nnz_s, 8), 0)
  p = 0
  do i = 1, nl_s
    if (rc_s(i) > 0) call S%add_row(rc_s(i), vals_s(p + 1 : p + rc_s(i)), cols_s(p + 1 : p + rc_s(i)), 0)
    p = p + rc_s(i)
    call S%new_row(0)
  enddo
  call S%finalize(0)

  call C%initialize(nl_c, ncols, int(max(nnz_c, 1), 8), 0)
  p = 0
  do i = 1, nl_c
    if (rc_c(i) > 0) call C%add_row(rc_c(i), vals_c(p + 1 : p + rc_c(i)), cols_c(p + 1 : p + rc_c(i)), 0)
    p = p + rc_c(i)
    call C%new_row(0)
  enddo
  call C%finalize(0)

  open(22, file=trim(fout), form='unformatted', access='stream', status='replace', action='write')
  call S%mult_vector(xin, sx)
  call S%trans_mult_vector(yin, sty)
  write(22) sx, sty

  SOLVE_PROBLEM = .true.
  do irun = 1, nruns
    read(21) niter, rmin, gamma
    u = b
    x = 0.d0
    call lsqr_solve_sensit(nl_s + nl_c, ncols, niter, rmin, gamma, -1.d0, S, C, u, x, &
                           SOLVE_PROBLEM, ncols, ncols, 1, 1, 1, 0, .true., memory, 0, 1)
    write(22) x
  enddo
  close(21)
  close(22)
  call MPI_Finalize(ierr)
end program gold_lsqr
