!=========================================================================================================
! One process per GPU for the Fortran host: MPI (the MPICH of this image, the reference's own launcher `mpiexec -n P`) for
! start-up, the two small LSQR reductions and the gathers of the host-side vectors.  The sensitivity matrix never moves
! through MPI: every rank builds its own column range on its GPU.
!
! The all-reduce hook stages the device buffer through the host (tfx_copy + MPI_Allreduce): 0.8 MB per LSQR iteration at
! the headline size.  A device-side collective (ncclAllReduce on the ctx stream, the RCCL path the Python host uses through
! torch.distributed) plugs into the same tfx_set_allreduce slot.
!
! MPI is initialised only under a launcher (PMI_RANK / PMI_SIZE in the environment); a plain `./tomofastx_amd -p ...`
! runs single-rank without touching MPI.
!=========================================================================================================
module tfx_host_mpi
  use iso_c_binding
  use tfx_binding
  implicit none
  include 'mpif.h'
  logical, save :: mpi_on = .false.
  integer, save :: myrank = 0, nbproc = 1
  type(c_ptr), save :: hook_ctx = c_null_ptr
  real(c_double), allocatable, target, save :: hook_buf(:)

contains

  subroutine host_mpi_init()
    character(len=32) :: v
    integer :: l, st, ierr
    call get_environment_variable('PMI_RANK', v, l, st)
    if (st /= 0 .or. l == 0) call get_environment_variable('PMI_SIZE', v, l, st)
    if (st == 0 .and. l > 0) then
      call MPI_Init(ierr)
      mpi_on = .true.
      call MPI_Comm_rank(MPI_COMM_WORLD, myrank, ierr)
      call MPI_Comm_size(MPI_COMM_WORLD, nbproc, ierr)
    endif
  end subroutine host_mpi_init

  subroutine host_mpi_finalize()
    integer :: ierr
    if (mpi_on) call MPI_Finalize(ierr)
  end subroutine host_mpi_finalize

  subroutine host_mpi_abort()
    integer :: ierr
    if (mpi_on) call MPI_Abort(MPI_COMM_WORLD, 1, ierr)
  end subroutine host_mpi_abort

  ! tfx_allreduce_fn: sum over ranks of n doubles in a DEVICE buffer (lsqr_solver2.F90:214, :511-515; model.F90:290)
  integer(c_int) function allreduce_hook(user, buf, n, stream) bind(C)
    type(c_ptr), value :: user, buf, stream
    integer(c_int64_t), value :: n
    integer :: ierr
    allreduce_hook = 1
    if (.not. allocated(hook_buf)) then
      allocate(hook_buf(max(n, 65536_c_int64_t)))
    else if (size(hook_buf, kind=c_int64_t) < n) then
      deallocate(hook_buf)
      allocate(hook_buf(n))
    endif
    if (tfx_copy(hook_ctx, c_loc(hook_buf), buf, 8_c_int64_t * n) /= 0) return
    call MPI_Allreduce(MPI_IN_PLACE, hook_buf, int(n), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
    if (ierr /= 0) return
    if (tfx_copy(hook_ctx, buf, c_loc(hook_buf), 8_c_int64_t * n) /= 0) return
    allreduce_hook = 0
  end function allreduce_hook

  ! sensit_nnz summed over the ranks (sensitivity_gravmag.F90:322)
  subroutine allreduce_sum_i32(a, n)
    integer, intent(in) :: n
    integer(c_int32_t), intent(inout) :: a(n)
    integer :: ierr
    if (mpi_on .and. nbproc > 1) call MPI_Allreduce(MPI_IN_PLACE, a, n, MPI_INTEGER4, MPI_SUM, MPI_COMM_WORLD, ierr)
  end subroutine allreduce_sum_i32

  subroutine allreduce_sum_dp(a, n)
    integer, intent(in) :: n
    real(c_double), intent(inout) :: a(n)
    integer :: ierr
    if (mpi_on .and. nbproc > 1) call MPI_Allreduce(MPI_IN_PLACE, a, n, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
  end subroutine allreduce_sum_dp

  ! the column slices of all ranks -> the full vector (get_full_array, src/utils/parallel_tools.f90)
  subroutine allgather_slices(loc, nloc, full, counts, displs)
    integer, intent(in) :: nloc, counts(:), displs(:)
    real(c_double), intent(in) :: loc(nloc)
    real(c_double), intent(out) :: full(*)
    integer :: ierr
    if (mpi_on .and. nbproc > 1) then
      call MPI_Allgatherv(loc, nloc, MPI_DOUBLE_PRECISION, full, counts, displs, MPI_DOUBLE_PRECISION, MPI_COMM_WORLD, ierr)
    else
      full(1:nloc) = loc
    endif
  end subroutine allgather_slices

  ! counts of every rank's rows, rank after rank (counts_loc: nparts x nrows_loc, row blocks are dealt out contiguously)
  subroutine allgather_counts(counts_loc, nparts, nrows_loc, counts_all, rows_at, row_displs)
    integer, intent(in) :: nparts, nrows_loc, rows_at(:), row_displs(:)
    integer(c_int32_t), intent(in) :: counts_loc(nparts, *)
    integer(c_int32_t), intent(out) :: counts_all(nparts, *)
    integer :: ierr
    call MPI_Allgatherv(counts_loc, nparts * nrows_loc, MPI_INTEGER4, counts_all, nparts * rows_at, nparts * row_displs, &
                        MPI_INTEGER4, MPI_COMM_WORLD, ierr)
  end subroutine allgather_counts

  subroutine send_piece(dest, n, cols, vals, tag)
    integer, intent(in) :: dest, n, tag
    integer(c_int32_t), intent(in) :: cols(n)
    real(c_float), intent(in) :: vals(n)
    integer :: ierr
    call MPI_Send(cols, n, MPI_INTEGER4, dest, tag, MPI_COMM_WORLD, ierr)
    call MPI_Send(vals, n, MPI_REAL4, dest, tag + 1, MPI_COMM_WORLD, ierr)
  end subroutine send_piece

  subroutine recv_piece(src, n, cols, vals, tag)
    integer, intent(in) :: src, n, tag
    integer(c_int32_t), intent(out) :: cols(n)
    real(c_float), intent(out) :: vals(n)
    integer :: ierr, st(MPI_STATUS_SIZE)
    call MPI_Recv(cols, n, MPI_INTEGER4, src, tag, MPI_COMM_WORLD, st, ierr)
    call MPI_Recv(vals, n, MPI_REAL4, src, tag + 1, MPI_COMM_WORLD, st, ierr)
  end subroutine recv_piece

end module tfx_host_mpi
