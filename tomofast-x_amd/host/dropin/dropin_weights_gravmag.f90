!=========================================================================================================
! DROP-IN module `weights_gravmag` for the UNMODIFIED Tomofast-x sources (build recipe: INTEGRATION.md 0).
! Replaces src/forward/gravmag/weights_gravmag.f90: calculate_depth_weight (:46-196; types 1, 2, 3 on the GPU: k_depth_weight,
! k_distance_weight, k_mindist_weight of libtfx.so) and apply_local_depth_weighting (:255-310; a file read and a division per cell:
! host control plane, restated here because it lives in the replaced module).  Same argument lists.  The repository's own code.
!=========================================================================================================
module weights_gravmag
  use global_typedefs
  use mpi_tools, only: exit_MPI
  use parameters_gravmag
  use grid
  use data_gravmag
  use inversion_arrays
  use parallel_tools, only: get_nsmaller
  use dropin_gravmag_convert
  use tfx_reference_api, only: api_par_base => t_parameters_base, api_grid => t_grid, api_data => t_data, &
                               api_calculate_depth_weight => calculate_depth_weight
  implicit none
  private

  public :: calculate_depth_weight
  public :: apply_local_depth_weighting

contains

  subroutine calculate_depth_weight(par, iarr, grid_full, data, myrank, nbproc)
    class(t_parameters_base), intent(in) :: par
    type(t_grid), intent(in) :: grid_full
    integer, intent(in) :: myrank, nbproc
    type(t_data), intent(in) :: data
    type(t_inversion_arrays), intent(inout) :: iarr
    class(api_par_base), allocatable :: apar
    type(api_grid) :: agrid
    type(api_data) :: adata
    real(kind=CUSTOM_REAL), allocatable :: cw_full(:)
    integer :: nsmaller

    call to_api_parameters(par, apar)
    call to_api_grid(grid_full, agrid)
    call to_api_data(data, adata)
    allocate(cw_full(par%nx * par%ny * par%nz))
    call api_calculate_depth_weight(apar, cw_full, agrid, adata, myrank, nbproc)       ! all cells, on the GPU
    ! this rank's cells (weights_gravmag.f90:66-67)
    nsmaller = get_nsmaller(par%nelements, myrank, nbproc)
    iarr%column_weight(1:par%nelements) = cw_full(nsmaller + 1:nsmaller + par%nelements)
  end subroutine calculate_depth_weight

  ! weights_gravmag.f90:255-310: column_weight /= local_weight (a zero local weight zeroes the column weight), the local weights
  ! read from par%local_weight_file (count line, then one value per cell)
  subroutine apply_local_depth_weighting(par, column_weight, myrank, nbproc)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: myrank, nbproc
    real(kind=CUSTOM_REAL), intent(inout) :: column_weight(par%nelements)
    real(kind=CUSTOM_REAL), allocatable :: lw(:)
    character(len=256) :: msg
    integer :: ierr, u, ntotal, nread, nsmaller, i

    if (par%apply_local_weight <= 0) return
    open(newunit=u, file=trim(par%local_weight_file), status='old', action='read', iostat=ierr, iomsg=msg)
    if (ierr /= 0) call exit_MPI("Error in opening the local weight file! path="//par%local_weight_file//" iomsg="//msg, myrank, ierr)
    read(u, *, iostat=ierr) nread
    ntotal = par%nx * par%ny * par%nz
    if (ierr /= 0 .or. nread /= ntotal) call exit_MPI("The local weight is not correctly defined!", myrank, 0)
    allocate(lw(ntotal))
    read(u, *, iostat=ierr) lw
    if (ierr /= 0) call exit_MPI("Problem with reading the local weight!", myrank, ierr)
    close(u)
    nsmaller = get_nsmaller(par%nelements, myrank, nbproc)
    do i = 1, par%nelements
      if (lw(nsmaller + i) /= 0.d0) then
        column_weight(i) = column_weight(i) / lw(nsmaller + i)
      else
        column_weight(i) = 0.d0
      endif
    enddo
  end subroutine apply_local_depth_weighting

end module weights_gravmag
