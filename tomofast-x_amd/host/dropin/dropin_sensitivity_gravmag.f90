!=========================================================================================================
! DROP-IN module `sensitivity_gravmag` for the UNMODIFIED Tomofast-x sources (build recipe: INTEGRATION.md 0).
! Replaces src/forward/gravmag/sensitivity_gravmag.F90 (and with it gravity_field.f90 / magnetic_field.f90, which only it uses):
! the six public procedures (:44-49) with the reference's argument lists over the reference's own types.  The kernel is calculated,
! wavelet-transformed, thresholded and compacted on the GPU (tfx_build_kernel of libtfx.so) and STAYS there; the SENSIT files are
! still written in the reference's format, and read_sensitivity_kernel registers the device-resident kernel in the drop-in
! t_sparse_matrix instead of filling a host CSR.  The repository's own code.
!=========================================================================================================
module sensitivity_gravmag
  use iso_c_binding
  use global_typedefs
  use mpi_tools, only: exit_MPI
  use file_utils, only: create_directory
  use parameters_gravmag
  use parameters_grav
  use parameters_mag
  use grid
  use data_gravmag
  use sparse_matrix
  use parallel_tools, only: get_full_array, scatter_full_array, get_nsmaller
  use dropin_gravmag_convert
  use tfx_reference_api, only: api_par_base => t_parameters_base, api_grid => t_grid, api_data => t_data, &
                               api_matrix => t_sparse_matrix, &
                               api_calculate_and_write_sensit => calculate_and_write_sensit, &
                               api_calculate_new_partitioning => calculate_new_partitioning, &
                               api_read_sensitivity_kernel => read_sensitivity_kernel, tfx_api_kernel_slot
  implicit none
  private

  public :: calculate_and_write_sensit
  public :: read_sensitivity_kernel
  public :: read_sensitivity_metadata
  public :: calculate_new_partitioning
  public :: write_depth_weight
  public :: read_depth_weight

  character(len=4), parameter :: SUFFIX(2) = ["grav", "magn"]

contains

  ! sensitivity_gravmag.F90:82-410
  subroutine calculate_and_write_sensit(par, grid_full, data, column_weight, memory, myrank, nbproc)
    class(t_parameters_base), intent(in) :: par
    type(t_grid), intent(in) :: grid_full
    type(t_data), intent(in) :: data
    real(kind=CUSTOM_REAL), intent(in) :: column_weight(par%nelements)
    integer, intent(in) :: myrank, nbproc
    real(kind=CUSTOM_REAL), intent(out) :: memory
    class(api_par_base), allocatable :: apar
    type(api_grid) :: agrid
    type(api_data) :: adata
    real(kind=CUSTOM_REAL), allocatable :: cw_full(:)

    call to_api_parameters(par, apar)
    apar%sensit_read = 0                                   ! (sensit_read = 2 also calculates the kernel, problem_joint_gravmag.F90:195)
    call to_api_grid(grid_full, agrid)
    call to_api_data(data, adata)
    ! the column weight of ALL cells on every rank (:203-205: the reference gathers it the same way)
    allocate(cw_full(par%nx * par%ny * par%nz))
    call get_full_array(column_weight, par%nelements, cw_full, .true., myrank, nbproc)
    if (myrank == 0) call create_directory(trim(path_output)//'/SENSIT')
    call api_calculate_and_write_sensit(apar, agrid, adata, cw_full, memory, myrank, nbproc)
  end subroutine calculate_and_write_sensit

  ! :573-640
  subroutine calculate_new_partitioning(par, nnz, nelements_at_cpu, problem_type, myrank, nbproc)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: problem_type
    integer, intent(in) :: myrank, nbproc
    integer(kind=8), intent(out) :: nnz
    integer, intent(out) :: nelements_at_cpu(nbproc)
    class(api_par_base), allocatable :: apar
    call to_api_parameters(par, apar)
    call api_calculate_new_partitioning(apar, nnz, nelements_at_cpu, problem_type, myrank, nbproc)
  end subroutine calculate_new_partitioning

  ! :648-883.  The kernel of this problem becomes block `problem_type` of the joint matrix: rows after the kernels loaded before it,
  ! columns from (problem_type - 1) * nmodel_components * nelements (:829-846).
  subroutine read_sensitivity_kernel(par, sensit_matrix, column_weight, problem_weight, data_weight, problem_type, myrank, nbproc, &
                                     nelements_at_cpu)
    class(t_parameters_base), intent(in) :: par
    real(kind=CUSTOM_REAL), intent(in) :: problem_weight
    real(kind=CUSTOM_REAL), intent(in) :: data_weight(par%ndata_components, par%ndata)
    integer, intent(in) :: problem_type
    integer, intent(in) :: myrank, nbproc
    integer, intent(in) :: nelements_at_cpu(nbproc)
    type(t_sparse_matrix), intent(inout) :: sensit_matrix
    real(kind=CUSTOM_REAL), intent(out) :: column_weight(par%nelements)
    class(api_par_base), allocatable :: apar
    type(api_matrix) :: amat
    logical :: loaded
    integer :: slot, nr, nc, r0, c0, other, row0, nrows, ncols

    call to_api_parameters(par, apar)
    call api_read_sensitivity_kernel(apar, amat, column_weight, problem_weight, data_weight, problem_type, myrank, nbproc, nelements_at_cpu)
    nrows = par%ndata * par%ndata_components
    ncols = par%nmodel_components * par%nelements
    row0 = 0
    other = 3 - problem_type
    call sensit_matrix%device_block(other, loaded, slot, nr, nc, r0, c0)
    if (loaded .and. other < problem_type) row0 = nr
    if (loaded .and. other > problem_type) call exit_MPI("read_sensitivity_kernel: the kernels must be loaded in problem order!", myrank, 0)
    if (amat%nl_device /= nrows .or. amat%ncolumns_device /= ncols) &
      call exit_MPI("read_sensitivity_kernel: the kernel on the device does not have the Parfile's size!", myrank, amat%nl_device)
    call sensit_matrix%register_device_kernel(problem_type, tfx_api_kernel_slot(problem_type), nrows, ncols, row0, (problem_type - 1) * ncols)
  end subroutine read_sensitivity_kernel

  ! :974-1037: the number of ranks that wrote the kernel files (second line of sensit_*_meta.txt)
  subroutine read_sensitivity_metadata(par, nbproc_sensit, problem_type, myrank)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: problem_type
    integer, intent(in) :: myrank
    integer, intent(out) :: nbproc_sensit
    character(len=256) :: filename_full, msg
    integer :: u, ierr, nxr, nyr, nzr, ndr, prec, wtype
    if (par%sensit_read /= 0) then
      filename_full = trim(par%sensit_path)//"sensit_"//SUFFIX(problem_type)//"_meta.txt"
    else
      filename_full = trim(path_output)//"/SENSIT/"//"sensit_"//SUFFIX(problem_type)//"_meta.txt"
    endif
    open(newunit=u, file=trim(filename_full), form='formatted', status='old', action='read', iostat=ierr, iomsg=msg)
    if (ierr /= 0) call exit_MPI("Error in opening the sensitivity metadata file! path="//trim(filename_full)//", iomsg="//msg, myrank, ierr)
    read(u, *) nxr, nyr, nzr, ndr
    read(u, *) nbproc_sensit, prec, wtype
    close(u)
    if (nxr /= par%nx .or. nyr /= par%ny .or. nzr /= par%nz .or. ndr /= par%ndata) &
      call exit_MPI("Sensitivity metadata file info does not match the Parfile!", myrank, 0)
  end subroutine read_sensitivity_metadata

  ! :415-464: sensit_<problem>_weight = int32 count + fp64 weights of all cells, big-endian stream
  subroutine write_depth_weight(par, column_weight, myrank, nbproc)
    class(t_parameters_base), intent(in) :: par
    real(kind=CUSTOM_REAL), intent(in) :: column_weight(par%nelements)
    integer, intent(in) :: myrank, nbproc
    real(kind=CUSTOM_REAL), allocatable :: cw_full(:)
    character(len=256) :: filename_full
    integer :: u, ntotal
    ntotal = par%nx * par%ny * par%nz
    allocate(cw_full(ntotal))
    call get_full_array(column_weight, par%nelements, cw_full, .true., myrank, nbproc)
    if (myrank /= 0) return
    call create_directory(trim(path_output)//'/SENSIT')
    filename_full = trim(path_output)//"/SENSIT/sensit_"//SUFFIX(problem_type_of(par))//"_weight"
    print *, 'Writing the depth weight to file ', trim(filename_full)
    open(newunit=u, file=trim(filename_full), form='unformatted', status='replace', action='write', access='stream', convert='big_endian')
    write(u) int(ntotal, c_int32_t)
    write(u) cw_full
    close(u)
  end subroutine write_depth_weight

  ! :887-969
  subroutine read_depth_weight(par, column_weight, myrank, nbproc)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: myrank, nbproc
    real(kind=CUSTOM_REAL), intent(out) :: column_weight(par%nelements)
    real(kind=CUSTOM_REAL), allocatable :: cw_full(:)
    character(len=256) :: filename_full, msg
    integer :: u, ierr, ntotal, nsmaller
    integer(c_int32_t) :: nread
    ntotal = par%nx * par%ny * par%nz
    if (par%sensit_read /= 0) then
      filename_full = trim(par%sensit_path)//"sensit_"//SUFFIX(problem_type_of(par))//"_weight"
    else
      filename_full = trim(path_output)//"/SENSIT/sensit_"//SUFFIX(problem_type_of(par))//"_weight"
    endif
    if (myrank == 0) print *, "Reading the depth weight file ", trim(filename_full)
    allocate(cw_full(ntotal))
    open(newunit=u, file=trim(filename_full), form='unformatted', status='old', action='read', access='stream', convert='big_endian', &
         iostat=ierr, iomsg=msg)
    if (ierr /= 0) call exit_MPI("Error in opening the depth weight file! path="//trim(filename_full)//", iomsg="//msg, myrank, ierr)
    read(u) nread
    if (nread /= ntotal) call exit_MPI("Depth weight file header does not match the Parfile!", myrank, 0)
    read(u) cw_full
    close(u)
    nsmaller = get_nsmaller(par%nelements, myrank, nbproc)
    column_weight = cw_full(nsmaller + 1:nsmaller + par%nelements)
  end subroutine read_depth_weight

end module sensitivity_gravmag
