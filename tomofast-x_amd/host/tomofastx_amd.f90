!=========================================================================================================
! tomofastx_amd - Fortran host for the MI355X path with the reference's command line and Parfile interface:
!
!     ./tomofastx_amd -p <Parfile>          (src/program_tomofastx.F90:77-80, src/parameters_init.f90:104-119)
!
! The main program reads the Parfile and calls solve_problem_joint_gravmag(gpar, mpar, ipar, myrank, nbproc) (module
! problem_joint_gravmag below), which re-states the control flow of the reference's routine of that name
! (src/problem_joint_gravmag.F90:65-613) and of joint_inversion_solve (src/inversion/joint_inverse_problem.F90:393-573) for
! gravity, magnetic or JOINT gravity + magnetic inversion (selected by the problem weights like the reference).  Every step on
! the sensitivity-kernel hot path goes through the reference's own entry points, provided over libtfx.so by module
! tfx_reference_api: calculate_depth_weight, calculate_and_write_sensit, calculate_new_partitioning, read_sensitivity_kernel,
! model%calculate_data, lsqr_solve_sensit (with the constraint rows built by add / new_row like damping%add), forward_wavelet /
! inverse_wavelet.  No other call reaches the GPU library from this file.
! What stays in Fortran is what the reference also does on the host: Parfile parsing, ASCII readers / writers in
! the reference's formats, the ADMM projection (src/inversion/admm_method.F90:70-134), residuals and costs.
!
! Supported Parfile subset: gravity (g_z, or gradiometry Gzz / full tensor with forward.data.grav.type = 2), magnetic (TMI or
! three-component data; susceptibility or magnetisation-vector model), or both jointly in one LSQR system; depth weighting
! types 1 and 2 with optional local weights; data errors; Haar / D4 compression or none; SENSIT checkpoint (write, readFromFiles
! 1 / 2); model damping (L2 or Lp, optional local weights); gradient damping; ADMM with global or per-cell bounds and weights
! (dynamic weight); prior / starting model by value or file; data from file or from a synthetic model; 1..P ranks under mpiexec.
! The solver switches to spatial unknowns (WAVELET_DOMAIN = F) by the reference's rule (joint_inverse_problem.F90:189-198).
! Joint runs may add the cross-gradient coupling constraint (forward or central differences, keepModelConstant); its rows are built
! on the host and uploaded as the general constraint matrix like the reference's matrix_cons; so are the rows of the clustering
! (Gaussian-mixture) constraint, logarithmic or normal objective, global or per-cell cluster weights.  The cross-gradient with a
! given vector field stops with a message; unknown keys only warn (parameters_init.f90:944-947).
!=========================================================================================================
module tfx_host_params
  implicit none
  integer, parameter :: dp = kind(1.d0)
  integer, parameter :: ROW_BLOCK = 2048   ! rows per row block of the device matrix (tfx_matrix_append_rows)
  logical, save :: io_rank = .true.        ! only rank 0 writes output files (the reference: `if (myrank == 0)` around every writer)

  type t_par
    character(len=256) :: path_output = 'output/test/'
    character(len=256) :: description = ''
    real(dp) :: data_units_mult(2) = 1.d0, model_units_mult(2) = 1.d0
    integer :: z_axis_dir = 1
    integer :: nx = 0, ny = 0, nz = 0
    character(len=256) :: grid_file(2) = 'None'
    integer :: ndata(2) = 0
    character(len=256) :: data_grid_file(2) = 'None'
    integer :: use_synth(2) = 0
    character(len=256) :: synth_file(2) = 'None'
    real(dp) :: mag_incl = 90.d0, mag_decl = 0.d0, mag_intensity = 50000.d0, mag_xaxis_decl = 0.d0
    integer :: dw_type = 2
    real(dp) :: dw_power(2) = (/2.d0, 3.d0/), dw_Z0(2) = 0.d0, dw_beta(2) = 1.d0
    integer :: comp_type = 0
    real(dp) :: comp_rate = 0.1d0
    integer :: prior_type = 1, start_type = 1
    real(dp) :: prior_val(2) = 0.d0, start_val(2) = 0.d0
    character(len=256) :: prior_file(2) = 'None', start_file(2) = 'None'
    integer :: nmajor = 10, nminor = 100
    real(dp) :: target_misfit = 0.d0, rmin = 1.d-13, gamma = 0.d0
    real(dp) :: alpha(2) = (/1.d-11, 1.d-8/), norm_power = 2.d0
    real(dp) :: pw(2) = (/1.d0, 0.d0/), cwm(2) = (/4.d3, 1.d0/)
    integer :: admm = 0, admm_bound_type = 1, nlithos = 1
    real(dp), allocatable :: bounds(:, :)         ! (2*nlithos, problem)
    real(dp) :: rho(2) = 1.d-7, admm_cost_thr = 1.d-4, admm_mult = 1.d0, admm_max = 1.d+10
    ! features that need the out-of-scope constraint builders
    real(dp) :: beta_grad(2) = 0.d0, w_cross = 0.d0, w_clust(2) = 0.d0
    integer :: der_type = 1, keep_const(2) = 0, vec_field_type = 0   ! inversion.crossGradient.* (parameters_init.f90:368-373)
    integer :: nclusters = 4, clust_opt = 2, clust_cons = 2          ! inversion.clustering.* (parameters_init.f90:375-381)
    character(len=256) :: mixture_file = 'NILL', cell_weights_file = 'NILL'
    integer :: apply_local_dw = 0, apply_local_damp = 0, use_error(2) = 0, sensit_read = 0, nmodel_comp = 1, ndata_comp(2) = 1
    integer :: grav_data_type = 1
    character(len=256) :: sensit_path = 'SENSIT/'        ! src/parameters_init.f90:296-297
    character(len=256) :: local_dw_file(2) = 'NILL'      ! forward.depthWeighting.{grav,magn}.file
    character(len=256) :: local_damp_file(2) = 'NILL'    ! inversion.modelDamping.{grav,magn}.file
    character(len=256) :: error_file(2) = 'NILL'         ! forward.data.{grav,magn}.errorFile (useError = 1)
    character(len=256) :: bounds_file(2) = 'NILL'        ! inversion.admm.{grav,magn}.boundsFile (boundType 2)
  end type t_par

  ! Parfile keys of the callers on either side of the hot path (constraint builders, file names, ADMM schedule): the program sets
  ! this before it calls solve_problem_joint_gravmag; what the hot path itself reads travels in gpar / mpar / ipar
  type(t_par), save :: host_par

  ! set by the program to MPI_Abort under a launcher, so that a failure on one rank takes the others down instead of leaving them
  ! blocked in a collective (src/utils/mpi_tools.F90:29-53: exit_MPI aborts every rank)
  abstract interface
    subroutine abort_all_ranks()
    end subroutine abort_all_ranks
  end interface
  procedure(abort_all_ranks), pointer, save :: abort_hook => null()

contains

  subroutine stop_msg(msg)
    character(len=*), intent(in) :: msg
    ! src/utils/mpi_tools.F90:29-53: banner + abort
    write(0, *) '**********************************************'
    write(0, *) 'ERROR: ', trim(msg)
    write(0, *) '**********************************************'
    print *, 'ERROR: ', trim(msg)
    flush(6)
    if (associated(abort_hook)) call abort_hook()
    stop 1
  end subroutine stop_msg

  subroutine read_parfile(path, par)
    character(len=*), intent(in) :: path
    type(t_par), intent(inout) :: par
    character(len=512) :: line, key, val, bounds_text(2)
    integer :: ios, eq, u
    bounds_text = ''
    open(newunit=u, file=trim(path), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Parfile "'//trim(path)//'" cannot be opened!')
    do
      read(u, '(A)', iostat=ios) line
      if (ios /= 0) exit
      if (line(1:1) == '#') cycle
      eq = index(line, '=')
      if (eq <= 1) cycle
      key = adjustl(line(:eq - 1))
      val = adjustl(line(eq + 1:))
      if (len_trim(key) == 0) cycle
      select case (trim(key))
      case ('global.outputFolderPath');            par%path_output = trim(val)
      case ('global.description');                 par%description = trim(val)
      case ('global.grav.dataUnitsMultiplier');    read(val, *) par%data_units_mult(1)
      case ('global.magn.dataUnitsMultiplier');    read(val, *) par%data_units_mult(2)
      case ('global.grav.modelUnitsMultiplier');   read(val, *) par%model_units_mult(1)
      case ('global.magn.modelUnitsMultiplier');   read(val, *) par%model_units_mult(2)
      case ('global.zAxisDirection');              read(val, *) par%z_axis_dir
      case ('modelGrid.size');                     read(val, *) par%nx, par%ny, par%nz
      case ('modelGrid.grav.file');                par%grid_file(1) = trim(val)
      case ('modelGrid.magn.file');                par%grid_file(2) = trim(val)
      case ('modelGrid.magn.nModelComponents');    read(val, *) par%nmodel_comp
      case ('forward.data.grav.nData');            read(val, *) par%ndata(1)
      case ('forward.data.magn.nData');            read(val, *) par%ndata(2)
      case ('forward.data.grav.dataGridFile');     par%data_grid_file(1) = trim(val)
      case ('forward.data.magn.dataGridFile');     par%data_grid_file(2) = trim(val)
      case ('forward.data.grav.nDataComponents');  read(val, *) par%ndata_comp(1)
      case ('forward.data.magn.nDataComponents');  read(val, *) par%ndata_comp(2)
      case ('forward.data.grav.type');             read(val, *) par%grav_data_type
      case ('forward.data.grav.useError');         read(val, *) par%use_error(1)
      case ('forward.data.magn.useError');         read(val, *) par%use_error(2)
      case ('forward.data.grav.errorFile');        par%error_file(1) = trim(val)
      case ('forward.data.magn.errorFile');        par%error_file(2) = trim(val)
      case ('forward.data.grav.useSyntheticModelForDataValues'); read(val, *) par%use_synth(1)
      case ('forward.data.magn.useSyntheticModelForDataValues'); read(val, *) par%use_synth(2)
      case ('forward.data.grav.syntheticModelFile'); par%synth_file(1) = trim(val)
      case ('forward.data.magn.syntheticModelFile'); par%synth_file(2) = trim(val)
      case ('forward.magneticField.inclination');  read(val, *) par%mag_incl
      case ('forward.magneticField.declination');  read(val, *) par%mag_decl
      case ('forward.magneticField.intensity_nT'); read(val, *) par%mag_intensity
      case ('forward.magneticField.XaxisDeclination'); read(val, *) par%mag_xaxis_decl
      case ('forward.depthWeighting.type');        read(val, *) par%dw_type
      case ('forward.depthWeighting.grav.power');  read(val, *) par%dw_power(1)
      case ('forward.depthWeighting.magn.power');  read(val, *) par%dw_power(2)
      case ('forward.depthWeighting.grav.beta');   read(val, *) par%dw_beta(1)
      case ('forward.depthWeighting.magn.beta');   read(val, *) par%dw_beta(2)
      case ('forward.depthWeighting.grav.Z0');     read(val, *) par%dw_Z0(1)
      case ('forward.depthWeighting.magn.Z0');     read(val, *) par%dw_Z0(2)
      case ('forward.depthWeighting.applyLocalWeight'); read(val, *) par%apply_local_dw
      case ('forward.depthWeighting.grav.file');   par%local_dw_file(1) = trim(val)
      case ('forward.depthWeighting.magn.file');   par%local_dw_file(2) = trim(val)
      case ('inversion.modelDamping.grav.file');   par%local_damp_file(1) = trim(val)
      case ('inversion.modelDamping.magn.file');   par%local_damp_file(2) = trim(val)
      case ('sensit.readFromFiles');               read(val, *) par%sensit_read
      case ('sensit.folderPath');                  if (len_trim(val) > 0) par%sensit_path = trim(val)
      case ('forward.matrixCompression.type');     read(val, *) par%comp_type
      case ('forward.matrixCompression.rate');     read(val, *) par%comp_rate
      case ('inversion.priorModel.type');          read(val, *) par%prior_type
      case ('inversion.priorModel.grav.value');    read(val, *) par%prior_val(1)
      case ('inversion.priorModel.magn.value');    read(val, *) par%prior_val(2)
      case ('inversion.priorModel.grav.file');     par%prior_file(1) = trim(val)
      case ('inversion.priorModel.magn.file');     par%prior_file(2) = trim(val)
      case ('inversion.startingModel.type');       read(val, *) par%start_type
      case ('inversion.startingModel.grav.value'); read(val, *) par%start_val(1)
      case ('inversion.startingModel.magn.value'); read(val, *) par%start_val(2)
      case ('inversion.startingModel.grav.file');  par%start_file(1) = trim(val)
      case ('inversion.startingModel.magn.file');  par%start_file(2) = trim(val)
      case ('inversion.nMajorIterations');         read(val, *) par%nmajor
      case ('inversion.nMinorIterations');         read(val, *) par%nminor
      case ('inversion.targetMisfit');             read(val, *) par%target_misfit
      case ('inversion.minResidual');              read(val, *) par%rmin
      case ('inversion.softThresholdL1');          read(val, *) par%gamma
      case ('inversion.modelDamping.grav.weight'); read(val, *) par%alpha(1)
      case ('inversion.modelDamping.magn.weight'); read(val, *) par%alpha(2)
      case ('inversion.modelDamping.normPower');   read(val, *) par%norm_power
      case ('inversion.modelDamping.applyLocalWeight'); read(val, *) par%apply_local_damp
      case ('inversion.joint.grav.problemWeight'); read(val, *) par%pw(1)
      case ('inversion.joint.magn.problemWeight'); read(val, *) par%pw(2)
      case ('inversion.joint.grav.columnWeightMultiplier'); read(val, *) par%cwm(1)
      case ('inversion.joint.magn.columnWeightMultiplier'); read(val, *) par%cwm(2)
      case ('inversion.admm.enableADMM');          read(val, *) par%admm
      case ('inversion.admm.boundType');           read(val, *) par%admm_bound_type
      case ('inversion.admm.nLithologies');        read(val, *) par%nlithos
      case ('inversion.admm.grav.bounds');         bounds_text(1) = val      ! parsed after the whole file: they need nLithologies,
      case ('inversion.admm.magn.bounds');         bounds_text(2) = val      ! enableADMM and boundType, wherever those keys stand
      case ('inversion.admm.grav.boundsFile');     par%bounds_file(1) = trim(val)
      case ('inversion.admm.magn.boundsFile');     par%bounds_file(2) = trim(val)
      case ('inversion.admm.grav.weight');         read(val, *) par%rho(1)
      case ('inversion.admm.magn.weight');         read(val, *) par%rho(2)
      case ('inversion.admm.dataCostThreshold');   read(val, *) par%admm_cost_thr
      case ('inversion.admm.weightMultiplier');    read(val, *) par%admm_mult
      case ('inversion.admm.maxWeight');           read(val, *) par%admm_max
      case ('inversion.dampingGradient.grav.weight'); read(val, *) par%beta_grad(1)
      case ('inversion.dampingGradient.magn.weight'); read(val, *) par%beta_grad(2)
      case ('inversion.crossGradient.weight');     read(val, *) par%w_cross
      case ('inversion.crossGradient.derivativeType'); read(val, *) par%der_type
      case ('inversion.crossGradient.grav.keepModelConstant'); read(val, *) par%keep_const(1)
      case ('inversion.crossGradient.magn.keepModelConstant'); read(val, *) par%keep_const(2)
      case ('inversion.crossGradient.vectorFieldType'); read(val, *) par%vec_field_type
      case ('inversion.clustering.grav.weight');   read(val, *) par%w_clust(1)
      case ('inversion.clustering.magn.weight');   read(val, *) par%w_clust(2)
      case ('inversion.clustering.nClusters');     read(val, *) par%nclusters
      case ('inversion.clustering.mixtureFile');   par%mixture_file = trim(val)
      case ('inversion.clustering.cellWeightsFile'); par%cell_weights_file = trim(val)
      case ('inversion.clustering.optimizationType'); read(val, *) par%clust_opt
      case ('inversion.clustering.constraintsType'); read(val, *) par%clust_cons
      case ('inversion.writeModelEveryNiter', 'inversion.solver', &
            'output.paraview.grav.modelLabel', 'output.paraview.magn.modelLabel', 'inversion.priorModel.nModels')
        continue
      case default
        print *, 'WARNING: Unknown parameter name: ', trim(key)
      end select
    enddo
    close(u)
    if (par%admm > 0 .and. par%admm_bound_type == 1) then        ! parameters_init.f90:880-905
      do eq = 1, 2
        if (len_trim(bounds_text(eq)) == 0) cycle
        if (.not. allocated(par%bounds)) then
          allocate(par%bounds(2 * par%nlithos, 2))
          par%bounds = huge(1.d0)
        endif
        read(bounds_text(eq), *, iostat=ios) par%bounds(:, eq)
        if (ios /= 0) call stop_msg('Wrong number of ADMM bounds: define bounds as min1 max1 ... minN maxN (nLithologies pairs)!')
      enddo
    endif
    print *, 'Finished reading the parameter file.'
  end subroutine read_parfile

end module tfx_host_params

!=========================================================================================================
module tfx_host_io
  use tfx_host_params
  implicit none
contains

  ! Model grid file, 9-column format "X1 X2 Y1 Y2 Z1 Z2 i j k" after a header line with the cell count
  ! (src/inversion/model_IO.F90:135-241)
  subroutine read_model_grid(file, n, X1, X2, Y1, Y2, Z1, Z2)
    character(len=*), intent(in) :: file
    integer, intent(in) :: n
    real(dp), intent(out) :: X1(n), X2(n), Y1(n), Y2(n), Z1(n), Z2(n)
    integer :: u, ios, nfile, p, i, j, k
    open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the model grid file '//trim(file))
    read(u, *) nfile
    if (nfile /= n) call stop_msg('The grid is not correctly defined (nx*ny*nz differs from the file)!')
    do p = 1, n
      read(u, *, iostat=ios) X1(p), X2(p), Y1(p), Y2(p), Z1(p), Z2(p), i, j, k
      if (ios /= 0) call stop_msg('Problem while reading the model grid file!')
    enddo
    close(u)
  end subroutine read_model_grid

  ! nc values (model components) per line after a header line with the count (src/inversion/model_IO.F90:87-130)
  subroutine read_model_values(file, n, nc, val)
    character(len=*), intent(in) :: file
    integer, intent(in) :: n, nc
    real(dp), intent(out) :: val(n, nc)
    integer :: u, ios, nfile, p
    open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the model file '//trim(file))
    read(u, *) nfile
    if (nfile /= n) call stop_msg('The model size in the file differs from the grid!')
    do p = 1, n
      read(u, *, iostat=ios) val(p, :)
      if (ios /= 0) call stop_msg('Problem while reading the model file!')
    enddo
    close(u)
  end subroutine read_model_values

  ! "x y z value(1:ncomp)" per line after a header line with the count (src/forward/gravmag/data_gravmag.f90:204-239)
  subroutine read_data(file, n, ncomp, X, Y, Z, val)
    character(len=*), intent(in) :: file
    integer, intent(in) :: n, ncomp
    real(dp), intent(out) :: X(n), Y(n), Z(n), val(ncomp, n)
    integer :: u, ios, nfile, i
    open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the data file!')
    read(u, *) nfile
    if (nfile /= n) call stop_msg('The number of data in Parfile differs from the data file!')
    do i = 1, n
      read(u, *, iostat=ios) X(i), Y(i), Z(i), val(:, i)
      if (ios /= 0) call stop_msg('Problem while reading the data file! Verify the number of data components.')
    enddo
    close(u)
  end subroutine read_data

  ! one value per cell after a count line (local depth weights weights_gravmag.f90:268-309, damping weights model_IO.F90:425-476)
  subroutine read_cell_values(file, n, w, what)
    character(len=*), intent(in) :: file, what
    integer, intent(in) :: n
    real(dp), intent(out) :: w(n)
    integer :: u, ios, nfile, p
    print *, 'Reading '//what//' from file ', trim(file)
    open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the '//what//' file! path='//trim(file))
    read(u, *, iostat=ios) nfile
    if (ios /= 0 .or. nfile /= n) call stop_msg('The '//what//' are not correctly defined!')
    do p = 1, n
      read(u, *, iostat=ios) w(p)
      if (ios /= 0) call stop_msg('Problem with reading the local weight!')
    enddo
    close(u)
  end subroutine read_cell_values

  ! data errors -> data weights 1 / (units_mult * error)  (src/forward/gravmag/data_gravmag.f90:243-279)
  subroutine read_data_error(file, n, ncomp, units_mult, w)
    character(len=*), intent(in) :: file
    integer, intent(in) :: n, ncomp
    real(dp), intent(in) :: units_mult
    real(dp), intent(out) :: w(ncomp, n)
    real(dp) :: e(ncomp)
    integer :: u, ios, nfile, i
    print *, 'Reading data error from file '//trim(file)
    open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the data error file!')
    read(u, *) nfile
    if (nfile /= n) call stop_msg('The number of data in Parfile differs from the data file!')
    do i = 1, n
      read(u, *, iostat=ios) e
      if (ios /= 0) call stop_msg('Problem while reading the data error file!')
      w(:, i) = 1.d0 / (units_mult * e)
    enddo
    close(u)
  end subroutine read_data_error

  ! local bound constraints for the ADMM: "nelements nlithos", then per cell min1 max1 ... minL maxL weight
  ! (src/inversion/model_IO.F90:311-372)
  subroutine read_bound_constraints(file, n, nlithos, bnd, w)
    character(len=*), intent(in) :: file
    integer, intent(in) :: n, nlithos
    real(dp), intent(out) :: bnd(2 * nlithos, n), w(n)
    integer :: u, ios, nread, lread, p, j
    print *, 'Reading local bound constraints from file ', trim(file)
    open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the bound constraints file! path='//trim(file))
    read(u, *, iostat=ios) nread, lread
    if (ios /= 0) call stop_msg('Problem while reading the bound constraints file!')
    if (nread /= n .or. lread /= nlithos) call stop_msg('The constraints are not correctly defined!')
    do p = 1, n
      read(u, *, iostat=ios) bnd(:, p), w(p)
      if (ios /= 0) call stop_msg('Problem with reading the bound constraints!')
      do j = 1, nlithos
        if (bnd(2 * j - 1, p) > bnd(2 * j, p)) call stop_msg('Wrong admm bounds: define bounds as: min1 max1 ... minN maxN.')
      enddo
    enddo
    close(u)
  end subroutine read_bound_constraints

  subroutine make_dir(path)
    character(len=*), intent(in) :: path
    if (.not. io_rank) return
    call execute_command_line('mkdir -p "'//trim(path)//'"')       ! src/utils/file_utils.F90:31-41
  end subroutine make_dir

  ! src/forward/gravmag/data_gravmag.f90:293-336
  subroutine write_data(path_output, name, n, ncomp, X, Y, Z, val, units_mult, z_axis_dir)
    character(len=*), intent(in) :: path_output, name
    integer, intent(in) :: n, ncomp, z_axis_dir
    real(dp), intent(in) :: X(n), Y(n), Z(n), val(ncomp, n), units_mult
    integer :: u, i
    if (.not. io_rank) return
    call make_dir(trim(path_output)//'/data')
    open(newunit=u, file=trim(path_output)//'/data/'//trim(name)//'.txt', status='replace', action='write')
    write(u, *) n
    do i = 1, n
      write(u, *) X(i), Y(i), real(z_axis_dir, dp) * Z(i), val(:, i) / units_mult
    enddo
    close(u)
  end subroutine write_data

  ! src/inversion/model_IO.F90:504-539
  subroutine write_model(path_output, name, n, nc, val, units_mult)
    character(len=*), intent(in) :: path_output, name
    integer, intent(in) :: n, nc
    real(dp), intent(in) :: val(n, nc), units_mult
    integer :: u, p
    if (.not. io_rank) return
    call make_dir(trim(path_output)//'/model')
    open(newunit=u, file=trim(path_output)//'/model/'//trim(name), status='replace', action='write')
    write(u, *) n
    do p = 1, n
      write(u, *) val(p, :) / units_mult
    enddo
    close(u)
  end subroutine write_model

end module tfx_host_io

!=========================================================================================================
module problem_joint_gravmag
  use iso_c_binding
  use tfx_host_params
  use tfx_host_io
  use tfx_reference_api
  use tfx_host_mpi, only: allreduce_sum_dp, allgather_slices
  implicit none
  private
  public :: solve_problem_joint_gravmag

contains

!=========================================================================================================
! Solves gravity AND magnetism joint problem (forward + inversion): src/problem_joint_gravmag.F90:65-613, same arguments.
!=========================================================================================================
subroutine solve_problem_joint_gravmag(gpar, mpar, ipar, myrank, nbproc)
  type(t_parameters_grav), intent(inout) :: gpar
  type(t_parameters_mag), intent(inout) :: mpar
  type(t_parameters_inversion), intent(inout) :: ipar
  integer, intent(in) :: myrank, nbproc

  ! everything one problem (gravity or magnetic) owns; model vectors are component-major, data vectors d-fastest
  type t_prob
    logical :: on = .false.
    integer :: slot = 0, nd = 0, ndc = 1, nc = 1, nm = 0, ndt = 0, dtype = 1, col0 = 0, row0 = 0
    integer :: nml = 0                        ! local unknowns: nc * (cells of this rank)
    real(dp) :: pw = 0.d0, rho = 0.d0, cost_data = 0.d0, cost_model = 0.d0, cost_admm = 0.d0
    real(dp), allocatable :: X1(:), X2(:), Y1(:), Y2(:), Z1(:), Z2(:), cw(:)
    real(dp), allocatable :: Xd(:), Yd(:), Zd(:), d_meas(:), d_calc(:)
    real(dp), allocatable :: damp_w(:)                  ! local model-damping weight per cell (model%damping_weight)
    real(dp), allocatable :: dw(:)                      ! data weight (ndc, nd) = 1 / data error, or 1 (data_gravmag.f90:243-279)
    real(dp), allocatable :: m(:), m_prior(:), m_synth(:)
    real(dp), allocatable :: bnd(:, :), bnd_w(:)        ! ADMM intervals (2*nlithos, cell) and per-cell weight (model%bound_weight)
  end type t_prob

  type(t_par) :: par
  type(t_prob), target :: pr(2)
  ! the reference's objects of this routine (problem_joint_gravmag.F90:71-77): data with its grid, model with its grid, and the
  ! joint inversion's two matrices + right-hand side (joint_inverse_problem.F90:60-80)
  type(t_data) :: data(2)
  type(t_model) :: model(2)
  type(t_inversion_arrays) :: iarr(2)                                               ! :73
  type(t_joint_inversion) :: jinv                                                   ! :77
  real(dp), allocatable :: delta_model(:, :, :), cw_loc(:, :)
  integer, allocatable :: nelements_at_cpu(:)
  integer(c_int64_t) :: nnz_part
  real(dp) :: memory_fwd, memory_inv
  logical :: SOLVE_PROBLEM(2), WAVELET_DOMAIN
  integer :: line_start(2), line_end(2), param_shift(2), problem_type_part
  ! output file prefixes (src/problem_joint_gravmag.F90:340-362, :554-555): 'grav_...' and 'mag_...'
  character(len=4) :: suffix(2) = (/'grav', 'mag '/)
  integer :: ip, n, it, i, k, ucost, kadm, nprob, ntot, ndtot, c0, r0
  integer :: lc0
  logical :: spatial = .false.      ! gradient damping acts in space: LSQR unknowns are spatial, S goes through the device transform
  integer(c_int64_t), allocatable, target :: g_rowptr(:)
  integer(c_int32_t), allocatable, target :: g_cols(:)
  real(c_float), allocatable, target :: g_vals(:)
  real(dp), allocatable, target :: g_rhs(:)
  real(dp), allocatable :: clust_mix(:, :), clust_cellw(:, :), clust_max(:)   ! mixtures (6, cluster), cell weights (cluster, cell), P_max
  integer(c_int64_t) :: g_nrows, g_nnz, e8
  integer :: cb, ce, nloc                             ! this rank's cells (cb, ce], nloc = ce - cb
  integer, allocatable :: counts(:), displs(:)
  real(dp), allocatable, target :: xfull(:)
  real(dp), allocatable, target :: work(:)
  real(dp) :: s1, s2, s3, admm_costs(2)
  integer :: cc, row
  ! wall clock per phase of the run (printed at the end and written to <output>/phase_timing.json on rank 0)
  integer, parameter :: NPHASE = 9
  character(len=40), parameter :: phase_name(NPHASE) = [character(len=40) :: 'read inputs (ASCII)', 'column weights', &
    'kernel build (calculate_and_write_sensit)', 'partition + relayout (read_sensitivity_kernel)', 'constraint assembly', &
    'LSQR (lsqr_solve_sensit)', 'forward data (calculate_data)', 'model update + costs', 'write outputs']
  real(dp) :: phase_t(NPHASE), tph, t_run

  par = host_par
  phase_t = 0.d0
  t_run = wall()
  io_rank = myrank == 0
  if (myrank == 0) print *, 'Solving problem grav/mag.'
  memory_fwd = 0.d0
  memory_inv = 0.d0
  g_nrows = 0
  g_nnz = 0

  ! ---- which problems (src/problem_joint_gravmag.F90:108-112): both weights non-zero = joint inversion
  do i = 1, 2
    SOLVE_PROBLEM(i) = (ipar%problem_weight(i) /= 0.d0)
    if (myrank == 0) print *, 'SOLVE_PROBLEM(', i, ') = ', SOLVE_PROBLEM(i)
  enddo
  pr(1)%on = SOLVE_PROBLEM(1)
  pr(2)%on = SOLVE_PROBLEM(2)
  nprob = count(pr%on)
  if (nprob == 0) call stop_msg('Both problem weights are zero!')
  if (par%dw_type < 1 .or. par%dw_type > 3) call stop_msg('Not known depth weight type!')      ! weights_gravmag.f90:164
  if (par%w_cross /= 0.d0) then                                   ! structural coupling (joint_inverse_problem.F90:189-198, :529-541)
    if (par%pw(1) == 0.d0 .or. par%pw(2) == 0.d0) call stop_msg('The cross-gradient constraint needs both problems (joint inversion).')
    if (par%vec_field_type > 0) call stop_msg('Cross-gradient with a given vector field is not supported by this host.')
    if (par%nmodel_comp /= 1) call stop_msg('Cross-gradient constraints need scalar models.')
    spatial = .true.
  endif
  if (par%apply_local_damp > 0) spatial = .true.                 ! local damping weights act in space (:189-198)
  if (par%norm_power /= 2.d0) spatial = .true.                   ! Lp damping acts in space (joint_inverse_problem.F90:189-198)
  if (par%admm > 0 .and. par%admm_bound_type /= 1 .and. par%admm_bound_type /= 2) call stop_msg('Unknown inversion.admm.boundType!')
  if (par%admm_bound_type /= 1) spatial = .true.       ! local bounds / weights, whether ADMM is on or not (joint_inverse_problem.F90:189-198)
  if (par%sensit_read < 0 .or. par%sensit_read > 2) call stop_msg('sensit.readFromFiles must be 0, 1 or 2.')
  if (par%admm > 0 .and. par%admm_bound_type == 1 .and. .not. allocated(par%bounds)) call stop_msg('Global bounds are not defined!')
  n = par%nx * par%ny * par%nz
  if (n <= 0) call stop_msg('Wrong model grid size!')

  ! components (src/parameters_init.f90:187-197, src/forward/gravmag/sensitivity_gravmag.F90:193-220) and sizes
  ntot = 0
  ndtot = 0
  k = 0
  do ip = 1, 2
    if (.not. pr(ip)%on) cycle
    if (par%w_clust(ip) /= 0.d0) spatial = .true.
    if (par%beta_grad(ip) /= 0.d0) spatial = .true.              ! WAVELET_DOMAIN = .false. (joint_inverse_problem.F90:189-198)
    pr(ip)%slot = k
    k = k + 1
    pr(ip)%pw = par%pw(ip)
    pr(ip)%rho = par%rho(ip)
    pr(ip)%nd = par%ndata(ip)
    if (pr(ip)%nd <= 0) call stop_msg('Wrong number of data!')
    pr(ip)%nc = 1
    if (ip == 2) pr(ip)%nc = par%nmodel_comp
    if (par%nmodel_comp > 1 .and. pr(1)%on) call stop_msg('For the magnetisation inversion the gravity problem should be disabled!')
    pr(ip)%ndc = par%ndata_comp(ip)
    pr(ip)%dtype = 1
    if (ip == 1) pr(ip)%dtype = par%grav_data_type
    if (ip == 1 .and. pr(ip)%dtype == 1 .and. pr(ip)%ndc /= 1) call stop_msg('Gravity data (type 1) has one data component!')
    if (ip == 1 .and. pr(ip)%dtype == 2 .and. pr(ip)%ndc /= 1 .and. pr(ip)%ndc /= 6) &
      call stop_msg('Wrong number of gravity gradiometry data components!')
    if (ip == 1 .and. pr(ip)%dtype /= 1 .and. pr(ip)%dtype /= 2) call stop_msg('Unknown gravity data type!')
    if (ip == 2 .and. .not. ((pr(ip)%nc == 1 .or. pr(ip)%nc == 3) .and. (pr(ip)%ndc == 1 .or. pr(ip)%ndc == 3))) &
      call stop_msg('Wrong number of components in magnetic_field_magprism!')
    pr(ip)%nm = n * pr(ip)%nc            ! m((k-1)*n + cell) = model%val(cell, k)
    pr(ip)%ndt = pr(ip)%nd * pr(ip)%ndc  ! d((i-1)*ndc + d) = data(d, i)
    pr(ip)%col0 = ntot                   ! param_shift / line_start of the joint system (joint_inverse_problem.F90:712-739)
    pr(ip)%row0 = ndtot
    ntot = ntot + pr(ip)%nm
    ndtot = ndtot + pr(ip)%ndt
    allocate(pr(ip)%X1(n), pr(ip)%X2(n), pr(ip)%Y1(n), pr(ip)%Y2(n), pr(ip)%Z1(n), pr(ip)%Z2(n), pr(ip)%cw(n))
    allocate(pr(ip)%Xd(pr(ip)%nd), pr(ip)%Yd(pr(ip)%nd), pr(ip)%Zd(pr(ip)%nd), pr(ip)%d_meas(pr(ip)%ndt), pr(ip)%d_calc(pr(ip)%ndt))
    allocate(pr(ip)%dw(pr(ip)%ndt))
    pr(ip)%dw = 1.d0
    allocate(pr(ip)%m(pr(ip)%nm), pr(ip)%m_prior(pr(ip)%nm), pr(ip)%m_synth(pr(ip)%nm))
    allocate(pr(ip)%damp_w(n))
    pr(ip)%damp_w = 1.d0
    if (par%apply_local_damp > 0) call read_cell_values(par%local_damp_file(ip), n, pr(ip)%damp_w, 'model damping weights')
    if (par%admm > 0) then                                           ! set_model_bounds, src/inversion/model_IO.F90:273-305
      allocate(pr(ip)%bnd(2 * par%nlithos, n), pr(ip)%bnd_w(n))
      pr(ip)%bnd_w = 1.d0
      if (par%admm_bound_type == 1) then
        do i = 1, n
          pr(ip)%bnd(:, i) = par%bounds(:, ip)
        enddo
      else
        call read_bound_constraints(par%bounds_file(ip), n, par%nlithos, pr(ip)%bnd, pr(ip)%bnd_w)
      endif
      pr(ip)%bnd = pr(ip)%bnd * par%model_units_mult(ip)
    endif
  enddo
  allocate(xfull(ntot), work(ntot))
  if (nprob == 2 .and. myrank == 0) print *, 'JOINT inversion: two sensitivity kernels in one system.'
  WAVELET_DOMAIN = .not. spatial
  if (myrank == 0) print *, 'WAVELET_DOMAIN =', WAVELET_DOMAIN

  ! (I) MODEL GRID, (II) DATA (:135-162): the reference's objects, filled from the files
  if (myrank == 0) print *, '(I) MODEL GRID ALLOCATION.'
  tph = wall()
  do ip = 1, 2
    if (.not. pr(ip)%on) cycle
    call model(ip)%grid_full%allocate(ipar%nx, ipar%ny, ipar%nz, par%z_axis_dir, myrank)
    call read_model_grid(par%grid_file(ip), n, model(ip)%grid_full%X1, model(ip)%grid_full%X2, model(ip)%grid_full%Y1, &
                         model(ip)%grid_full%Y2, model(ip)%grid_full%Z1, model(ip)%grid_full%Z2)
    pr(ip)%X1 = model(ip)%grid_full%X1; pr(ip)%X2 = model(ip)%grid_full%X2       ! (the constraint builders read the spacings)
    pr(ip)%Y1 = model(ip)%grid_full%Y1; pr(ip)%Y2 = model(ip)%grid_full%Y2
    pr(ip)%Z1 = model(ip)%grid_full%Z1; pr(ip)%Z2 = model(ip)%grid_full%Z2
  enddo
  if (myrank == 0) print *, '(II) DATA ALLOCATION.'
  do ip = 1, 2
    if (.not. pr(ip)%on) cycle
    call data(ip)%initialize(pr(ip)%nd, pr(ip)%ndc, par%data_units_mult(ip), par%z_axis_dir, myrank)
    call read_data(par%data_grid_file(ip), pr(ip)%nd, pr(ip)%ndc, data(ip)%X, data(ip)%Y, data(ip)%Z, data(ip)%val_meas)
    data(ip)%val_meas = data(ip)%val_meas * par%data_units_mult(ip)
    if (par%use_error(ip) == 1) call read_data_error(par%error_file(ip), pr(ip)%nd, pr(ip)%ndc, par%data_units_mult(ip), data(ip)%weight)
    pr(ip)%Xd = data(ip)%X; pr(ip)%Yd = data(ip)%Y; pr(ip)%Zd = data(ip)%Z
    pr(ip)%d_meas = reshape(data(ip)%val_meas, (/pr(ip)%ndt/))
    pr(ip)%dw = reshape(data(ip)%weight, (/pr(ip)%ndt/))
  enddo
  call lap(1, tph)

  ! (III) SENSITIVITY MATRIX CALCULATION (:164-215)
  if (myrank == 0) print *, '(III) SENSITIVITY MATRIX CALCULATION.'
  tph = wall()
  if (gpar%sensit_read == 0) then
    do ip = 1, 2
      if (.not. pr(ip)%on) cycle
      if (ip == 1) then
        call calculate_depth_weight(gpar, iarr(ip), model(ip)%grid_full, data(ip), myrank, nbproc)          ! :174-175
      else
        call calculate_depth_weight(mpar, iarr(ip), model(ip)%grid_full, data(ip), myrank, nbproc)
      endif
      iarr(ip)%column_weight = ipar%column_weight_multiplier(ip) * iarr(ip)%column_weight                  ! :178-179
      pr(ip)%cw = iarr(ip)%column_weight
      call apply_local_depth_weighting(ip)                                         ! :181-182
    enddo
  endif
  call lap(2, tph)
  tph = wall()
  if (gpar%sensit_read == 0 .or. gpar%sensit_read == 2) then
    do ip = 1, 2
      if (.not. pr(ip)%on) cycle
      if (gpar%sensit_read == 2) call read_depth_weight(ip)                         ! :189-193
      if (ip == 1) then
        call calculate_and_write_sensit(gpar, model(ip)%grid_full, data(ip), pr(ip)%cw, memory_fwd, myrank, nbproc)
      else
        call calculate_and_write_sensit(mpar, model(ip)%grid_full, data(ip), pr(ip)%cw, memory_fwd, myrank, nbproc)
      endif
    enddo
  endif
  call lap(3, tph)
  tph = wall()
  ! new partitioning for the load balancing (:205-221)
  allocate(nelements_at_cpu(nbproc), counts(nbproc), displs(nbproc))
  problem_type_part = merge(3, merge(1, 2, pr(1)%on), nprob == 2)
  if (problem_type_part == 2) then
    call calculate_new_partitioning(mpar, nnz_part, nelements_at_cpu, problem_type_part, myrank, nbproc)
  else
    call calculate_new_partitioning(gpar, nnz_part, nelements_at_cpu, problem_type_part, myrank, nbproc)
  endif
  gpar%nelements = nelements_at_cpu(myrank + 1)
  mpar%nelements = nelements_at_cpu(myrank + 1)
  ipar%nelements = nelements_at_cpu(myrank + 1)
  cb = sum(nelements_at_cpu(1:myrank))
  nloc = nelements_at_cpu(myrank + 1)
  ce = cb + nloc
  counts = nelements_at_cpu
  displs(1) = 0
  do i = 2, nbproc
    displs(i) = displs(i - 1) + counts(i - 1)
  enddo
  do ip = 1, 2
    if (pr(ip)%on) pr(ip)%nml = pr(ip)%nc * nloc
  enddo
  ! (IV) MATRIX ALLOCATION + READING THE SENSITIVITY KERNEL (:227-248)
  if (myrank == 0) print *, '(IV) MATRIX ALLOCATION.'
  allocate(cw_loc(nloc, 2))
  call jinv%initialize(ipar, nnz_part, myrank)                                       ! :236
  if (jinv%WAVELET_DOMAIN .neqv. WAVELET_DOMAIN) call stop_msg('WAVELET_DOMAIN of the joint inversion differs from the Parfile rule!')
  do ip = 1, 2
    if (.not. pr(ip)%on) cycle
    call iarr(ip)%reallocate_aux(ipar%nelements, pr(ip)%nd, pr(ip)%ndc, myrank)     ! :221-222
    if (ip == 1) then
      call read_sensitivity_kernel(gpar, jinv%matrix_sensit, iarr(ip)%column_weight, ipar%problem_weight(ip), data(ip)%weight, ip, &
                                   myrank, nbproc, nelements_at_cpu)                  ! :241-246
    else
      call read_sensitivity_kernel(mpar, jinv%matrix_sensit, iarr(ip)%column_weight, ipar%problem_weight(ip), data(ip)%weight, ip, &
                                   myrank, nbproc, nelements_at_cpu)
    endif
    cw_loc(:, ip) = iarr(ip)%column_weight
    if (gpar%sensit_read == 1) then                    ! the full weight comes from the SENSIT folder (:920-970)
      call read_depth_weight(ip)
    endif
    call model(ip)%initialize(nloc, pr(ip)%nc, n, myrank)
    model(ip)%grid_full%nx = ipar%nx; model(ip)%grid_full%ny = ipar%ny; model(ip)%grid_full%nz = ipar%nz
  enddo
  call jinv%matrix_sensit%finalize(myrank)                                          ! :248
  call jinv%initialize2(ipar, iarr, model, myrank, nbproc)                           ! :263
  call jinv%calculate_matrix_partitioning(ipar, line_start, line_end, param_shift)   ! :327
  allocate(delta_model(ipar%nelements, ipar%nmodel_components, 2))
  call lap(4, tph)

  do ip = 1, 2
    if (.not. pr(ip)%on) cycle
    ! ---- data from the synthetic model (:318-345)
    if (par%use_synth(ip) > 0) then
      call read_model_values(par%synth_file(ip), n, pr(ip)%nc, pr(ip)%m_synth)
      pr(ip)%m_synth = pr(ip)%m_synth * par%model_units_mult(ip)
      call calculate_data(ip, pr(ip)%m_synth, pr(ip)%d_calc)
      pr(ip)%d_meas = pr(ip)%d_calc
      call write_data(par%path_output, trim(suffix(ip))//'_synthetic', pr(ip)%nd, pr(ip)%ndc, pr(ip)%Xd, pr(ip)%Yd, pr(ip)%Zd, &
                      pr(ip)%d_calc, par%data_units_mult(ip), par%z_axis_dir)
    endif
    call write_data(par%path_output, trim(suffix(ip))//'_observed', pr(ip)%nd, pr(ip)%ndc, pr(ip)%Xd, pr(ip)%Yd, pr(ip)%Zd, &
                    pr(ip)%d_meas, par%data_units_mult(ip), par%z_axis_dir)

    ! ---- prior and starting models (:350-441)
    if (par%prior_type == 1) then
      pr(ip)%m_prior = par%prior_val(ip)
    else
      call read_model_values(par%prior_file(ip), n, pr(ip)%nc, pr(ip)%m_prior)
    endif
    pr(ip)%m_prior = pr(ip)%m_prior * par%model_units_mult(ip)
    if (par%start_type == 1) then
      pr(ip)%m = par%start_val(ip)
    else
      call read_model_values(par%start_file(ip), n, pr(ip)%nc, pr(ip)%m)
    endif
    pr(ip)%m = pr(ip)%m * par%model_units_mult(ip)
    call calculate_data(ip, pr(ip)%m, pr(ip)%d_calc)
    call write_data(par%path_output, trim(suffix(ip))//'_starting', pr(ip)%nd, pr(ip)%ndc, pr(ip)%Xd, pr(ip)%Yd, pr(ip)%Zd, &
                    pr(ip)%d_calc, par%data_units_mult(ip), par%z_axis_dir)

    ! ---- this rank's cells of the prior, the local weights and the ADMM intervals: what jinv%solve reads from the model object
    ! (model.F90:33-70: val_prior, damping_weight, min_bound / max_bound / bound_weight)
    do k = 1, pr(ip)%nc
      model(ip)%val_prior(:, k) = pr(ip)%m_prior((k - 1) * n + cb + 1:(k - 1) * n + ce)
    enddo
    model(ip)%damping_weight = pr(ip)%damp_w(cb + 1:ce)
    if (par%admm > 0) then
      call model(ip)%allocate_bound_arrays(par%nlithos, myrank)
      do i = 1, nloc
        do k = 1, par%nlithos
          model(ip)%min_bound(k, i) = pr(ip)%bnd(2 * k - 1, cb + i)
          model(ip)%max_bound(k, i) = pr(ip)%bnd(2 * k, cb + i)
        enddo
      enddo
      model(ip)%bound_weight = pr(ip)%bnd_w(cb + 1:ce)
    endif

    ! ---- costs (:443-470)
    call model_cost(ip, pr(ip)%cost_model)
    pr(ip)%cost_data = norm2(pr(ip)%d_calc - pr(ip)%d_meas) / norm2(pr(ip)%d_meas)
    pr(ip)%cost_admm = 0.d0
  enddo
  call make_dir(par%path_output)
  if (io_rank) then
    open(newunit=ucost, file=trim(par%path_output)//'/costs.txt', status='replace', action='write')
  else
    open(newunit=ucost, status='scratch', action='readwrite')
  endif
  write(ucost, '(A)') '# 1:iteration, then per active problem: data_cost, model_cost, ADMM_cost, ADMM_weight'

  ! ---- (V) major inversion loop (:473-547)
  do it = 1, par%nmajor
    if (stop_file_exists()) then
      print *, 'Stop file found! Exiting the loop.'
      exit
    endif
    print *, '======================================================='
    print *, 'Iteration =', it
    print *, '======================================================='
    ! ---- general constraint rows first (their count sizes the system): gradient damping, cross-gradient, clustering
    tph = wall()
    g_nrows = 0
    if (spatial) then
      call build_gradient_damping()
      if (par%w_cross /= 0.d0) call build_cross_gradient()
      if (any(par%w_clust /= 0.d0)) call build_clustering()
    endif
    ! ---- the builders' rows go to the joint inversion, which assembles the rest of the system (damping and ADMM blocks, right-hand
    ! side) and solves it on the GPU: joint_inversion_solve, joint_inverse_problem.F90:393-573
    if (g_nrows > 0) then
      call jinv%set_general_rows(g_nrows, g_rowptr, g_cols, g_vals, g_rhs)
    else
      call jinv%set_general_rows(0_c_int64_t)
    endif
    do ip = 1, 2
      if (.not. pr(ip)%on) cycle
      ! residuals (:486-491; calculate_residuals: data weight * (measured - calculated)) and the current model of this rank's cells
      iarr(ip)%residuals = reshape(pr(ip)%dw * (pr(ip)%d_meas - pr(ip)%d_calc), (/pr(ip)%ndc, pr(ip)%nd/))
      do k = 1, pr(ip)%nc
        model(ip)%val(:, k) = pr(ip)%m((k - 1) * n + cb + 1:(k - 1) * n + ce)
      enddo
      ipar%rho_ADMM(ip) = pr(ip)%rho                                 ! (dynamic ADMM weight, :618-638)
    enddo
    call lap(5, tph)
    tph = wall()
    if (it > 1) call jinv%reset(myrank)                              ! :494
    call jinv%solve(ipar, iarr, model, delta_model, memory_inv, myrank, nbproc)        ! :497
    call lap(6, tph)
    tph = wall()
    call write_costs(it - 1)                                       ! :519-528 (costs of the previous iteration)
    do ip = 1, 2
      if (.not. pr(ip)%on) cycle
      if (par%admm > 0) then
        admm_costs = jinv%get_admm_cost()                          ! :520-528
        pr(ip)%cost_admm = admm_costs(ip)
      endif
      call model(ip)%update(delta_model(:, 1:pr(ip)%nc, ip))         ! :500
      do k = 1, pr(ip)%nc                                          ! the full model from its local parts (model_update_full)
        call get_full_array(model(ip)%val(:, k), nloc, pr(ip)%m((k - 1) * n + 1:k * n), myrank, nbproc)
      enddo
      call calculate_data(ip, pr(ip)%m, pr(ip)%d_calc)             ! :513
      call model_cost(ip, pr(ip)%cost_model)
      pr(ip)%cost_data = norm2(pr(ip)%d_calc - pr(ip)%d_meas) / norm2(pr(ip)%d_meas)   ! data_gravmag.f90:123-129
      print *, 'data cost (new) =', pr(ip)%cost_data
      if (par%admm > 0 .and. pr(ip)%cost_data < par%admm_cost_thr .and. pr(ip)%rho < par%admm_max .and. par%admm_mult /= 1.d0) then
        pr(ip)%rho = par%admm_mult * pr(ip)%rho                    ! :618-638
        print *, 'Increased the ADMM weight to:', pr(ip)%rho
      endif
    enddo
    call lap(8, tph)
  enddo
  call write_costs(par%nmajor)
  close(ucost)
  tph = wall()

  ! ---- outputs (:552-600)
  do ip = 1, 2
    if (.not. pr(ip)%on) cycle
    call write_data(par%path_output, trim(suffix(ip))//'_final', pr(ip)%nd, pr(ip)%ndc, pr(ip)%Xd, pr(ip)%Yd, pr(ip)%Zd, &
                    pr(ip)%d_calc, par%data_units_mult(ip), par%z_axis_dir)
    call write_model(par%path_output, trim(suffix(ip))//'_final_model_full.txt', n, pr(ip)%nc, pr(ip)%m, par%model_units_mult(ip))
    print *, 'model min / max =', minval(pr(ip)%m), maxval(pr(ip)%m)
  enddo
  call lap(9, tph)
  call report_phases()
  if (myrank == 0) print *, 'MEMORY USED (device matrix) [GB] =', memory_fwd
  call tfx_api_finalize()

contains

  real(dp) function wall()
    integer(c_int64_t) :: c, rate
    call system_clock(c, rate)
    wall = real(c, dp) / real(rate, dp)
  end function wall

  subroutine lap(idx, t0)                       ! phase idx += time since t0
    integer, intent(in) :: idx
    real(dp), intent(in) :: t0
    phase_t(idx) = phase_t(idx) + (wall() - t0)
  end subroutine lap

  ! (calculate_data runs inside other phases: its time is reported on its own line AND stays inside the enclosing phase)
  subroutine report_phases()
    integer :: q, u
    real(dp) :: total
    if (myrank /= 0) return
    total = wall() - t_run
    print *, 'PHASE TIMING [s] (rank 0 wall clock):'
    do q = 1, NPHASE
      print '(a,a44,f12.3)', '   ', phase_name(q), phase_t(q)
    enddo
    print '(a,a44,f12.3)', '   ', 'whole run', total
    open(newunit=u, file=trim(par%path_output)//'/phase_timing.json', status='replace', action='write')
    write(u, '(a)') '{'
    do q = 1, NPHASE
      write(u, '(a,a,a,es14.6,a)') '  "', trim(phase_name(q)), '": ', phase_t(q), ','
    enddo
    write(u, '(a,i0,a)') '  "ranks": ', nbproc, ','
    write(u, '(a,i0,a)') '  "cells": ', n, ','
    write(u, '(a,i0,a)') '  "major_iterations": ', par%nmajor, ','
    write(u, '(a,es14.6)') '  "whole run": ', total
    write(u, '(a)') '}'
    close(u)
  end subroutine report_phases

  subroutine write_costs(iter)
    integer, intent(in) :: iter
    integer :: jp
    write(ucost, '(I8)', advance='no') iter
    do jp = 1, 2
      if (pr(jp)%on) write(ucost, '(4(1X,ES24.16))', advance='no') pr(jp)%cost_data, pr(jp)%cost_model, pr(jp)%cost_admm, pr(jp)%rho
    enddo
    write(ucost, *)
    flush(ucost)
  end subroutine write_costs

  ! damping_gradient%add for every active problem, component and direction (src/inversion/damping_gradient.F90:94-205,
  ! joint_inverse_problem.F90:466-488): forward differences (gradient.F90:77-81) over the structured grid (grid.F90:371-391);
  ! 3 N rows per component, two entries each except in the last layer of the direction; columns ascending for the upload
  subroutine build_gradient_damping()
    integer :: jp, kc, dir, i, j, kk, p, me, nb
    integer(c_int64_t) :: row, e, ec
    real(dp) :: delta, gval, coef
    g_nrows = 0
    do jp = 1, 2
      if (pr(jp)%on .and. par%beta_grad(jp) /= 0.d0) g_nrows = g_nrows + 3_c_int64_t * n * pr(jp)%nc
    enddo
    e = 0
    if (par%w_cross /= 0.d0) e = 3_c_int64_t * n                  ! the cross-gradient rows follow (up to 8 entries each),
    ec = 0
    if (any(par%w_clust /= 0.d0)) ec = 2_c_int64_t * n            ! then the clustering rows (one entry each)
    if (.not. allocated(g_rowptr)) then
      allocate(g_rowptr(g_nrows + e + ec + 1), g_cols(2 * g_nrows + 8 * e + ec), g_vals(2 * g_nrows + 8 * e + ec), g_rhs(g_nrows + e + ec))
      g_rowptr(1) = 0
    endif
    g_nnz = 0
    if (g_nrows == 0) return
    row = 0
    e = 0
    g_rowptr(1) = 0
    do jp = 1, 2
      if (.not. (pr(jp)%on .and. par%beta_grad(jp) /= 0.d0)) cycle
      coef = pr(jp)%pw * par%beta_grad(jp)
      do kc = 1, pr(jp)%nc
        do dir = 1, 3
          p = 0
          do kk = 1, par%nz
            do j = 1, par%ny
              do i = 1, par%nx
                p = p + 1
                row = row + 1
                g_rhs(row) = 0.d0
                me = p
                nb = 0
                if (dir == 1 .and. i /= par%nx) then
                  nb = p + 1
                  delta = abs(pr(jp)%X2(i) - pr(jp)%X1(i))                       ! dX(i): cell (i, 1, 1)
                else if (dir == 2 .and. j /= par%ny) then
                  nb = p + par%nx
                  delta = abs(pr(jp)%Y2((j - 1) * par%nx + 1) - pr(jp)%Y1((j - 1) * par%nx + 1))
                else if (dir == 3 .and. kk /= par%nz) then
                  nb = p + par%nx * par%ny
                  delta = abs(pr(jp)%Z2((kk - 1) * par%nx * par%ny + 1) - pr(jp)%Z1((kk - 1) * par%nx * par%ny + 1))
                endif
                if (nb > 0) then
                  gval = (pr(jp)%m((kc - 1) * n + nb) - pr(jp)%m((kc - 1) * n + me)) / delta
                  if (local_column(jp, kc, me) > 0) then           ! rows replicated, columns of this rank only
                    e = e + 1
                    g_cols(e) = local_column(jp, kc, me)
                    g_vals(e) = real(-(1.d0 / delta) * coef * pr(jp)%cw(me), c_float)
                  endif
                  if (local_column(jp, kc, nb) > 0) then
                    e = e + 1
                    g_cols(e) = local_column(jp, kc, nb)
                    g_vals(e) = real((1.d0 / delta) * coef * pr(jp)%cw(nb), c_float)
                  endif
                  g_rhs(row) = -coef * gval
                endif
                g_rowptr(row + 1) = e
              enddo
            enddo
          enddo
        enddo
      enddo
    enddo
    g_nnz = e
  end subroutine build_gradient_damping

  ! v / column_weight with the reference's zero guard (damping.F90:129-135)
  ! cross_gradient_calculate (src/inversion/cross_gradient.F90:220-391): 3 rows per cell, tau = grad m1 x grad m2, over the
  ! columns of both models; appended after the gradient-damping rows.  Forward differences (or central with forward / backward
  ! on the boundary layers, :255-285); the derivative tables follow calculate_tau (:457-577) and calculate_tau_backward (:675-743).
  subroutine build_cross_gradient()
    integer :: i, j, kk, p, comp, t, ne, scheme, a, b, cmin
    integer(c_int64_t) :: row, e
    real(dp) :: g1(3), g2(3), st(3), tau(3), cost(3), d1(4, 3), d2(4, 3)
    integer :: cell(4, 3), ecol(8)
    real(c_float) :: eval(8), f
    logical :: lft, rgt
    row = g_nrows
    e = g_nnz
    cost = 0.d0
    p = 0
    do kk = 1, par%nz
      do j = 1, par%ny
        do i = 1, par%nx
          p = p + 1
          lft = (i == 1 .or. j == 1 .or. kk == 1)
          rgt = (i == par%nx .or. j == par%ny .or. kk == par%nz)
          scheme = 0                                              ! 0 none, 1 forward, 2 central, 3 backward
          if (lft .and. rgt) then
            scheme = 0
          else if (rgt) then
            scheme = 3
          else if (par%der_type /= 1 .and. .not. lft) then
            scheme = 2
          else
            scheme = 1
          endif
          tau = 0.d0
          ne = 0
          if (scheme /= 0) then
            st = (/ abs(pr(1)%X2(i) - pr(1)%X1(i)), abs(pr(1)%Y2((j - 1) * par%nx + 1) - pr(1)%Y1((j - 1) * par%nx + 1)), &
                    abs(pr(1)%Z2((kk - 1) * par%nx * par%ny + 1) - pr(1)%Z1((kk - 1) * par%nx * par%ny + 1)) /)
            call cell_gradient(pr(1)%m, i, j, kk, scheme, st, g1)
            call cell_gradient(pr(2)%m, i, j, kk, scheme, st, g2)
            if (scheme == 2) st = 2.d0 * st
            tau = (/ g1(2) * g2(3) - g1(3) * g2(2), g1(3) * g2(1) - g1(1) * g2(3), g1(1) * g2(2) - g1(2) * g2(1) /)
            a = 1
            if (scheme == 3) a = -1                               ! neighbours on the + side (forward, central) or - side (backward)
            ! entries 1, 2: the two neighbours of each component; 3: the cell itself (one-sided) or the opposite neighbours (central)
            cell(1, 1) = p + a * par%nx;            cell(2, 1) = p + a * par%nx * par%ny
            cell(1, 2) = p + a;                     cell(2, 2) = p + a * par%nx * par%ny
            cell(1, 3) = p + a;                     cell(2, 3) = p + a * par%nx
            d1(1, 1) = g2(3) / st(2);   d2(1, 1) = -g1(3) / st(2);  d1(2, 1) = -g2(2) / st(3);  d2(2, 1) = g1(2) / st(3)
            d1(1, 2) = -g2(3) / st(1);  d2(1, 2) = g1(3) / st(1);   d1(2, 2) = g2(1) / st(3);   d2(2, 2) = -g1(1) / st(3)
            d1(1, 3) = g2(2) / st(1);   d2(1, 3) = -g1(2) / st(1);  d1(2, 3) = -g2(1) / st(2);  d2(2, 3) = g1(1) / st(2)
            if (scheme == 2) then
              ne = 4
              cell(3, 1) = p - par%nx;  cell(4, 1) = p - par%nx * par%ny
              cell(3, 2) = p - 1;       cell(4, 2) = p - par%nx * par%ny
              cell(3, 3) = p - 1;       cell(4, 3) = p - par%nx
              d1(3:4, :) = -d1(1:2, :)
              d2(3:4, :) = -d2(1:2, :)
            else
              ne = 3
              cell(3, :) = p
              d1(3, 1) = g2(3) / st(2) - g2(2) / st(3);  d2(3, 1) = g1(2) / st(3) - g1(3) / st(2)
              d1(3, 2) = g2(1) / st(3) - g2(3) / st(1);  d2(3, 2) = g1(3) / st(1) - g1(1) / st(3)
              d1(3, 3) = g2(2) / st(1) - g2(1) / st(2);  d2(3, 3) = g1(1) / st(2) - g1(2) / st(1)
              if (scheme == 1) then
                d1(3, :) = -d1(3, :)
                d2(3, :) = -d2(3, :)
              else
                d1(1:2, :) = -d1(1:2, :)
                d2(1:2, :) = -d2(1:2, :)
              endif
            endif
            if (par%keep_const(1) > 0) d1 = 0.d0                  ! :294-295
            if (par%keep_const(2) > 0) d2 = 0.d0
          endif
          cost = cost + tau**2
          do comp = 1, 3
            row = row + 1
            b = 0
            do t = 1, ne                                          ! model 1 columns, then model 2 columns (+ N), zeros dropped
              f = real(d1(t, comp) * pr(1)%cw(cell(t, comp)) * par%w_cross, c_float)
              if (f /= 0.0 .and. local_column(1, 1, cell(t, comp)) > 0) then
                b = b + 1;  ecol(b) = local_column(1, 1, cell(t, comp));  eval(b) = f
              endif
            enddo
            do t = 1, ne
              f = real(d2(t, comp) * pr(2)%cw(cell(t, comp)) * par%w_cross, c_float)
              if (f /= 0.0 .and. local_column(2, 1, cell(t, comp)) > 0) then
                b = b + 1;  ecol(b) = local_column(2, 1, cell(t, comp));  eval(b) = f
              endif
            enddo
            do t = 1, b                                           ! ascending columns for the upload (selection sort, <= 8 entries)
              cmin = t
              do a = t + 1, b
                if (ecol(a) < ecol(cmin)) cmin = a
              enddo
              g_cols(e + t) = ecol(cmin);  g_vals(e + t) = eval(cmin)
              ecol(cmin) = ecol(t);  eval(cmin) = eval(t)
            enddo
            e = e + b
            g_rowptr(row + 1) = e
            g_rhs(row) = -tau(comp) * par%w_cross
          enddo
        enddo
      enddo
    enddo
    g_nrows = row
    g_nnz = e
    print *, 'cross-grad cost =', cost
  end subroutine build_cross_gradient

  ! clustering_add for problem 1 then 2 (src/inversion/clustering.F90:393-499, joint_inverse_problem.F90:613-631): 2 N rows with one
  ! entry each - the derivative of the Gaussian mixture P(m1, m2) (or of -log P) times weight and column weight - and the
  ! right-hand side -weight (P - P_max) resp. -weight (log P_max - log P); appended after the other constraint rows.
  subroutine build_clustering()
    integer :: jp, p, i
    integer(c_int64_t) :: row, e
    real(dp) :: wloc(2), gauss, deriv(2), func, val(2), gc, dtmp(2), cost
    real(c_float) :: f
    if (.not. allocated(clust_mix)) call read_mixtures()
    wloc = merge(0.d0, 1.d0, par%w_clust == 0.d0)                 ! 1-D Gaussians when one weight is zero (:131-138)
    if (.not. allocated(clust_max)) then                          ! calculate_Gaussian_mixture_max (:647-674)
      allocate(clust_max(n))
      do p = 1, n
        clust_max(p) = 0.d0
        do i = 1, par%nclusters
          call gaussian_mixture((/ clust_mix(2, i), clust_mix(4, i) /), clust_cellw(:, p), wloc, gc, dtmp)
          if (gc > clust_max(p)) clust_max(p) = gc
        enddo
      enddo
      print *, 'Clustering mixture_max =', maxval(clust_max)
    endif
    row = g_nrows
    e = g_nnz
    do jp = 1, 2
      cost = 0.d0
      do p = 1, n
        val = (/ pr(1)%m(p), pr(2)%m(p) /)
        call gaussian_mixture(val, clust_cellw(:, p), wloc, gauss, deriv)
        if (par%clust_opt == 2) then
          if (gauss /= 0.d0) then
            deriv = -deriv / gauss
          else
            deriv = 0.d0
          endif
        endif
        row = row + 1
        f = real(par%w_clust(jp) * pr(jp)%cw(p) * deriv(jp) * wloc(jp), c_float)
        if (f /= 0.0 .and. local_column(jp, 1, p) > 0) then
          e = e + 1
          g_cols(e) = local_column(jp, 1, p)
          g_vals(e) = f
        endif
        g_rowptr(row + 1) = e
        if (par%clust_opt == 1) then
          func = gauss - clust_max(p)
        else if (par%clust_opt == 2) then
          func = 0.d0
          if (gauss > 0.d0) func = -log(gauss) + log(clust_max(p))
        else
          call stop_msg('Wrong optimization type in clustering_add!')
        endif
        g_rhs(row) = -par%w_clust(jp) * func * wloc(jp)
        cost = cost + g_rhs(row)**2
      enddo
      print *, 'clustering term', jp, 'cost = ', cost
    enddo
    g_nrows = row
    g_nnz = e
  end subroutine build_clustering

  ! clustering_calculate_Gaussian_mixture (:591-642) with the Gaussians of :505-584
  subroutine gaussian_mixture(val, cellw, wloc, gauss, deriv)
    real(dp), intent(in) :: val(2), cellw(:), wloc(2)
    real(dp), intent(out) :: gauss, deriv(2)
    real(dp), parameter :: PI = 3.14159265358979323846264338327950288d0
    real(dp) :: x, y, mu1, mu2, s11, s22, s12, arg, norm, gl, den
    integer :: i
    gauss = 0.d0
    deriv = 0.d0
    x = val(1)
    y = val(2)
    do i = 1, par%nclusters
      mu1 = clust_mix(2, i);  s11 = clust_mix(3, i);  mu2 = clust_mix(4, i);  s22 = clust_mix(5, i);  s12 = clust_mix(6, i)
      if (wloc(1) /= 0.d0 .and. wloc(2) /= 0.d0) then
        arg = (-((-mu2 + y) * (mu2 * s11**2 - mu1 * s12**2 + s12**2 * x - s11**2 * y)) / (s12**4 - s11**2 * s22**2) &
               - ((-mu1 + x) * (mu2 * s12**2 - mu1 * s22**2 + s22**2 * x - s12**2 * y)) / (-s12**4 + s11**2 * s22**2)) / 2.d0
        norm = 2.d0 * PI * sqrt(-s12**4 + s11**2 * s22**2)
      else if (wloc(2) == 0.d0) then
        arg = -(x - mu1)**2 / s11**2 / 2.d0
        norm = sqrt(2.d0 * PI * s11**2)
      else
        arg = -(y - mu2)**2 / s22**2 / 2.d0
        norm = sqrt(2.d0 * PI * s22**2)
      endif
      if (norm == 0.d0) call stop_msg('Zero norm in clustering_calculate_Gaussian!')
      if (arg < -100.d0) then
        gl = cellw(i) * exp(-100.d0)
      else
        gl = cellw(i) * (exp(arg) / norm)
      endif
      gauss = gauss + gl
      den = s12**4 - s11**2 * s22**2
      deriv(1) = deriv(1) + (s22**2 * (-mu1 + x) + s12**2 * (mu2 - y)) / den * gl
      deriv(2) = deriv(2) + (s12**2 * (mu1 - x) + s11**2 * (-mu2 + y)) / den * gl
    enddo
  end subroutine gaussian_mixture

  ! clustering_read_mixtures (:159-283): "nclusters", then per cluster: weight mu1 sigma1 mu2 sigma2 sigma12; per-cell cluster
  ! weights "nelements nclusters" + one line per cell (constraintsType 2), else the normalised global weights for every cell
  subroutine read_mixtures()
    integer :: u, ios, nread, cread, i, p
    if (.not. (pr(1)%on .and. pr(2)%on)) call stop_msg('The clustering constraint needs both problems (joint inversion).')
    if (par%nmodel_comp /= 1) call stop_msg('Clustering constraints need scalar models.')
    allocate(clust_mix(6, par%nclusters), clust_cellw(par%nclusters, n))
    print *, 'Reading clustering parameters from file ', trim(par%mixture_file)
    open(newunit=u, file=trim(par%mixture_file), status='old', action='read', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the mixture file! path='//trim(par%mixture_file))
    read(u, *, iostat=ios) nread
    if (ios /= 0 .or. nread /= par%nclusters) call stop_msg('The number of clusters is inconsistent!')
    do i = 1, par%nclusters
      read(u, *, iostat=ios) clust_mix(:, i)
      if (ios /= 0) call stop_msg('Problem while reading the mixture file in clustering_read_mixtures!')
    enddo
    close(u)
    clust_mix(1, :) = clust_mix(1, :) / sum(clust_mix(1, :))
    if (par%clust_cons /= 1) then
      print *, 'Reading clustering cell-weights:'
      open(newunit=u, file=trim(par%cell_weights_file), status='old', action='read', iostat=ios)
      if (ios /= 0) call stop_msg('Error in opening the cell-weights file! path='//trim(par%cell_weights_file))
      read(u, *, iostat=ios) nread, cread
      if (ios /= 0 .or. cread /= par%nclusters) call stop_msg('The number of clusters is inconsistent!')
      if (nread /= n) call stop_msg('The number of cells is inconsistent!')
      do p = 1, n
        read(u, *, iostat=ios) clust_cellw(:, p)
        if (ios /= 0) call stop_msg('Problem while reading the cell-weights file!')
      enddo
      close(u)
    else
      do p = 1, n
        clust_cellw(:, p) = clust_mix(1, :)
      enddo
    endif
  end subroutine read_mixtures

  ! get_grad (src/inversion/gradient.F90:68-86) with zeros outside the grid (grad_get_par, :196-225)
  subroutine cell_gradient(f, i, j, kk, scheme, st, g)
    real(dp), intent(in) :: f(:), st(3)
    integer, intent(in) :: i, j, kk, scheme
    real(dp), intent(out) :: g(3)
    real(dp) :: c
    c = fpar(f, i, j, kk)
    if (scheme == 1) then
      g = (/ (fpar(f, i + 1, j, kk) - c) / st(1), (fpar(f, i, j + 1, kk) - c) / st(2), (fpar(f, i, j, kk + 1) - c) / st(3) /)
    else if (scheme == 3) then
      g = (/ (c - fpar(f, i - 1, j, kk)) / st(1), (c - fpar(f, i, j - 1, kk)) / st(2), (c - fpar(f, i, j, kk - 1)) / st(3) /)
    else
      g = (/ (fpar(f, i + 1, j, kk) - fpar(f, i - 1, j, kk)) / 2.d0 / st(1), (fpar(f, i, j + 1, kk) - fpar(f, i, j - 1, kk)) / 2.d0 / st(2), &
             (fpar(f, i, j, kk + 1) - fpar(f, i, j, kk - 1)) / 2.d0 / st(3) /)
    endif
  end subroutine cell_gradient

  real(dp) function fpar(f, i, j, kk)
    real(dp), intent(in) :: f(:)
    integer, intent(in) :: i, j, kk
    fpar = 0.d0
    if (i < 1 .or. j < 1 .or. kk < 1 .or. i > par%nx .or. j > par%ny .or. kk > par%nz) return
    fpar = f(((kk - 1) * par%ny + (j - 1)) * par%nx + i)
  end function fpar

  ! column of (problem jp, model component kc, cell) in this rank's unknown vector [m1 cells (cb, ce]; m2 cells (cb, ce]], 1-based;
  ! 0: the cell belongs to another rank (constraint rows are replicated, every rank fills its own columns)
  integer function local_column(jp, kc, cell)
    integer, intent(in) :: jp, kc, cell
    local_column = 0
    if (cell <= cb .or. cell > ce) return
    local_column = (kc - 1) * nloc + cell - cb
    if (jp == 2 .and. pr(1)%on) local_column = local_column + pr(1)%nml
  end function local_column

  subroutine unweight(jp, v, res)
    integer, intent(in) :: jp
    real(dp), intent(in) :: v(:)
    real(dp), intent(out) :: res(:)
    integer :: p
    do p = 1, n
      if (pr(jp)%cw(p) /= 0.d0) then
        res(p) = v(p) / pr(jp)%cw(p)
      else
        res(p) = 0.d0
      endif
    enddo
  end subroutine unweight

  ! this rank's cells (cb, ce] of every model component of a full vector
  subroutine to_local(jp, vfull, vloc)
    integer, intent(in) :: jp
    real(dp), intent(in) :: vfull(:)
    real(dp), intent(out) :: vloc(:)
    integer :: kc
    do kc = 1, pr(jp)%nc
      vloc((kc - 1) * nloc + 1:kc * nloc) = vfull((kc - 1) * n + cb + 1:(kc - 1) * n + ce)
    enddo
  end subroutine to_local

  subroutine allreduce_sum_dp_scalar(v)
    real(dp), intent(inout) :: v
    real(dp) :: a(1)
    a(1) = v
    call allreduce_sum_dp(a, 1)
    v = a(1)
  end subroutine allreduce_sum_dp_scalar

  ! apply_local_depth_weighting, weights_gravmag.f90:255-309
  subroutine apply_local_depth_weighting(jp)
    integer, intent(in) :: jp
    real(dp), allocatable :: lw(:)
    integer :: p
    if (par%apply_local_dw <= 0) return
    allocate(lw(n))
    call read_cell_values(par%local_dw_file(jp), n, lw, 'local depth weights')
    do p = 1, n
      if (lw(p) /= 0.d0) then
        pr(jp)%cw(p) = pr(jp)%cw(p) / lw(p)
      else
        pr(jp)%cw(p) = 0.d0
      endif
    enddo
    deallocate(lw)
  end subroutine apply_local_depth_weighting

  ! read_depth_weight, sensitivity_gravmag.F90:920-970: the full column weight from the SENSIT folder (big-endian stream)
  subroutine read_depth_weight(jp)
    integer, intent(in) :: jp
    integer :: u, ios
    integer(c_int32_t) :: nread
    character(len=4), parameter :: sfx(2) = (/'grav', 'magn'/)
    open(newunit=u, file=trim(par%sensit_path)//'sensit_'//sfx(jp)//'_weight', status='old', access='stream', &
         form='unformatted', action='read', convert='big_endian', iostat=ios)
    if (ios /= 0) call stop_msg('Error in opening the depth weight file! path='//trim(par%sensit_path))
    read(u) nread
    if (nread /= n) call stop_msg('Depth weight file header does not match the Parfile!')
    read(u) pr(jp)%cw
    close(u)
  end subroutine read_depth_weight

  subroutine to_wavelet(v, ncomp)
    real(dp), intent(inout) :: v(:)
    integer, intent(in) :: ncomp
    integer :: kc
    if (ipar%compression_type <= 0) return
    do kc = 1, ncomp                ! every model component on its own (src/inversion/wavelet_utils.F90:37-72)
      call forward_wavelet(v((kc - 1) * n + 1:kc * n), ipar%nx, ipar%ny, ipar%nz, ipar%compression_type)
    enddo
  end subroutine to_wavelet

  ! model%calculate_data (src/inversion/model.F90:220-307) for the full model vector the host keeps: this rank's cells go into the
  ! reference's model object, the call is the reference's (problem_joint_gravmag.F90:333, :414, :438, :513)
  subroutine calculate_data(jp, mfull, dcalc)
    integer, intent(in) :: jp
    real(dp), intent(in) :: mfull(:)
    real(dp), intent(out) :: dcalc(:)
    real(dp), allocatable :: dc(:, :)
    real(dp) :: t0
    integer :: kc
    t0 = wall()
    do kc = 1, pr(jp)%nc
      model(jp)%val(:, kc) = mfull((kc - 1) * n + cb + 1:(kc - 1) * n + ce)
    enddo
    allocate(dc(pr(jp)%ndc, pr(jp)%nd))
    call model(jp)%calculate_data(pr(jp)%nd, pr(jp)%ndc, jinv%matrix_sensit, ipar%problem_weight(jp), cw_loc(:, jp), data(jp)%weight, dc, &
                                  ipar%compression_type, line_start(jp), param_shift(jp), myrank, nbproc)
    dcalc = reshape(dc, (/pr(jp)%ndt/))
    call lap(7, t0)
  end subroutine calculate_data

  ! calculate_cost_model, src/utils/costs.f90:74-113 (first model component only, problem_joint_gravmag.F90:655-657)
  subroutine model_cost(jp, cost)
    integer, intent(in) :: jp
    real(dp), intent(out) :: cost
    integer :: p
    cost = 0.d0
    do p = 1, n
      if (pr(jp)%cw(p) /= 0.d0) cost = cost + (abs((pr(jp)%m(p) - pr(jp)%m_prior(p)) / pr(jp)%cw(p)))**par%norm_power
    enddo
  end subroutine model_cost

  ! src/problem_joint_gravmag.F90:680-700
  logical function stop_file_exists()
    inquire(file='stop', exist=stop_file_exists)
  end function stop_file_exists

end subroutine solve_problem_joint_gravmag

end module problem_joint_gravmag

!=========================================================================================================
! program_tomofastx (src/program_tomofastx.F90:25-103): command line, MPI start-up, Parfile, then the reference's entry point
program tomofastx_amd
  use tfx_host_params
  use tfx_host_mpi
  use tfx_reference_api
  use problem_joint_gravmag
  implicit none
  type(t_par) :: par
  type(t_parameters_grav) :: gpar
  type(t_parameters_mag) :: mpar
  type(t_parameters_inversion) :: ipar
  character(len=256) :: arg, parfile
  integer :: i, narg

  ! ---- command line (src/parameters_init.f90:104-119)
  parfile = ''
  narg = command_argument_count()
  i = 1
  do while (i <= narg)
    call get_command_argument(i, arg)
    if (trim(arg) == '-p' .or. trim(arg) == '-j') then
      if (i + 1 > narg) call stop_msg('UNKNOWN Parfile! Use -p <Parfile_path>')
      call get_command_argument(i + 1, parfile)
      i = i + 1
    endif
    i = i + 1
  enddo
  if (len_trim(parfile) == 0) call stop_msg('UNKNOWN Parfile! Use -p <Parfile_path>')
  ! one process per GPU under `mpiexec -n P`; ranks other than 0 stay silent (the reference prints from rank 0 only)
  call host_mpi_init()
  abort_hook => host_mpi_abort
  io_rank = myrank == 0
  if (myrank /= 0) open(unit=6, file='/dev/null', status='old', action='write')
  print *, 'Started Tomofast-x (MI355X host), Parfile = ', trim(parfile)
  if (nbproc > 1) print *, 'Number of ranks (one GPU each) =', nbproc
  call read_parfile(parfile, par)
  if (par%admm > 0 .and. par%admm_bound_type == 1 .and. .not. allocated(par%bounds)) call stop_msg('Global bounds are not defined!')

  ! ---- the reference's three parameter objects (src/parameters_init.f90:412-966 fills them from the same keys)
  call set_base(gpar, 1)
  call set_base(mpar, 2)
  gpar%data_type = par%grav_data_type
  gpar%nmodel_components = 1
  mpar%mi = par%mag_incl
  mpar%md = par%mag_decl
  mpar%theta = par%mag_xaxis_decl
  mpar%intensity = par%mag_intensity
  ipar%nx = par%nx; ipar%ny = par%ny; ipar%nz = par%nz
  ipar%nelements_total = par%nx * par%ny * par%nz
  ipar%nelements = ipar%nelements_total
  ipar%ndata = par%ndata
  ipar%ndata_components = par%ndata_comp
  ipar%nmodel_components = par%nmodel_comp
  ipar%niter = par%nminor
  ipar%ninversions = par%nmajor
  ipar%alpha = par%alpha
  ipar%norm_power = par%norm_power
  ipar%rmin = par%rmin
  ipar%target_misfit = par%target_misfit
  ipar%gamma = par%gamma
  ipar%compression_type = par%comp_type
  ipar%problem_weight = par%pw
  ipar%column_weight_multiplier = par%cwm
  ipar%admm_type = par%admm
  ipar%rho_ADMM = par%rho
  ipar%admm_bound_type = par%admm_bound_type
  ipar%nlithos = par%nlithos
  ipar%apply_local_damping_weight = par%apply_local_damp
  ipar%beta = par%beta_grad
  ipar%cross_grad_weight = par%w_cross
  ipar%clustering_weight_glob = par%w_clust
  host_par = par

  call solve_problem_joint_gravmag(gpar, mpar, ipar, myrank, nbproc)

  print *, 'THE END.'
  call host_mpi_finalize()

contains

  subroutine set_base(b, ip)
    class(t_parameters_base), intent(inout) :: b
    integer, intent(in) :: ip
    b%nx = par%nx; b%ny = par%ny; b%nz = par%nz
    b%nelements = par%nx * par%ny * par%nz
    b%ndata = par%ndata(ip)
    b%ndata_components = par%ndata_comp(ip)
    b%nmodel_components = par%nmodel_comp
    b%depth_weighting_type = par%dw_type
    b%depth_weighting_power = par%dw_power(ip)
    b%depth_weighting_beta = par%dw_beta(ip)
    b%Z0 = par%dw_Z0(ip)
    b%compression_type = par%comp_type
    b%compression_rate = par%comp_rate
    b%sensit_read = par%sensit_read
    if (par%sensit_read == 0) then                 ! a fresh kernel is (optionally) written under the output folder (:142-153)
      b%sensit_path = trim(par%path_output)//'/SENSIT/'
    else
      b%sensit_path = par%sensit_path
    endif
  end subroutine set_base

end program tomofastx_amd
