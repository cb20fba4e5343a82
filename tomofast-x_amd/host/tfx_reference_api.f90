!=========================================================================================================
! tfx_reference_api - the reference's own entry points for the sensitivity-kernel hot path, over libtfx.so.
!
! Tomofast-x has no plugin interface: its boundary for this path is the set of Fortran procedures that
! src/problem_joint_gravmag.F90 and src/inversion/joint_inverse_problem.F90 call (SURVEY.md 8b).  This module keeps those
! names, argument orders and consume / overwrite semantics and forwards every O(N), O(N.Ndata) and O(nnz) step to the HIP
! library through tfx_binding (iso_c_binding) - a maintainer swaps `use sensitivity_gravmag / model / lsqr_solver /
! wavelet_transform` for `use tfx_reference_api` and the call sites stay as they are:
!
!   reference call site (src/problem_joint_gravmag.F90)                     reference procedure replaced
!   :174   calculate_depth_weight(par, iarr, grid_full, data, myrank, nbproc)         weights_gravmag.f90:46-196
!          (iarr: t_inversion_arrays with column_weight / residuals and allocate_aux / reallocate_aux, inversion_arrays.f90:30-44)
!   :197   calculate_and_write_sensit(par, grid_full, data, column_weight, memory, myrank, nbproc)
!                                                                                sensitivity_gravmag.F90:82-410
!   :207   calculate_new_partitioning(par, nnz, nelements_at_cpu, problem_type, myrank, nbproc)     :573-640
!   :241   read_sensitivity_kernel(par, sensit_matrix, column_weight, problem_weight, data_weight, problem_type,
!                                  myrank, nbproc, nelements_at_cpu)                                 :648-883
!   :333   model%calculate_data(ndata, ndata_components, matrix_sensit, problem_weight, column_weight, data_weight,
!                               data_calc, compression_type, line_start, param_shift, myrank, nbproc)  model.F90:220-307
!   joint_inverse_problem.F90:549  lsqr_solve_sensit(nlines, ncolumns, niter, rmin, gamma, target_misfit, matrix_sensit,
!                               matrix_cons, u, x, SOLVE_PROBLEM, nelements, nx, ny, nz, ncomponents, compression_type,
!                               WAVELET_DOMAIN, memory, myrank, nbproc)                      lsqr_solver2.F90:47-308
!   anywhere  forward_wavelet / inverse_wavelet(s, n1, n2, n3, wavelet_type)           wavelet_transform.F90:37-70
!
! Derived types carry the reference's field names for the fields the path reads (t_parameters_base / _grav / _mag,
! t_grid, t_data, t_model); t_sparse_matrix is the handle of the device-resident matrix for the sensitivity kernel and a
! host-side row builder (add / new_row / finalize, sparse_matrix.f90:213-293) for the constraint rows, which
! lsqr_solve_sensit hands to the device: runs of single-entry rows over this rank's cells (what damping%add builds,
! damping.F90:158-179) become diagonal blocks applied on the fly, everything else is uploaded as general rows.
!
! What differs from the reference by design: no SENSIT files are needed between calculate_and_write_sensit and
! read_sensitivity_kernel - the compressed rows stay on the GPU (single rank: already as the tiled matrix; several ranks: in
! the device row store until the partition is known, then GPU-to-GPU relayout).  The files are still written when asked for
! (par%sensit_write, default for small kernels) and read when par%sensit_read = 1, in the reference's format.
! One process drives one GPU (rank r -> device mod(r, ndev)); errors follow the reference's convention: banner + abort.
!=========================================================================================================
module tfx_reference_api
  use iso_c_binding
  use tfx_binding
  use tfx_host_mpi
  implicit none
  private

  integer, parameter, public :: CUSTOM_REAL = c_double, MATRIX_PRECISION = c_float
  integer, parameter :: dp = c_double
  integer, parameter :: ROW_BLOCK = 2048           ! rows per row block of the device matrix (tfx_matrix_append_rows)
  character(len=4), parameter :: SENSIT_SUFFIX(2) = (/'grav', 'magn'/)

  ! ---- src/forward/gravmag/parameters_gravmag.f90:30-108 (fields the path reads; same names)
  type, public :: t_parameters_base
    integer :: nx = 0, ny = 0, nz = 0
    integer :: nelements = 0                       ! cells on this rank (the whole model before calculate_new_partitioning)
    integer :: ndata = 0, ndata_components = 1, nmodel_components = 1
    integer :: data_type = 1                       ! 1 = g_z, 2 = gradiometry (gravity problem)
    integer :: depth_weighting_type = 2
    real(dp) :: depth_weighting_power = 2.d0, depth_weighting_beta = 1.d0, Z0 = 0.d0
    integer :: compression_type = 0
    real(dp) :: compression_rate = 1.d0
    integer :: sensit_read = 0
    character(len=256) :: sensit_path = 'SENSIT/'
    integer :: sensit_write = -1                   ! not in the reference (it always writes): 1 write, 0 do not, -1 only small kernels
  end type t_parameters_base

  type, extends(t_parameters_base), public :: t_parameters_grav    ! parameters_grav.f90:30-38
  end type t_parameters_grav

  type, extends(t_parameters_base), public :: t_parameters_mag     ! parameters_mag.f90:30-48
    real(dp) :: mi = 90.d0, md = 0.d0, theta = 0.d0, intensity = 50000.d0
  end type t_parameters_mag

  ! ---- src/inversion/parameters_inversion.f90:31-130 (the scalar fields the solve path reads)
  type, public :: t_parameters_inversion
    integer :: nx = 0, ny = 0, nz = 0
    integer :: nelements = 0, nelements_total = 0
    integer :: ndata(2) = 0, ndata_components(2) = 1, nmodel_components = 1
    integer :: niter = 100, ninversions = 10
    real(dp) :: alpha(2) = 0.d0, norm_power = 2.d0
    real(dp) :: rmin = 1.d-13, target_misfit = 0.d0, gamma = 0.d0
    integer :: method = 1
    integer :: compression_type = 0
    real(dp) :: problem_weight(2) = (/1.d0, 0.d0/), column_weight_multiplier(2) = 1.d0
    integer :: admm_type = 0
    real(dp) :: rho_ADMM(2) = 1.d-7
    ! what decides WAVELET_DOMAIN (joint_inverse_problem.F90:189-198) and the shape of the damping / ADMM blocks; same names
    integer :: admm_bound_type = 1, nlithos = 1
    integer :: apply_local_damping_weight = 0
    real(dp) :: beta(2) = 0.d0, cross_grad_weight = 0.d0, clustering_weight_glob(2) = 0.d0
  end type t_parameters_inversion

  ! ---- src/inversion/grid.F90:30-50 (plain allocatables instead of shared-memory windows: one process per GPU)
  type, public :: t_grid
    integer :: nx = 0, ny = 0, nz = 0
    integer :: z_axis_dir = 1
    real(dp), allocatable :: X1(:), Y1(:), Z1(:), X2(:), Y2(:), Z2(:)
  contains
    procedure, public, pass :: allocate => grid_allocate
    procedure, public, pass :: deallocate => grid_deallocate
  end type t_grid

  ! ---- src/forward/gravmag/data_gravmag.f90:30-52
  type, public :: t_data
    integer :: ndata = 0, ncomponents = 1
    real(dp) :: units_mult = 1.d0
    integer :: z_axis_dir = 1
    real(dp), allocatable :: X(:), Y(:), Z(:)
    real(dp), allocatable :: val_meas(:, :), val_calc(:, :), weight(:, :)
  contains
    procedure, public, pass :: initialize => data_initialize
  end type t_data

  ! ---- src/inversion/sparse_matrix.f90:31-98
  type, public :: t_sparse_matrix
    ! the sensitivity kernel: device-resident, this object only knows where (ctx) and how large
    logical :: on_device = .false.
    integer :: nproblems = 0                        ! kernels loaded so far (joint inversion: two column / row blocks)
    integer :: nl_device = 0, ncolumns_device = 0
    ! constraint rows: host-side CSR under construction (add / new_row / finalize)
    integer :: nl = 0, nl_current = 0, ncolumns = 0
    integer(c_int64_t) :: nel = 0
    integer(c_int64_t), allocatable :: ijl(:)       ! row offsets (0-based), nl + 1
    integer(c_int32_t), allocatable :: ija(:)       ! 1-based local columns
    real(c_float), allocatable :: sa(:)
    integer(c_int64_t) :: nnz = 0                   ! predicted number of elements (initialize)
    real(dp), allocatable, public :: lsqr_var(:)    ! sparse_matrix.f90:67 (auxiliary array of the solution variance)
    integer, public :: tag = 0                      ! sparse_matrix.f90:70
  contains
    ! the reference's public interface, same names and argument lists (sparse_matrix.f90:72-98)
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
  end type t_sparse_matrix

  ! ---- src/inversion/model.F90:33-110 (the fields model_calculate_data reads)
  type, public :: t_model
    integer :: nelements = 0, nelements_total = 0, ncomponents = 1
    real(dp), allocatable :: val(:, :)               ! (nelements, ncomponents): this rank's cells
    real(dp), allocatable :: val_prior(:, :)         ! prior model (local), same shape
    integer :: nlithos = 1                           ! data arrays for the ADMM constraints (local cells)
    real(dp), allocatable :: min_bound(:, :), max_bound(:, :)      ! (nlithos, nelements)
    real(dp), allocatable :: bound_weight(:)
    real(dp), allocatable :: damping_weight(:)       ! local damping weights (for the prior model term)
    type(t_grid) :: grid_full
  contains
    procedure, public, pass :: initialize => model_initialize
    procedure, public, pass :: allocate_bound_arrays => model_allocate_bound_arrays
    procedure, public, pass :: update => model_update
    procedure, public, pass :: calculate_data => model_calculate_data
  end type t_model

  ! ---- what the reference keeps in the SENSIT folder between the calls, kept here (and on the device) instead
  type t_kernel_state
    logical :: built = .false.
    integer :: slot = -1                             ! tfx_select_problem slot of this kernel
    integer :: mode = 0                              ! 1 tiles built on this rank (single rank), 2 rows in the device row store,
                                                     ! 3 counted only: every rank builds all rows for its columns on reload
    integer :: row_a = 0, row_b = 0                  ! this rank's data (row_a, row_b] of the row-parallel build
    integer(c_int32_t), allocatable :: nnz_hist(:)   ! sensit_nnz, summed over the ranks (sensitivity_gravmag.F90:322)
    real(dp), allocatable :: cw_full(:)              ! the column weight the kernel was built with (sensit_*_weight)
    real(dp), allocatable :: Xd(:), Yd(:), Zd(:)
    real(dp) :: mag_field(4) = 0.d0
    integer :: data_type = 1, ndc = 1, ncm = 1
    real(dp) :: err_sum = 0.d0
    integer(c_int64_t) :: nnz_total = 0
  end type t_kernel_state

  type(t_kernel_state), save, target :: kst(2)
  type(c_ptr), save :: api_ctx = c_null_ptr
  integer, save :: nslots_used = 0
  integer, save :: part_cb = 0, part_ce = 0          ! this rank's cells (part_cb, part_ce] after calculate_new_partitioning
  integer, allocatable, save :: part_nel(:)
  logical, save :: partitioned = .false.

  ! ---- src/inversion/inversion_arrays.f90:30-44: the auxiliary arrays of the inversion, same field and procedure names
  type, public :: t_inversion_arrays
    real(dp), allocatable :: residuals(:, :)       ! (ndata_components, ndata): data measured - data calculated
    real(dp), allocatable :: column_weight(:)      ! weights that scale the sensitivity matrix columns (here over ALL cells)
  contains
    procedure, public, pass :: allocate_aux => inversion_arrays_allocate_aux
    procedure, public, pass :: reallocate_aux => inversion_arrays_reallocate_aux
  end type t_inversion_arrays

  ! calculate_depth_weight(par, iarr, grid_full, data, myrank, nbproc) is the reference's form (weights_gravmag.f90:46); the form with
  ! the plain array in iarr's place is kept for callers that hold no t_inversion_arrays
  interface calculate_depth_weight
    module procedure calculate_depth_weight_iarr, calculate_depth_weight_array
  end interface calculate_depth_weight

  ! ---- src/inversion/joint_inverse_problem.F90:42-123: the joint inversion (also used for single inversions) - the two matrices
  ! of the system, its right-hand side, the ADMM state.  Same procedure names and argument lists as the reference:
  !   problem_joint_gravmag.F90:236  call jinv%initialize(ipar, nnz, myrank)
  !                            :263  call jinv%initialize2(ipar, iarr, model, myrank, nbproc)
  !                            :327  call jinv%calculate_matrix_partitioning(ipar, line_start, line_end, param_shift)
  !                       :393, :494  call jinv%reset(myrank)
  !                            :497  call jinv%solve(ipar, iarr, model, delta_model, memory_inv, myrank, nbproc)
  ! solve assembles b_RHS = [problem_weight * residuals ; constraint right-hand sides] and the constraint rows - the damping block
  ! (damping.F90:97-234: L2 or Lp, optional local weights) and the ADMM block (admm_method.F90:70-134) of every active problem,
  ! then the general rows the caller registered - and hands the system to lsqr_solve_sensit, i.e. to the GPU; the update comes back
  ! in model space (inverse transform when WAVELET_DOMAIN, times the column weight: joint_inverse_problem.F90:556-571).
  ! The builders of gradient damping, cross-gradient and clustering rows live with the caller (they are outside the hot path): it
  ! passes their rows in with set_general_rows before solve, where the reference builds them inside solve.
  type, public :: t_joint_inversion
    type(t_sparse_matrix), public :: matrix_sensit          ! the (device-resident) sensitivity kernel(s)
    type(t_sparse_matrix), public :: matrix_cons            ! the constraint rows of the current major iteration
    real(dp), allocatable :: b_RHS(:)
    integer :: nelements_total = 0, ndata_lines = 0
    logical :: add_damping(2) = .false., add_damping_gradient(2) = .false., add_admm(2) = .false.
    logical, public :: add_cross_grad = .false., add_clustering = .false.
    real(dp) :: admm_cost(2) = 0.d0
    logical, public :: WAVELET_DOMAIN = .true.
    real(dp), allocatable :: z_admm(:, :), u_admm(:, :), x0_ADMM(:, :)      ! (nelements, problem): admm_method's arrays
    integer(c_int64_t) :: g_nrows = 0                        ! general rows of the next solve (set_general_rows)
    integer(c_int64_t), allocatable :: g_rowptr(:)
    integer(c_int32_t), allocatable :: g_cols(:)
    real(c_float), allocatable :: g_vals(:)
    real(dp), allocatable :: g_rhs(:)
  contains
    procedure, public, pass :: initialize => joint_inversion_initialize
    procedure, public, pass :: initialize2 => joint_inversion_initialize2
    procedure, public, pass :: reset => joint_inversion_reset
    procedure, public, pass :: solve => joint_inversion_solve
    procedure, public, pass :: get_admm_cost => joint_inversion_get_admm_cost
    procedure, public, pass :: set_general_rows => joint_inversion_set_general_rows
    procedure, public, nopass :: calculate_matrix_partitioning => joint_inversion_calculate_matrix_partitioning
  end type t_joint_inversion

  public :: tfx_api_context, tfx_api_finalize, exit_MPI
  public :: calculate_depth_weight, calculate_and_write_sensit, calculate_new_partitioning, read_sensitivity_kernel
  public :: model_calculate_data, lsqr_solve_sensit, forward_wavelet, inverse_wavelet
  public :: get_full_array, write_sensit_rank_file_enabled
  public :: tfx_api_kernel_slot, api_check, api_canonical_csr

contains

  !-------------------------------------------------------------------------------------------------------
  ! exit_MPI (src/utils/mpi_tools.F90:29-53): banner, then abort every rank
  subroutine exit_MPI(msg, rank, ierr)
    character(len=*), intent(in) :: msg
    integer, intent(in) :: rank, ierr
    print *, '**********************************************'
    print *, 'ERROR: ', trim(msg)
    print *, 'rank =', rank, ' error code =', ierr
    print *, '**********************************************'
    flush(6)
    call host_mpi_abort()
    stop 1
  end subroutine exit_MPI

  subroutine api_check(rc, where, rank)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    integer, intent(in) :: rank
    character(kind=c_char), pointer :: msg(:)
    character(len=1024) :: text
    integer :: i
    if (rc == 0) return
    call c_f_pointer(tfx_last_error(), msg, [1024])
    text = ''
    do i = 1, 1024
      if (msg(i) == c_null_char) exit
      text(i:i) = msg(i)
    enddo
    call exit_MPI(where//': '//trim(text), rank, int(rc))
  end subroutine api_check

  ! The GPU context of this process (created on first use: rank r drives device mod(r, ndev); several ranks get their collectives)
  function tfx_api_context(rank, nranks) result(ctx)
    integer, intent(in) :: rank, nranks
    type(c_ptr) :: ctx
    integer :: ndev
    if (.not. c_associated(api_ctx)) then
      call host_mpi_attach()          ! (a caller that started MPI itself: the reference's program with the drop-in modules)
      ndev = tfx_device_count()
      if (ndev <= 0) call exit_MPI('No HIP device visible - the MI355X path has no CPU fallback.', rank, 0)
      call api_check(tfx_create(int(host_device_for_rank(), c_int), c_null_ptr, api_ctx), 'tfx_create', rank)
      if (nranks > 1) call host_comm_setup(api_ctx)
    endif
    ctx = api_ctx
  end function tfx_api_context

  ! tfx_select_problem slot of the kernel of problem ip (1 gravity, 2 magnetic) once it has been built or loaded; -1 before
  integer function tfx_api_kernel_slot(ip)
    integer, intent(in) :: ip
    tfx_api_kernel_slot = -1
    if (ip >= 1 .and. ip <= 2) then
      if (kst(ip)%built) tfx_api_kernel_slot = kst(ip)%slot
    endif
  end function tfx_api_kernel_slot

  subroutine tfx_api_finalize()
    integer :: ip
    if (c_associated(api_ctx)) then
      if (tfx_destroy(api_ctx) /= 0) continue
      api_ctx = c_null_ptr
    endif
    do ip = 1, 2
      kst(ip)%built = .false.
      kst(ip)%slot = -1
      kst(ip)%mode = 0
    enddo
    nslots_used = 0
    partitioned = .false.
  end subroutine tfx_api_finalize

  !-------------------------------------------------------------------------------------------------------
  subroutine grid_allocate(this, nx, ny, nz, z_axis_dir, myrank)
    class(t_grid), intent(inout) :: this
    integer, intent(in) :: nx, ny, nz, z_axis_dir, myrank
    integer :: n, ierr
    this%nx = nx; this%ny = ny; this%nz = nz
    this%z_axis_dir = z_axis_dir
    n = nx * ny * nz
    call this%deallocate()
    allocate(this%X1(n), this%Y1(n), this%Z1(n), this%X2(n), this%Y2(n), this%Z2(n), stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in grid_allocate!', myrank, ierr)
  end subroutine grid_allocate

  subroutine grid_deallocate(this)
    class(t_grid), intent(inout) :: this
    if (allocated(this%X1)) deallocate(this%X1, this%Y1, this%Z1, this%X2, this%Y2, this%Z2)
  end subroutine grid_deallocate

  subroutine data_initialize(this, ndata, ncomponents, units_mult, z_axis_dir, myrank)
    class(t_data), intent(inout) :: this
    integer, intent(in) :: ndata, ncomponents, z_axis_dir, myrank
    real(dp), intent(in) :: units_mult
    integer :: ierr
    this%ndata = ndata; this%ncomponents = ncomponents
    this%units_mult = units_mult; this%z_axis_dir = z_axis_dir
    if (allocated(this%X)) deallocate(this%X, this%Y, this%Z, this%val_meas, this%val_calc, this%weight)
    allocate(this%X(ndata), this%Y(ndata), this%Z(ndata), this%val_meas(ncomponents, ndata), this%val_calc(ncomponents, ndata), &
             this%weight(ncomponents, ndata), stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in data_initialize!', myrank, ierr)
    this%val_meas = 0.d0; this%val_calc = 0.d0
    this%weight = 1.d0                                                           ! data_gravmag.f90:85
  end subroutine data_initialize

  subroutine model_initialize(this, nelements, ncomponents, nelements_total, myrank)
    class(t_model), intent(inout) :: this
    integer, intent(in) :: nelements, ncomponents, nelements_total, myrank
    integer :: ierr
    this%nelements = nelements; this%ncomponents = ncomponents; this%nelements_total = nelements_total
    if (allocated(this%val)) deallocate(this%val)
    if (allocated(this%val_prior)) deallocate(this%val_prior)
    if (allocated(this%damping_weight)) deallocate(this%damping_weight)
    allocate(this%val(nelements, ncomponents), source=0.d0, stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in model_initialize!', myrank, ierr)
    allocate(this%val_prior(nelements, ncomponents), source=0.d0, stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in model_initialize!', myrank, ierr)
    allocate(this%damping_weight(nelements), source=1.d0, stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in model_initialize!', myrank, ierr)
  end subroutine model_initialize

  subroutine model_allocate_bound_arrays(this, nlithos, myrank)                   ! model.F90:150-170
    class(t_model), intent(inout) :: this
    integer, intent(in) :: nlithos, myrank
    integer :: ierr
    this%nlithos = nlithos
    if (allocated(this%min_bound)) deallocate(this%min_bound, this%max_bound, this%bound_weight)
    allocate(this%min_bound(nlithos, this%nelements), this%max_bound(nlithos, this%nelements), this%bound_weight(this%nelements), stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in model_allocate_bound_arrays!', myrank, ierr)
    this%min_bound = 0.d0; this%max_bound = 0.d0; this%bound_weight = 1.d0
  end subroutine model_allocate_bound_arrays

  subroutine model_update(this, delta_model)                                      ! model.F90:194-200
    class(t_model), intent(inout) :: this
    real(dp), intent(in) :: delta_model(this%nelements, this%ncomponents)
    this%val = this%val + delta_model
  end subroutine model_update

  !-------------------------------------------------------------------------------------------------------
  ! t_sparse_matrix as a row builder for the constraint matrix (sparse_matrix.f90:107-293)
  subroutine sparse_matrix_initialize(this, nl, ncolumns, nnz, myrank, nl_empty)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: nl, ncolumns, myrank
    integer, intent(in), optional :: nl_empty       ! (the reference sizes its non-empty-row index by it; every row has an offset here)
    integer(c_int64_t), intent(in) :: nnz
    integer :: ierr
    this%nl = nl; this%ncolumns = ncolumns; this%nnz = nnz
    this%nl_current = 0; this%nel = 0
    if (allocated(this%ijl)) deallocate(this%ijl, this%ija, this%sa)
    allocate(this%ijl(nl + 1), this%ija(max(nnz, 1_c_int64_t)), this%sa(max(nnz, 1_c_int64_t)), stat=ierr)
    if (ierr /= 0) call exit_MPI('Dynamic memory allocation error in sparse_matrix_initialize!', myrank, ierr)
    this%ijl = 0
  end subroutine sparse_matrix_initialize

  subroutine sparse_matrix_reset(this)                                          ! sparse_matrix.f90:143-150
    class(t_sparse_matrix), intent(inout) :: this
    this%nl_current = 0; this%nel = 0
    if (allocated(this%ijl)) this%ijl = 0
  end subroutine sparse_matrix_reset

  subroutine sparse_matrix_add(this, value, column, myrank)                     ! sparse_matrix.f90:213-236
    class(t_sparse_matrix), intent(inout) :: this
    real(dp), intent(in) :: value
    integer, intent(in) :: column, myrank
    if (value == 0.d0) return                                                   ! zeros are not stored (:219)
    if (this%nel >= size(this%sa, kind=c_int64_t)) &
      call exit_MPI('Error in total number of elements in sparse_matrix_add!', myrank, 0)
    this%nel = this%nel + 1
    this%sa(this%nel) = real(value, MATRIX_PRECISION)                           ! the one cast to the matrix precision (:226)
    this%ija(this%nel) = column
  end subroutine sparse_matrix_add

  subroutine sparse_matrix_add_row(this, nel_add, values, columns, myrank)      ! sparse_matrix.f90:234-248 (values already in matrix precision)
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: nel_add
    real(c_float), intent(in) :: values(nel_add)
    integer, intent(in) :: columns(nel_add)
    integer, intent(in) :: myrank
    if (this%nel + nel_add > size(this%sa, kind=c_int64_t)) &
      call exit_MPI('Error in total number of elements in sparse_matrix_add_row!', myrank, 0)
    this%sa(this%nel + 1:this%nel + nel_add) = values
    this%ija(this%nel + 1:this%nel + nel_add) = columns
    this%nel = this%nel + nel_add
  end subroutine sparse_matrix_add_row

  subroutine sparse_matrix_add_empty_rows(this, nrows, myrank)                  ! sparse_matrix.f90:281-293
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: nrows, myrank
    integer :: i
    if (this%nl_current + nrows > this%nl) call exit_MPI('Error in number of rows in sparse_matrix_add_empty_rows!', myrank, 0)
    do i = 1, nrows
      this%nl_current = this%nl_current + 1
      this%ijl(this%nl_current + 1) = this%nel
    enddo
  end subroutine sparse_matrix_add_empty_rows

  subroutine sparse_matrix_new_row(this, myrank)                                ! sparse_matrix.f90:242-276
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: myrank
    if (this%nl_current >= this%nl) call exit_MPI('Error in number of rows in sparse_matrix_new_row!', myrank, 0)
    this%nl_current = this%nl_current + 1
    this%ijl(this%nl_current + 1) = this%nel
  end subroutine sparse_matrix_new_row

  subroutine sparse_matrix_finalize(this, myrank)                               ! sparse_matrix.f90:282-293
    class(t_sparse_matrix), intent(inout) :: this
    integer, intent(in) :: myrank
    if (this%on_device) return
    if (this%nl_current /= this%nl) call exit_MPI('Wrong total number of rows in sparse_matrix_finalize!', myrank, this%nl_current)
  end subroutine sparse_matrix_finalize

  pure function sparse_matrix_get_total_row_number(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer :: res
    res = merge(this%nl_device, this%nl, this%on_device)
  end function sparse_matrix_get_total_row_number

  pure function sparse_matrix_get_current_row_number(this) result(res)            ! sparse_matrix.f90:458-463
    class(t_sparse_matrix), intent(in) :: this
    integer :: res
    res = this%nl_current
  end function sparse_matrix_get_current_row_number

  pure function sparse_matrix_get_nnz(this) result(res)                           ! sparse_matrix.f90:488-493
    class(t_sparse_matrix), intent(in) :: this
    integer(c_int64_t) :: res
    res = this%nnz
  end function sparse_matrix_get_nnz

  pure function sparse_matrix_get_ncolumns(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer :: res
    res = merge(this%ncolumns_device, this%ncolumns, this%on_device)
  end function sparse_matrix_get_ncolumns

  pure function sparse_matrix_get_number_elements(this) result(res)
    class(t_sparse_matrix), intent(in) :: this
    integer(c_int64_t) :: res
    res = this%nel
  end function sparse_matrix_get_number_elements

  ! b (+)= S x / b (+)= S^T x with the device-resident kernel (sparse_matrix.f90:298-329, :373-405), the reference's argument lists;
  ! single kernel (slot 0) only - the joint system is applied by lsqr_solve_sensit.  There is no host-side product: a host matrix aborts.
  subroutine sparse_matrix_mult_vector(this, x, b)
    class(t_sparse_matrix), intent(in) :: this
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: b(:)
    if (.not. this%on_device) call exit_MPI('mult_vector: only the device-resident sensitivity kernel has products.', 0, 0)
    call api_check(tfx_select_problem(api_ctx, 0_c_int), 'tfx_select_problem', 0)
    call api_check(tfx_spmv(api_ctx, x, b, 0_c_int), 'mult_vector', 0)
  end subroutine sparse_matrix_mult_vector

  subroutine sparse_matrix_add_mult_vector(this, x, b)
    class(t_sparse_matrix), intent(in) :: this
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: b(:)
    if (.not. this%on_device) call exit_MPI('add_mult_vector: only the device-resident sensitivity kernel has products.', 0, 0)
    call api_check(tfx_select_problem(api_ctx, 0_c_int), 'tfx_select_problem', 0)
    call api_check(tfx_spmv(api_ctx, x, b, 1_c_int), 'add_mult_vector', 0)
  end subroutine sparse_matrix_add_mult_vector

  ! sparse_matrix.f90:335-367: rows [line_start, line_start + ndata) of the joint matrix times x, columns shifted by param_shift - the
  ! block of ONE kernel: the kernel in the slot whose rows start there (0 / 1 in load order)
  subroutine sparse_matrix_part_mult_vector(this, nelements, x, ndata, b, line_start, param_shift, myrank)
    class(t_sparse_matrix), intent(in) :: this
    integer, intent(in) :: nelements, ndata
    real(dp), intent(in) :: x(nelements)
    integer, intent(in) :: line_start, param_shift
    integer, intent(in) :: myrank
    real(dp), intent(out) :: b(ndata)
    integer :: slot, ip
    integer(c_int64_t) :: nr, nc, nz, dbytes
    if (.not. this%on_device) call exit_MPI('part_mult_vector: only the device-resident sensitivity kernel has products.', myrank, 0)
    ! The caller addresses the block of problem 1 with (line_start 1, param_shift 0) and the block of problem 2 with the rows behind
    ! problem 1's and param_shift = nelements * nmodel_components(1) - ALSO when problem 2 is the only one solved (line_start 1,
    ! param_shift /= 0: problem_joint_gravmag.F90:333, model.F90:258-262), so neither number says which slot holds the kernel: the
    ! kernels recorded by read_sensitivity_kernel do (ADVICE r5)
    ip = merge(1, 2, param_shift == 0)
    if (.not. kst(ip)%built) ip = 3 - ip
    if (.not. kst(ip)%built .or. kst(ip)%slot < 0) call exit_MPI('part_mult_vector: no sensitivity kernel has been loaded.', myrank, 0)
    slot = kst(ip)%slot
    call api_check(tfx_select_problem(api_ctx, int(slot, c_int)), 'tfx_select_problem', myrank)
    call api_check(tfx_matrix_info(api_ctx, nr, nc, nz, dbytes), 'tfx_matrix_info', myrank)
    if (nr /= ndata .or. nc /= nelements) call exit_MPI('Wrong line index in sparse_matrix_part_mult_vector!', myrank, 0)
    call api_check(tfx_spmv(api_ctx, x, b, 0_c_int), 'part_mult_vector', myrank)
    call api_check(tfx_select_problem(api_ctx, 0_c_int), 'tfx_select_problem', myrank)
  end subroutine sparse_matrix_part_mult_vector

  subroutine sparse_matrix_trans_mult_vector(this, x, b)
    class(t_sparse_matrix), intent(in) :: this
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: b(:)
    if (.not. this%on_device) call exit_MPI('trans_mult_vector: only the device-resident sensitivity kernel has products.', 0, 0)
    call api_check(tfx_select_problem(api_ctx, 0_c_int), 'tfx_select_problem', 0)
    call api_check(tfx_spmtv(api_ctx, x, b, 0_c_int), 'trans_mult_vector', 0)
  end subroutine sparse_matrix_trans_mult_vector

  subroutine sparse_matrix_add_trans_mult_vector(this, x, b)
    class(t_sparse_matrix), intent(in) :: this
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: b(:)
    if (.not. this%on_device) call exit_MPI('add_trans_mult_vector: only the device-resident sensitivity kernel has products.', 0, 0)
    call api_check(tfx_select_problem(api_ctx, 0_c_int), 'tfx_select_problem', 0)
    call api_check(tfx_spmtv(api_ctx, x, b, 1_c_int), 'add_trans_mult_vector', 0)
  end subroutine sparse_matrix_add_trans_mult_vector

  ! sparse_matrix.f90:414-443: unit-length columns, returns the original norms (device-resident kernel only)
  subroutine sparse_matrix_normalize_columns(this, column_norm)
    class(t_sparse_matrix), intent(inout) :: this
    real(dp), intent(out) :: column_norm(:)
    if (.not. this%on_device) call exit_MPI('normalize_columns: only the device-resident sensitivity kernel is normalised.', 0, 0)
    call api_check(tfx_select_problem(api_ctx, 0_c_int), 'tfx_select_problem', 0)
    call api_check(tfx_matrix_normalize_columns(api_ctx, column_norm), 'normalize_columns', 0)
  end subroutine sparse_matrix_normalize_columns

  !-------------------------------------------------------------------------------------------------------
  ! forward_wavelet / inverse_wavelet, src/utils/wavelet_transform.F90:37-70: in place on s(n1*n2*n3)
  subroutine forward_wavelet(s, n1, n2, n3, wavelet_type)
    integer, intent(in) :: n1, n2, n3, wavelet_type
    real(dp), intent(inout) :: s(n1 * n2 * n3)
    if (wavelet_type /= 1 .and. wavelet_type /= 2) then
      print *, 'Unknown wavelet type!'                                        ! :46-48
      stop
    endif
    call api_check(tfx_wavelet(tfx_api_context(myrank, nbproc), s, n1, n2, n3, 1_c_int64_t, wavelet_type, 1_c_int), 'forward_wavelet', myrank)
  end subroutine forward_wavelet

  subroutine inverse_wavelet(s, n1, n2, n3, wavelet_type)
    integer, intent(in) :: n1, n2, n3, wavelet_type
    real(dp), intent(inout) :: s(n1 * n2 * n3)
    if (wavelet_type /= 1 .and. wavelet_type /= 2) then
      print *, 'Unknown wavelet type!'                                        ! :65-67
      stop
    endif
    call api_check(tfx_wavelet(tfx_api_context(myrank, nbproc), s, n1, n2, n3, 1_c_int64_t, wavelet_type, 2_c_int), 'inverse_wavelet', myrank)
  end subroutine inverse_wavelet

  ! get_full_array (src/utils/parallel_tools.f90): the cell slices of all ranks -> the full array, on every rank
  subroutine get_full_array(loc, nloc, full, myrank_, nbproc_)
    integer, intent(in) :: nloc, myrank_, nbproc_
    real(dp), intent(in) :: loc(nloc)
    real(dp), intent(out) :: full(:)
    integer, allocatable :: displs(:)
    integer :: r
    if (nbproc_ == 1 .or. .not. partitioned) then
      full(1:nloc) = loc
      return
    endif
    allocate(displs(nbproc_))
    displs(1) = 0
    do r = 2, nbproc_
      displs(r) = displs(r - 1) + part_nel(r - 1)
    enddo
    if (part_nel(myrank_ + 1) /= nloc) call exit_MPI('Wrong local size in get_full_array!', myrank_, nloc)
    call allgather_slices(loc, nloc, full, part_nel, displs)
  end subroutine get_full_array

  !-------------------------------------------------------------------------------------------------------
  ! inversion_arrays_allocate_aux / _reallocate_aux, src/inversion/inversion_arrays.f90:50-95.  Before calculate_new_partitioning
  ! every rank holds the whole model (par%nelements = all cells), so column_weight is the full vector the row generators need;
  ! reallocate_aux re-sizes it to the cells of the new partition like the reference.
  subroutine inversion_arrays_allocate_aux(this, nelements, ndata, ndata_components, myrank_)
    class(t_inversion_arrays), intent(inout) :: this
    integer, intent(in) :: nelements, ndata, ndata_components, myrank_
    if (myrank_ == 0) print *, 'Allocating auxiliarily inversion arrays...'
    if (ndata <= 0 .or. nelements <= 0) call exit_MPI('Wrong dimensions in inversion_arrays_allocate_aux!', myrank_, 0)
    if (allocated(this%residuals)) deallocate(this%residuals)
    if (allocated(this%column_weight)) deallocate(this%column_weight)
    allocate(this%residuals(ndata_components, ndata), this%column_weight(nelements))
    this%residuals = 0.d0
    this%column_weight = 1.d0
  end subroutine inversion_arrays_allocate_aux

  subroutine inversion_arrays_reallocate_aux(this, nelements, ndata, ndata_components, myrank_)
    class(t_inversion_arrays), intent(inout) :: this
    integer, intent(in) :: nelements, ndata, ndata_components, myrank_
    if (.not. allocated(this%residuals)) then
      call this%allocate_aux(nelements, ndata, ndata_components, myrank_)
      return
    endif
    if (size(this%residuals, 1) /= ndata_components .or. size(this%residuals, 2) /= ndata) then
      deallocate(this%residuals)
      allocate(this%residuals(ndata_components, ndata))
      this%residuals = 0.d0
    endif
    ! like the reference (inversion_arrays.f90:78-95): the column weight is re-sized to the cells of the new partition;
    ! read_sensitivity_kernel fills it (the library keeps the full vector the kernel was built with)
    if (allocated(this%column_weight)) then
      if (size(this%column_weight) /= nelements) deallocate(this%column_weight)
    endif
    if (.not. allocated(this%column_weight)) then
      allocate(this%column_weight(nelements))
      this%column_weight = 1.d0
    endif
  end subroutine inversion_arrays_reallocate_aux

  !-------------------------------------------------------------------------------------------------------
  ! calculate_depth_weight, src/forward/gravmag/weights_gravmag.f90:46-196 (types 1, 2 and 3), the reference's argument list:
  ! the result goes to iarr%column_weight, (re)sized to all cells.
  subroutine calculate_depth_weight_iarr(par, iarr, grid_full, data, myrank_, nbproc_)
    class(t_parameters_base), intent(in) :: par
    type(t_inversion_arrays), intent(inout) :: iarr
    type(t_grid), intent(in) :: grid_full
    type(t_data), intent(in) :: data
    integer, intent(in) :: myrank_, nbproc_
    integer :: ntot
    ntot = par%nx * par%ny * par%nz
    if (allocated(iarr%column_weight)) then
      if (size(iarr%column_weight) /= ntot) deallocate(iarr%column_weight)
    endif
    if (.not. allocated(iarr%column_weight)) allocate(iarr%column_weight(ntot))
    call calculate_depth_weight_array(par, iarr%column_weight, grid_full, data, myrank_, nbproc_)
  end subroutine calculate_depth_weight_iarr

  ! The same with the plain array.  The multiplier of problem_joint_gravmag.F90:178 is applied by the caller, as in the reference.
  subroutine calculate_depth_weight_array(par, column_weight, grid_full, data, myrank_, nbproc_)
    class(t_parameters_base), intent(in) :: par
    real(dp), intent(out) :: column_weight(:)
    type(t_grid), intent(in) :: grid_full
    type(t_data), intent(in) :: data
    integer, intent(in) :: myrank_, nbproc_
    type(c_ptr) :: ctx
    ctx = tfx_api_context(myrank_, nbproc_)
    if (myrank_ == 0) print *, 'Calculating the depth weight, type = ', par%depth_weighting_type
    call api_check(tfx_set_grid(ctx, par%nx, par%ny, par%nz, grid_full%X1, grid_full%X2, grid_full%Y1, grid_full%Y2, grid_full%Z1, &
                                grid_full%Z2), 'tfx_set_grid', myrank_)
    if (par%depth_weighting_type == 1) then
      call api_check(tfx_column_weight_type1(ctx, par%depth_weighting_power, par%Z0, 1.d0, column_weight), 'calculate_depth_weight', myrank_)
    else if (par%depth_weighting_type == 2) then
      call api_check(tfx_column_weight_type2(ctx, int(data%ndata, c_int64_t), data%X, data%Y, data%Z, par%depth_weighting_power, &
                                             par%depth_weighting_beta, 1.d0, column_weight), 'calculate_depth_weight', myrank_)
    else if (par%depth_weighting_type == 3) then
      call api_check(tfx_column_weight_type3(ctx, int(data%ndata, c_int64_t), data%X, data%Y, data%Z, par%depth_weighting_power, &
                                             1.d0, column_weight), 'calculate_depth_weight', myrank_)
    else
      call exit_MPI('Not known depth weight type!', myrank_, par%depth_weighting_type)      ! weights_gravmag.f90:164
    endif
  end subroutine calculate_depth_weight_array

  !-------------------------------------------------------------------------------------------------------
  integer function problem_of(par)
    class(t_parameters_base), intent(in) :: par
    problem_of = 1
    select type(par)
    class is (t_parameters_mag)
      problem_of = 2
    end select
  end function problem_of

  ! the data are dealt out by the reference's own rule (calculate_nelements_at_cpu, src/utils/parallel_tools.f90:46-63, used for the
  ! rows in sensitivity_gravmag.F90:179-189): contiguous ranges of ndat / nbproc data, the remainder to the last rank - this
  ! rank's data (row_a, row_b].  (Round 2 dealt whole 2048-row blocks: 7 vs 6 blocks on 8 ranks at the headline size.)
  subroutine my_row_blocks(ndat, myrank_, nbproc_, row_a, row_b)
    integer, intent(in) :: ndat, myrank_, nbproc_
    integer, intent(out) :: row_a, row_b
    integer :: base
    base = ndat / nbproc_
    row_a = myrank_ * base
    row_b = row_a + base
    if (myrank_ == nbproc_ - 1) row_b = ndat
  end subroutine my_row_blocks

  ! device pointer + byte offset
  type(c_ptr) function ptr_plus(p, bytes)
    type(c_ptr), intent(in) :: p
    integer(c_int64_t), intent(in) :: bytes
    ptr_plus = transfer(transfer(p, 0_c_intptr_t) + int(bytes, c_intptr_t), ptr_plus)
  end function ptr_plus

  logical function write_sensit_rank_file_enabled(par, nnz_total)
    class(t_parameters_base), intent(in) :: par
    integer(c_int64_t), intent(in) :: nnz_total
    character(len=8) :: v
    integer :: l, st
    ! the reference always writes; here the kernel stays on the GPU, so the files are a checkpoint: on request, or by default
    ! while they are small (8 bytes per entry have to cross PCIe and reach the disk)
    write_sensit_rank_file_enabled = par%sensit_write == 1 .or. (par%sensit_write < 0 .and. nnz_total <= 250000000_c_int64_t)
    call get_environment_variable('TFX_WRITE_SENSIT', v, l, st)
    if (st == 0 .and. l > 0) write_sensit_rank_file_enabled = v(1:1) /= '0'
  end function write_sensit_rank_file_enabled

  !-------------------------------------------------------------------------------------------------------
  ! calculate_and_write_sensit, src/forward/gravmag/sensitivity_gravmag.F90:82-410.
  ! Rows are parallelised by data like the reference (:179-189); every (datum, data component, model component) line is
  ! weighted, transformed, thresholded and compacted on the GPU (:193-311).  The kernel is kept UNSCALED, as the reference's
  ! files hold it; problem_weight and data weights are applied by read_sensitivity_kernel (:834-843).
  subroutine calculate_and_write_sensit(par, grid_full, data, column_weight, memory, myrank_, nbproc_)
    class(t_parameters_base), intent(in) :: par
    type(t_grid), intent(in) :: grid_full
    type(t_data), intent(in) :: data
    real(dp), intent(in) :: column_weight(:)
    integer, intent(in) :: myrank_, nbproc_
    real(dp), intent(out) :: memory
    type(t_kernel_state), pointer :: k
    type(c_ptr) :: ctx, mptr
    integer :: ip, n, nd, ra, rb
    integer(c_int64_t) :: nnz_k, nr, nc, nz, dbytes
    real(c_double) :: err_k
    real(dp), target, save :: mag_field(4)
    integer(c_int32_t), allocatable, target :: hist(:)
    real(dp) :: s1(1)
    logical :: exchange
    character(len=32) :: v
    integer :: l, st

    if (par%compression_rate < 0 .or. par%compression_rate > 1) &
      call exit_MPI('Wrong compression rate! It must be between 0 and 1.', myrank_, 0)                 ! :119-121
    ip = problem_of(par)
    k => kst(ip)
    ctx = tfx_api_context(myrank_, nbproc_)
    n = par%nx * par%ny * par%nz
    nd = par%ndata
    if (size(column_weight) /= n) call exit_MPI('calculate_and_write_sensit needs the full column weight!', myrank_, size(column_weight))
    mptr = c_null_ptr
    if (ip == 1) then
      if (myrank_ == 0) print *, 'Calculating GRAVITY sensitivity kernel...'
    else
      if (myrank_ == 0) print *, 'Calculating MAGNETIC sensitivity kernel...'
      select type(par)
      class is (t_parameters_mag)
        mag_field = (/par%mi, par%md, par%theta, par%intensity/)
      end select
      mptr = c_loc(mag_field)
    endif
    if (.not. k%built) then
      k%slot = nslots_used
      nslots_used = nslots_used + 1
    endif
    k%data_type = merge(par%data_type, 1, ip == 1)
    k%ndc = par%ndata_components
    k%ncm = par%nmodel_components
    k%mag_field = mag_field
    k%cw_full = column_weight
    k%Xd = data%X; k%Yd = data%Y; k%Zd = data%Z
    allocate(hist(n))
    hist = 0
    call api_check(tfx_select_problem(ctx, int(k%slot, c_int)), 'tfx_select_problem', myrank_)
    call api_check(tfx_set_grid(ctx, par%nx, par%ny, par%nz, grid_full%X1, grid_full%X2, grid_full%Y1, grid_full%Y2, grid_full%Z1, &
                                grid_full%Z2), 'tfx_set_grid', myrank_)
    err_k = 0.d0
    if (nbproc_ == 1) then
      ! one rank owns every column: the rows go straight into the tiled matrix
      k%mode = 1
      k%row_a = 0; k%row_b = nd
      call api_check(tfx_build_kernel(ctx, ip, k%data_type, k%ndc, k%ncm, int(nd, c_int64_t), data%X, data%Y, data%Z, column_weight, &
                                      mptr, par%compression_type, par%compression_rate, 1.d0, c_null_ptr, 0_c_int64_t, &
                                      int(n, c_int64_t), nnz_k, err_k, c_loc(hist)), 'calculate_and_write_sensit', myrank_)
      k%nnz_total = nnz_k
    else
      ! row-parallel: my blocks of ROW_BLOCK data (= ndc row blocks of the matrix each) with ALL their columns stay in the device
      ! row store until the partition is known (compressed kernels; a dense kernel is built per column range on reload)
      exchange = par%compression_type > 0
      call get_environment_variable('TFX_BUILD_MODE', v, l, st)
      if (st == 0 .and. l > 0) then
        if (v(1:l) == 'redundant') exchange = .false.
      endif
      if (exchange) then
        k%mode = 2
        if (myrank_ == 0) print *, '(row-parallel build: every rank compresses its row blocks, the pieces are re-laid out GPU to GPU)'
        call my_row_blocks(nd, myrank_, nbproc_, ra, rb)
        if (rb > ra) call api_check(tfx_rowstore_build_comp(ctx, ip, k%data_type, k%ndc, k%ncm, int(rb - ra, c_int64_t), data%X(ra + 1:rb), &
                                                           data%Y(ra + 1:rb), data%Z(ra + 1:rb), column_weight, mptr, par%compression_type, &
                                                           par%compression_rate, 1.d0, c_null_ptr, nnz_k, err_k, c_loc(hist)), &
                                    'calculate_and_write_sensit', myrank_)
      else
        k%mode = 3
        ra = (nd / nbproc_) * myrank_                                         ! calculate_nelements_at_cpu (parallel_tools.f90:46-63)
        rb = ra + nd / nbproc_
        if (myrank_ == nbproc_ - 1) rb = nd
        if (rb > ra) call api_check(tfx_build_kernel(ctx, ip, k%data_type, k%ndc, k%ncm, int(rb - ra, c_int64_t), data%X(ra + 1:rb), &
                                                    data%Y(ra + 1:rb), data%Z(ra + 1:rb), column_weight, mptr, par%compression_type, &
                                                    par%compression_rate, 1.d0, c_null_ptr, 0_c_int64_t, 0_c_int64_t, nnz_k, err_k, &
                                                    c_loc(hist)), 'calculate_and_write_sensit', myrank_)
      endif
      k%row_a = ra; k%row_b = rb
      call allreduce_sum_i32(hist, n)                                          ! :322
      s1(1) = err_k
      call allreduce_sum_dp(s1, 1)
      err_k = s1(1)
      k%nnz_total = sum(int(hist, c_int64_t))
    endif
    k%nnz_hist = hist
    k%err_sum = err_k
    k%built = .true.
    if (myrank_ == 0) then
      print *, 'nnz_total = ', k%nnz_total                                                              ! :340-358
      print *, 'COMPRESSION RATE = ', dble(k%nnz_total) / dble(n) / dble(nd) / dble(k%ncm) / dble(k%ndc)
      print *, 'COMPRESSION ERROR, r = ', err_k / dble(nd * k%ndc * k%ncm)
    endif
    ! the SENSIT folder (:142-153, :183, :306-309, :360-392, :415-464): a checkpoint here, not a hand-over
    if (write_sensit_rank_file_enabled(par, k%nnz_total)) call write_sensit_files(par, ip, myrank_, nbproc_)
    memory = 0.d0
    if (k%mode == 1) then
      if (tfx_matrix_info(ctx, nr, nc, nz, dbytes) == 0) memory = dble(dbytes) / 1024.d0**3
    endif
    if (myrank_ == 0) print *, 'Finished calculating the sensitivity kernel.'
  end subroutine calculate_and_write_sensit

  !-------------------------------------------------------------------------------------------------------
  ! The reference's SENSIT files from what the device holds: every rank writes the row file of ITS rows
  ! (sensit_{grav|magn}_{nbproc}_{rank}: header :183, one record per (datum, data comp, model comp) :306-309), rank 0 the metadata
  ! (:360-375), the per-cell counts (:380-392) and the depth weight (:415-464).  Unscaled values: bit-identical to the reference's.
  subroutine write_sensit_files(par, ip, myrank_, nbproc_)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: ip, myrank_, nbproc_
    type(t_kernel_state), pointer :: k
    integer :: u, n, i, d, kc, r, nrl
    integer(c_int64_t) :: nrows, ncols, nnz, dbytes, a, b, e, p, cap, got
    integer(c_int64_t), allocatable :: rowptr(:)
    integer(c_int32_t), allocatable, target :: cols(:), cnt(:, :)
    real(c_float), allocatable, target :: vals(:)
    integer(c_int64_t) :: bounds(2)
    type(c_ptr) :: dc, dv
    character(len=512) :: fname, folder
    character(len=32) :: tag
    integer, parameter :: RCHUNK = 256
    integer :: r0, r1, nel
    k => kst(ip)
    if (k%mode == 3) return                       ! nothing is stored yet in the count-only mode
    n = par%nx * par%ny * par%nz
    folder = trim(par%sensit_path)
    call execute_command_line('mkdir -p "'//trim(folder)//'"')
    write(tag, '(I0,A,I0)') nbproc_, '_', myrank_
    fname = trim(folder)//'/sensit_'//SENSIT_SUFFIX(ip)//'_'//trim(tag)
    if (myrank_ == 0) print *, 'Writing the sensitivity to file ', trim(fname)
    open(newunit=u, file=trim(fname), status='replace', access='stream', form='unformatted', action='write', convert='big_endian')
    nrl = k%row_b - k%row_a
    write(u) int(nrl, c_int32_t), int(par%ndata, c_int32_t), int(n, c_int32_t), int(myrank_, c_int32_t), int(nbproc_, c_int32_t)
    if (k%mode == 1) then
      ! single rank: the tiled matrix back as CSR (row = (datum, data component), columns (k-1)*n + cell ascending, :829-846)
      call api_check(tfx_matrix_info(api_ctx, nrows, ncols, nnz, dbytes), 'tfx_matrix_info', myrank_)
      allocate(rowptr(nrows + 1), cols(max(nnz, 1_c_int64_t)), vals(max(nnz, 1_c_int64_t)))
      call api_check(tfx_matrix_download_csr(api_ctx, rowptr, cols, vals), 'tfx_matrix_download_csr', myrank_)
      r = 0
      do i = 1, par%ndata
        do d = 1, k%ndc
          r = r + 1
          a = rowptr(r) + 1
          b = rowptr(r + 1)
          do kc = 1, k%ncm
            e = a
            do while (e <= b)
              if (cols(e) > kc * n) exit
              e = e + 1
            enddo
            write(u) int(i, c_int32_t), int(e - a, c_int32_t), int(kc, c_int32_t), int(d, c_int32_t)
            if (e > a) then
              do p = a, e - 1
                cols(p) = cols(p) - (kc - 1) * n
              enddo
              write(u) cols(a:e - 1), vals(a:e - 1)
            endif
            a = e
          enddo
        enddo
      enddo
    else
      ! several ranks: my rows out of the device row store, RCHUNK matrix rows at a time (local row = (i-1)*ndc + d)
      bounds = (/0_c_int64_t, int(n, c_int64_t)/)
      nrows = int(nrl, c_int64_t) * k%ndc
      allocate(cnt(1, max(nrows, 1_c_int64_t)))
      if (nrows > 0) call api_check(tfx_rowstore_counts(api_ctx, 1_c_int, bounds, cnt), 'tfx_rowstore_counts', myrank_)
      do r0 = 0, int(nrows) - 1, RCHUNK
        r1 = min(r0 + RCHUNK, int(nrows))
        cap = sum(int(cnt(1, r0 + 1:r1), c_int64_t))
        allocate(cols(max(cap, 1_c_int64_t)), vals(max(cap, 1_c_int64_t)))
        if (cap > 0) then
          call api_check(tfx_device_malloc(api_ctx, 4 * cap, dc), 'tfx_device_malloc', myrank_)
          call api_check(tfx_device_malloc(api_ctx, 4 * cap, dv), 'tfx_device_malloc', myrank_)
          call api_check(tfx_rowstore_pack(api_ctx, int(r0, c_int64_t), int(r1 - r0, c_int64_t), 0_c_int64_t, int(n, c_int64_t), dc, dv, &
                                           cap, got), 'tfx_rowstore_pack', myrank_)
          call api_check(tfx_copy(api_ctx, c_loc(cols), dc, 4 * cap), 'tfx_copy', myrank_)
          call api_check(tfx_copy(api_ctx, c_loc(vals), dv, 4 * cap), 'tfx_copy', myrank_)
          call api_check(tfx_device_free(api_ctx, dc), 'tfx_device_free', myrank_)
          call api_check(tfx_device_free(api_ctx, dv), 'tfx_device_free', myrank_)
        endif
        a = 0
        do r = r0 + 1, r1
          nel = cnt(1, r)
          e = a
          do kc = 1, k%ncm                             ! the packed row holds component kc at 0-based columns (kc-1)*n + cell
            p = e
            do while (p < a + nel)
              if (cols(p + 1) >= kc * n) exit
              p = p + 1
            enddo
            write(u) int(k%row_a + (r - 1) / k%ndc + 1, c_int32_t), int(p - e, c_int32_t), int(kc, c_int32_t), &
                     int(mod(r - 1, k%ndc) + 1, c_int32_t)
            if (p > e) then
              cols(e + 1:p) = cols(e + 1:p) - (kc - 1) * n + 1
              write(u) cols(e + 1:p), vals(e + 1:p)
            endif
            e = p
          enddo
          a = a + nel
        enddo
        deallocate(cols, vals)
      enddo
    endif
    close(u)
    if (myrank_ == 0) then
      fname = trim(folder)//'/sensit_'//SENSIT_SUFFIX(ip)//'_meta.txt'
      open(newunit=u, file=trim(fname), form='formatted', status='replace', action='write')
      write(u, *) par%nx, par%ny, par%nz, par%ndata
      write(u, *) nbproc_, 4, par%depth_weighting_type
      write(u, *) par%compression_type, k%err_sum / dble(par%ndata * k%ndc * k%ncm)
      write(u, *) k%ncm, k%ndc
      write(u, *) k%nnz_total
      close(u)
      fname = trim(folder)//'/sensit_'//SENSIT_SUFFIX(ip)//'_nnz'
      open(newunit=u, file=trim(fname), status='replace', access='stream', form='unformatted', action='write', convert='big_endian')
      write(u) int(n, c_int32_t)
      write(u) k%nnz_hist
      close(u)
      fname = trim(folder)//'/sensit_'//SENSIT_SUFFIX(ip)//'_weight'
      open(newunit=u, file=trim(fname), status='replace', access='stream', form='unformatted', action='write', convert='big_endian')
      write(u) int(n, c_int32_t)
      write(u) k%cw_full
      close(u)
    endif
  end subroutine write_sensit_files

  !-------------------------------------------------------------------------------------------------------
  ! calculate_new_partitioning, sensitivity_gravmag.F90:573-640 (+ get_load_balancing_nelements :470-524, exact integer rule in
  ! tfx_partition_columns).  problem_type 1 grav, 2 magn, 3 joint: the counts of both kernels added (:598-606).
  subroutine calculate_new_partitioning(par, nnz, nelements_at_cpu, problem_type, myrank_, nbproc_)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: problem_type, myrank_, nbproc_
    integer(c_int64_t), intent(out) :: nnz
    integer, intent(out) :: nelements_at_cpu(nbproc_)
    integer(c_int64_t) :: nnz_at_cpu(nbproc_)
    integer(c_int32_t), allocatable :: sensit_nnz(:), nel32(:)
    integer :: n, ip
    n = par%nx * par%ny * par%nz
    allocate(sensit_nnz(n), nel32(nbproc_))
    sensit_nnz = 0
    do ip = 1, 2
      if (problem_type /= 3 .and. problem_type /= ip) cycle
      if (par%sensit_read == 1) then
        call read_sensit_nnz(par, ip, n, sensit_nnz, myrank_)                  ! :530-568
      else
        if (.not. kst(ip)%built) call exit_MPI('calculate_new_partitioning: the kernel has not been calculated!', myrank_, ip)
        sensit_nnz = sensit_nnz + kst(ip)%nnz_hist
      endif
    enddo
    call api_check(tfx_partition_columns(sensit_nnz, int(n, c_int64_t), int(nbproc_, c_int), nel32, nnz_at_cpu), &
                   'calculate_new_partitioning', myrank_)
    nelements_at_cpu = nel32
    nnz = nnz_at_cpu(myrank_ + 1)
    part_nel = nelements_at_cpu
    part_cb = sum(nelements_at_cpu(1:myrank_))
    part_ce = part_cb + nelements_at_cpu(myrank_ + 1)
    partitioned = .true.
    if (myrank_ == 0) then
      print *, 'nelements_at_cpu =', nelements_at_cpu
      print *, 'nnz_at_cpu =', nnz_at_cpu
    endif
  end subroutine calculate_new_partitioning

  subroutine read_sensit_nnz(par, ip, n, sensit_nnz, myrank_)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: ip, n, myrank_
    integer(c_int32_t), intent(inout) :: sensit_nnz(n)
    integer(c_int32_t), allocatable :: h(:)
    integer(c_int32_t) :: nread
    integer :: u, ios
    open(newunit=u, file=trim(par%sensit_path)//'sensit_'//SENSIT_SUFFIX(ip)//'_nnz', status='old', access='stream', &
         form='unformatted', action='read', convert='big_endian', iostat=ios)
    if (ios /= 0) call exit_MPI('Error in opening the sensit_nnz file!', myrank_, ios)
    read(u) nread
    if (nread /= n) call exit_MPI('Wrong file header in calculate_new_partitioning!', myrank_, int(nread))
    allocate(h(n))
    read(u) h
    close(u)
    sensit_nnz = sensit_nnz + h
  end subroutine read_sensit_nnz

  !-------------------------------------------------------------------------------------------------------
  ! read_sensitivity_kernel, sensitivity_gravmag.F90:648-883: the kernel of one problem into the (joint) sensitivity matrix,
  ! parallelised by model cells; rows scaled by problem_weight * data_weight (:834-843); column_weight(par%nelements) returned
  ! like the reference returns the slice of the weight file (:920-970).
  subroutine read_sensitivity_kernel(par, sensit_matrix, column_weight, problem_weight, data_weight, problem_type, &
                                     myrank_, nbproc_, nelements_at_cpu)
    class(t_parameters_base), intent(in) :: par
    type(t_sparse_matrix), intent(inout) :: sensit_matrix
    real(dp), intent(out) :: column_weight(:)
    real(dp), intent(in) :: problem_weight
    real(dp), intent(in) :: data_weight(par%ndata_components, par%ndata)
    integer, intent(in) :: problem_type, myrank_, nbproc_
    integer, intent(in) :: nelements_at_cpu(nbproc_)
    type(t_kernel_state), pointer :: k
    type(c_ptr) :: ctx, mptr
    integer :: ip, n, cb, ce, nloc, i, d, r
    integer(c_int64_t) :: nnz_k, nr, nc, nz, dbytes
    real(c_double) :: err_k
    real(dp), allocatable, target :: scale(:), dwflat(:)
    real(dp), target, save :: mag_field(4)
    logical :: unit_scale
    ip = problem_type
    k => kst(ip)
    ctx = tfx_api_context(myrank_, nbproc_)
    n = par%nx * par%ny * par%nz
    cb = sum(nelements_at_cpu(1:myrank_))
    nloc = nelements_at_cpu(myrank_ + 1)
    ce = cb + nloc
    if (size(column_weight) /= nloc) call exit_MPI('read_sensitivity_kernel: column_weight must hold par%nelements values!', myrank_, nloc)
    if (myrank_ == 0) print *, 'Reading the sensitivity kernel...'
    allocate(scale(par%ndata * par%ndata_components))
    unit_scale = .true.
    r = 0
    do i = 1, par%ndata
      do d = 1, par%ndata_components
        r = r + 1
        scale(r) = problem_weight * data_weight(d, i)
        if (scale(r) /= 1.d0) unit_scale = .false.
      enddo
    enddo
    if (par%sensit_read == 1) then
      ! a kernel written by an earlier run (by this host or by the reference itself)
      if (.not. k%built .or. k%slot < 0) then
        k%slot = nslots_used
        nslots_used = nslots_used + 1
      endif
      call api_check(tfx_select_problem(ctx, int(k%slot, c_int)), 'tfx_select_problem', myrank_)
      call read_sensit_files(par, ip, scale, cb, ce, myrank_)
      allocate(k%cw_full(n))
      call read_weight_file(par, ip, n, k%cw_full, myrank_)
      k%built = .true.
      k%mode = 1
    else
      if (.not. k%built) call exit_MPI('read_sensitivity_kernel: the kernel has not been calculated!', myrank_, ip)
      call api_check(tfx_select_problem(ctx, int(k%slot, c_int)), 'tfx_select_problem', myrank_)
      select case (k%mode)
      case (1)                                              ! already the tiled matrix of this rank: only the scaling is left
        if (nbproc_ /= 1) call exit_MPI('read_sensitivity_kernel: inconsistent build mode!', myrank_, k%mode)
        if (.not. unit_scale) call api_check(tfx_matrix_scale_rows(ctx, scale), 'read_sensitivity_kernel', myrank_)
      case (2)                                              ! row store -> column ranges, GPU to GPU
        call relayout_rowstore(par, k, nelements_at_cpu, myrank_, nbproc_)
        if (.not. unit_scale) call api_check(tfx_matrix_scale_rows(ctx, scale), 'read_sensitivity_kernel', myrank_)
      case (3)                                              ! every rank builds all rows for its own columns
        mptr = c_null_ptr
        mag_field = k%mag_field
        if (ip == 2) mptr = c_loc(mag_field)
        allocate(dwflat(par%ndata * par%ndata_components))
        dwflat = reshape(data_weight, (/par%ndata * par%ndata_components/))
        ! (the histogram of the counting pass says what this rank's columns hold: not the rows x K of a whole kernel)
        call api_check(tfx_matrix_reserve(ctx, sum(int(k%nnz_hist(cb + 1:ce), c_int64_t))), 'tfx_matrix_reserve', myrank_)
        call api_check(tfx_build_kernel(ctx, ip, k%data_type, k%ndc, k%ncm, int(par%ndata, c_int64_t), k%Xd, k%Yd, k%Zd, k%cw_full, mptr, &
                                        par%compression_type, par%compression_rate, problem_weight, c_loc(dwflat), int(cb, c_int64_t), &
                                        int(ce, c_int64_t), nnz_k, err_k, c_null_ptr), 'read_sensitivity_kernel', myrank_)
      end select
    endif
    column_weight = k%cw_full(cb + 1:ce)
    call api_check(tfx_matrix_info(ctx, nr, nc, nz, dbytes), 'tfx_matrix_info', myrank_)
    sensit_matrix%on_device = .true.
    sensit_matrix%nproblems = sensit_matrix%nproblems + 1
    sensit_matrix%nl_device = sensit_matrix%nl_device + int(nr)
    sensit_matrix%ncolumns_device = sensit_matrix%ncolumns_device + int(nc)
    call api_check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem', myrank_)
    if (myrank_ == 0) print *, 'Finished reading the sensitivity kernel.'
  end subroutine read_sensitivity_kernel

  ! read_sensitivity_kernel's relayout (sensitivity_gravmag.F90:795-830) without the files: every row block is cut by column range
  ! on its owner's GPU and the pieces go to the owners of the columns, which lay them out as tiles
  subroutine relayout_rowstore(par, k, nel_at, myrank_, nbproc_)
    class(t_parameters_base), intent(in) :: par
    type(t_kernel_state), intent(inout) :: k
    integer, intent(in) :: nel_at(:), myrank_, nbproc_
    integer :: ndat, row_a, row_b, nrl, nblk, b, ga, gb, sa, sb, o, d, rr, nloc, ndc, ra, rb
    integer, allocatable :: rows_at(:), row_displs(:)
    integer(c_int64_t), allocatable :: bounds(:)
    integer(c_int32_t), allocatable, target :: cnt_loc(:, :), cnt_all(:, :), nel_blk(:)
    integer(c_int64_t) :: n_in, n_out, n_piece, off, got, mine
    type(c_ptr) :: dcols, dvals, scols, svals
    ! matrix rows: ndc per datum (row = (i-1)*ndc + d); rank rr built the rows [row_displs(rr+1), +rows_at(rr+1)) - any contiguous
    ! range (my_row_blocks), so a row block of the matrix may have two (rarely more) contributors: their pieces are contiguous row
    ! sub-ranges in rank order and land one behind the other in the receive buffer, which then holds the packed row block
    ndc = k%ndc
    ndat = par%ndata * ndc                       ! matrix rows of the kernel
    row_a = k%row_a * ndc; row_b = k%row_b * ndc
    nrl = row_b - row_a
    nloc = nel_at(myrank_ + 1)
    nblk = (ndat + ROW_BLOCK - 1) / ROW_BLOCK
    allocate(rows_at(nbproc_), row_displs(nbproc_), bounds(nbproc_ + 1))
    do rr = 0, nbproc_ - 1
      call my_row_blocks(par%ndata, rr, nbproc_, ra, rb)
      rows_at(rr + 1) = (rb - ra) * ndc
      row_displs(rr + 1) = ra * ndc
    enddo
    bounds(1) = 0
    do rr = 1, nbproc_
      bounds(rr + 1) = bounds(rr) + nel_at(rr)
    enddo
    allocate(cnt_loc(nbproc_, max(nrl, 1)), cnt_all(nbproc_, ndat))
    if (nrl > 0) call api_check(tfx_rowstore_counts(api_ctx, int(nbproc_, c_int), bounds, cnt_loc), 'tfx_rowstore_counts', myrank_)
    call allgather_counts(cnt_loc, nbproc_, nrl, cnt_all, rows_at, row_displs)
    mine = 0
    do rr = 1, ndat
      mine = mine + cnt_all(myrank_ + 1, rr)
    enddo
    call api_check(tfx_matrix_begin(api_ctx, int(ndat, c_int64_t), int(k%ncm * nloc, c_int64_t), max(mine, 1_c_int64_t)), 'tfx_matrix_begin', myrank_)
    do b = 1, nblk
      ga = (b - 1) * ROW_BLOCK
      gb = min(b * ROW_BLOCK, ndat)
      n_in = 0
      do rr = ga + 1, gb
        n_in = n_in + cnt_all(myrank_ + 1, rr)
      enddo
      call api_check(tfx_device_malloc(api_ctx, 4 * max(n_in, 1_c_int64_t), dcols), 'tfx_device_malloc', myrank_)
      call api_check(tfx_device_malloc(api_ctx, 4 * max(n_in, 1_c_int64_t), dvals), 'tfx_device_malloc', myrank_)
      off = 0
      do o = 0, nbproc_ - 1
        sa = max(ga, row_displs(o + 1))
        sb = min(gb, row_displs(o + 1) + rows_at(o + 1))
        if (sb <= sa) cycle
        n_piece = 0                                              ! what I receive of rank o's rows of this block
        do rr = sa + 1, sb
          n_piece = n_piece + cnt_all(myrank_ + 1, rr)
        enddo
        if (o == myrank_) then
          do d = 0, nbproc_ - 1
            n_out = 0
            do rr = sa + 1, sb
              n_out = n_out + cnt_all(d + 1, rr)
            enddo
            if (n_out == 0) cycle
            if (d == myrank_) then
              call api_check(tfx_rowstore_pack(api_ctx, int(sa - row_a, c_int64_t), int(sb - sa, c_int64_t), bounds(d + 1), bounds(d + 2), &
                                               ptr_plus(dcols, 4 * off), ptr_plus(dvals, 4 * off), n_out, got), 'tfx_rowstore_pack', myrank_)
            else
              call api_check(tfx_device_malloc(api_ctx, 4 * n_out, scols), 'tfx_device_malloc', myrank_)
              call api_check(tfx_device_malloc(api_ctx, 4 * n_out, svals), 'tfx_device_malloc', myrank_)
              call api_check(tfx_rowstore_pack(api_ctx, int(sa - row_a, c_int64_t), int(sb - sa, c_int64_t), bounds(d + 1), bounds(d + 2), &
                                               scols, svals, n_out, got), 'tfx_rowstore_pack', myrank_)
              call exchange_piece_send(api_ctx, d, n_out, scols, svals, 2 * b)
              call api_check(tfx_device_free(api_ctx, scols), 'tfx_device_free', myrank_)
              call api_check(tfx_device_free(api_ctx, svals), 'tfx_device_free', myrank_)
            endif
          enddo
        else if (n_piece > 0) then
          call exchange_piece_recv(api_ctx, o, n_piece, ptr_plus(dcols, 4 * off), ptr_plus(dvals, 4 * off), 2 * b)
        endif
        off = off + n_piece
      enddo
      allocate(nel_blk(gb - ga))
      nel_blk = cnt_all(myrank_ + 1, ga + 1:gb)
      call api_check(tfx_matrix_append_rows(api_ctx, int(ga, c_int64_t), int(gb - ga, c_int64_t), dcols, dvals, nel_blk), &
                     'tfx_matrix_append_rows', myrank_)
      deallocate(nel_blk)
      call api_check(tfx_device_free(api_ctx, dcols), 'tfx_device_free', myrank_)
      call api_check(tfx_device_free(api_ctx, dvals), 'tfx_device_free', myrank_)
    enddo
    call api_check(tfx_matrix_finish(api_ctx), 'tfx_matrix_finish', myrank_)
    call api_check(tfx_rowstore_free(api_ctx), 'tfx_rowstore_free', myrank_)
    k%mode = 1
  end subroutine relayout_rowstore

  subroutine read_weight_file(par, ip, n, cw, myrank_)                                                   ! :920-970
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: ip, n, myrank_
    real(dp), intent(out) :: cw(n)
    integer :: u, ios
    integer(c_int32_t) :: nread
    open(newunit=u, file=trim(par%sensit_path)//'sensit_'//SENSIT_SUFFIX(ip)//'_weight', status='old', access='stream', &
         form='unformatted', action='read', convert='big_endian', iostat=ios)
    if (ios /= 0) call exit_MPI('Error in opening the depth weight file! path='//trim(par%sensit_path), myrank_, ios)
    read(u) nread
    if (nread /= n) call exit_MPI('Depth weight file header does not match the Parfile!', myrank_, int(nread))
    read(u) cw
    close(u)
  end subroutine read_weight_file

  ! read_sensitivity_metadata + the row files of any number of writer ranks (:648-883, :975-1030) -> CSR of this rank's cells
  subroutine read_sensit_files(par, ip, scale, c0, c1, myrank_)
    class(t_parameters_base), intent(in) :: par
    integer, intent(in) :: ip, c0, c1, myrank_
    real(dp), intent(in) :: scale(:)
    integer :: u, ios, n, rank, nbproc_sensit, nloc, nd, ndc, nc
    integer(c_int64_t) :: jj, precision_read, wtype, ctype, ncm_read, ncd_read, nxr, nyr, nzr, ndr
    integer :: i, d, kc, r, idata_glob
    real(dp) :: comp_error
    integer(c_int64_t) :: nnz_total, pos, j
    integer(c_int32_t) :: hdr(5), desc(4)
    integer(c_int64_t), allocatable :: rowptr(:)
    integer(c_int32_t), allocatable :: cols(:)
    real(c_float), allocatable :: vals(:)
    character(len=512) :: fname
    character(len=16) :: s1, s2
    n = par%nx * par%ny * par%nz
    nd = par%ndata; ndc = par%ndata_components; nc = par%nmodel_components
    nloc = c1 - c0
    fname = trim(par%sensit_path)//'sensit_'//SENSIT_SUFFIX(ip)//'_meta.txt'
    if (myrank_ == 0) print *, 'Reading the sensitivity metadata file ', trim(fname)
    open(newunit=u, file=trim(fname), form='formatted', status='old', action='read', iostat=ios)
    if (ios /= 0) call exit_MPI('Error in opening the sensitivity metadata file! path='//trim(fname), myrank_, ios)
    read(u, *) nxr, nyr, nzr, ndr
    read(u, *) nbproc_sensit, precision_read, wtype
    read(u, *) ctype, comp_error
    read(u, *) ncm_read, ncd_read
    read(u, *) nnz_total
    close(u)
    if (myrank_ == 0) print *, 'COMPRESSION ERROR (read) =', comp_error
    if (nxr /= par%nx .or. nyr /= par%ny .or. nzr /= par%nz .or. ndr /= nd .or. wtype /= par%depth_weighting_type .or. &
        ncm_read /= nc .or. ncd_read /= ndc) call exit_MPI('Sensitivity metadata file info does not match the Parfile!', myrank_, 0)   ! :1001-1006
    if (ctype /= par%compression_type) call exit_MPI('Compression type is inconsistent!', myrank_, 0)
    if (precision_read /= 4) call exit_MPI('Matrix precision is not consistent!', myrank_, 0)
    allocate(rowptr(nd * ndc + 1), cols(max(nnz_total, 1_c_int64_t)), vals(max(nnz_total, 1_c_int64_t)))
    rowptr(1) = 0
    pos = 0
    r = 0
    idata_glob = 0
    do rank = 0, nbproc_sensit - 1
      write(s1, '(I0)') nbproc_sensit
      write(s2, '(I0)') rank
      fname = trim(par%sensit_path)//'sensit_'//SENSIT_SUFFIX(ip)//'_'//trim(s1)//'_'//trim(s2)
      if (rank == 0 .and. myrank_ == 0) print *, 'Reading the sensitivity file (new) ', trim(fname)
      open(newunit=u, file=trim(fname), status='old', access='stream', form='unformatted', action='read', convert='big_endian', &
           iostat=ios)
      if (ios /= 0) call exit_MPI('Error in opening the sensitivity file! path='//trim(fname), myrank_, ios)
      read(u) hdr
      if (hdr(2) /= nd .or. hdr(3) /= n .or. hdr(4) /= rank .or. hdr(5) /= nbproc_sensit) &
        call exit_MPI('Wrong file header in read_sensitivity_kernel!', myrank_, 0)                       ! :744-747
      do i = 1, hdr(1)
        idata_glob = idata_glob + 1
        do d = 1, ndc
          r = r + 1
          do kc = 1, nc
            read(u) desc
            if (desc(1) /= idata_glob) call exit_MPI('Wrong data index in read_sensitivity_kernel!', myrank_, int(desc(1)))
            if (desc(3) /= kc) call exit_MPI('Wrong model component index in read_sensitivity_kernel!', myrank_, int(desc(3)))
            if (desc(4) /= d) call exit_MPI('Wrong data component index in read_sensitivity_kernel!', myrank_, int(desc(4)))
            if (pos + desc(2) > nnz_total) call exit_MPI('Wrong number of elements in read_sensitivity_kernel!', myrank_, 0)
            if (desc(2) > 0) then
              read(u) cols(pos + 1:pos + desc(2)), vals(pos + 1:pos + desc(2))
              jj = pos
              do j = pos + 1, pos + desc(2)
                if (cols(j) <= c0 .or. cols(j) > c1) cycle
                jj = jj + 1
                cols(jj) = cols(j) - c0 + (kc - 1) * nloc                                                ! :832
                vals(jj) = vals(j) * real(scale(r), c_float)                                            ! :835-843
              enddo
              pos = jj
            endif
          enddo
          rowptr(r + 1) = pos
        enddo
      enddo
      close(u)
    enddo
    if (idata_glob /= nd .or. (nloc == n .and. pos /= nnz_total)) call exit_MPI('The SENSIT files do not hold the whole kernel!', myrank_, 0)
    if (myrank_ == 0) print *, 'nnz_total (of the read kernel)  = ', nnz_total
    call api_check(tfx_matrix_upload_csr(api_ctx, int(nd * ndc, c_int64_t), int(nloc, c_int64_t) * nc, rowptr, cols, vals), &
                   'tfx_matrix_upload_csr', myrank_)
  end subroutine read_sensit_files

  !-------------------------------------------------------------------------------------------------------
  ! model_calculate_data, src/inversion/model.F90:220-307: d = S Wav(m / column_weight) / problem_weight / data_weight.
  ! this%val and column_weight hold this rank's cells; line_start / param_shift select the problem's block of the joint matrix
  ! (part_mult_vector, sparse_matrix.f90:335-367): block 0 starts at (0, 0), the second kernel's at (rows, columns) of the first.
  subroutine model_calculate_data(this, ndata, ndata_components, matrix_sensit, problem_weight, column_weight, data_weight, &
                                  data_calc, compression_type, line_start, param_shift, myrank_, nbproc_)
    class(t_model), intent(in) :: this
    integer, intent(in) :: ndata, ndata_components, compression_type
    integer, intent(in) :: line_start, param_shift
    integer, intent(in) :: myrank_, nbproc_
    real(dp), intent(in) :: problem_weight
    type(t_sparse_matrix), intent(in) :: matrix_sensit
    real(dp), intent(in) :: column_weight(this%nelements)
    real(dp), intent(in), target :: data_weight(ndata_components, ndata)
    real(dp), intent(out) :: data_calc(ndata_components, ndata)
    real(dp), allocatable :: model_scaled(:, :), model_scaled_full(:)
    type(c_ptr) :: ctx
    integer :: i, k, slot, cb
    if (.not. matrix_sensit%on_device) call exit_MPI('model_calculate_data: the sensitivity kernel is not loaded!', myrank_, 0)
    ctx = tfx_api_context(myrank_, nbproc_)
    allocate(model_scaled(this%nelements, this%ncomponents))
    do k = 1, this%ncomponents                                                                          ! :240-249
      do i = 1, this%nelements
        if (column_weight(i) /= 0.d0) then
          model_scaled(i, k) = this%val(i, k) / column_weight(i)
        else
          model_scaled(i, k) = 0.d0
        endif
      enddo
    enddo
    if (compression_type > 0) then                                                                      ! :251-281
      if (nbproc_ > 1) then
        ! apply_wavelet_transform (wavelet_utils.F90:37-72): the slices of all ranks -> full model -> transform -> my slice
        cb = merge(part_cb, 0, partitioned)
        allocate(model_scaled_full(this%nelements_total))
        do k = 1, this%ncomponents
          call get_full_array(model_scaled(:, k), this%nelements, model_scaled_full, myrank_, nbproc_)
          call api_check(tfx_wavelet(ctx, model_scaled_full, this%grid_full%nx, this%grid_full%ny, this%grid_full%nz, 1_c_int64_t, &
                                     compression_type, 1_c_int), 'forward_wavelet', myrank_)
          model_scaled(:, k) = model_scaled_full(cb + 1:cb + this%nelements)
        enddo
      else
        call api_check(tfx_wavelet(ctx, model_scaled, this%grid_full%nx, this%grid_full%ny, this%grid_full%nz, &
                                   int(this%ncomponents, c_int64_t), compression_type, 1_c_int), 'forward_wavelet', myrank_)
      endif
    endif
    ! which kernel: the one whose rows start at line_start (0 = the first loaded, else the second)
    slot = merge(0, 1, line_start == 0 .and. param_shift == 0)
    call api_check(tfx_select_problem(ctx, int(slot, c_int)), 'tfx_select_problem', myrank_)
    call api_check(tfx_calc_data(ctx, model_scaled, problem_weight, c_loc(data_weight), data_calc), 'model_calculate_data', myrank_)   ! :285-302
    call api_check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem', myrank_)
  end subroutine model_calculate_data

  !-------------------------------------------------------------------------------------------------------
  ! lsqr_solve_sensit, src/inversion/lsqr_solver2.F90:47-308.  Solves min |[S; C] x - u| with the device-resident kernel S
  ! (matrix_sensit) and the constraint rows C (matrix_cons, built on the host with add / new_row like the reference).
  ! u(nlines) is consumed (overwritten), x(ncolumns) is fully overwritten, like the reference (:61-62, :120).
  ! Rows of C that form a diagonal over this rank's cells (damping%add, damping.F90:158-179: nelements_total rows per block, one
  ! entry per local cell at consecutive columns) are applied on the fly on the GPU; all other rows are uploaded as general rows.
  subroutine lsqr_solve_sensit(nlines, ncolumns, niter, rmin, gamma, target_misfit, matrix_sensit, matrix_cons, u, x, &
                               SOLVE_PROBLEM, nelements, nx, ny, nz, ncomponents, compression_type, WAVELET_DOMAIN, memory, &
                               myrank_, nbproc_)
    integer, intent(in) :: nlines, ncolumns, niter
    real(dp), intent(in) :: rmin, gamma, target_misfit
    logical, intent(in) :: SOLVE_PROBLEM(2)
    integer, intent(in) :: nelements, nx, ny, nz, ncomponents, compression_type
    logical, intent(in) :: WAVELET_DOMAIN
    integer, intent(in) :: myrank_, nbproc_
    type(t_sparse_matrix), intent(in) :: matrix_sensit
    type(t_sparse_matrix), intent(in) :: matrix_cons
    real(dp), intent(inout) :: x(ncolumns)
    real(dp), intent(inout), target :: u(nlines)
    real(dp), intent(out) :: memory
    integer, parameter :: MAXBLK = 16
    type(c_ptr) :: ctx, dptr(MAXBLK), rptr(MAXBLK)
    real(c_float), allocatable, target :: diag(:, :)
    real(dp), allocatable, target :: rhs(:, :), g_rhs(:)
    integer(c_int64_t), allocatable, target :: g_rowptr(:)
    integer(c_int32_t), allocatable, target :: g_cols(:)
    real(c_float), allocatable, target :: g_vals(:)
    logical, allocatable :: is_diag(:)
    integer :: nl_s, nl_c, ntot, nblk_rows, nb, b, i, r, r0, cb, ce, nloc, c0, col, nblocks, g_nrows, ncomp_total
    integer(c_int64_t) :: e, g_nnz, pos
    integer(c_int) :: iters
    real(c_double) :: rr
    logical :: ok

    if (myrank_ == 0) print *, 'Entered subroutine lsqr_solve_sensit, gamma =', gamma
    if (matrix_sensit%get_total_row_number() + matrix_cons%get_total_row_number() /= nlines .or. &
        matrix_sensit%get_ncolumns() /= ncolumns .or. (matrix_cons%nl > 0 .and. matrix_cons%get_ncolumns() /= ncolumns)) &
      call exit_MPI('Wrong matrix sizes in lsqr_solve_sensit! Exiting.', myrank_, 0)                    ! :77-82
    ctx = tfx_api_context(myrank_, nbproc_)
    memory = 0.d0
    nl_s = matrix_sensit%get_total_row_number()
    nl_c = matrix_cons%get_total_row_number()
    ntot = nx * ny * nz
    cb = merge(part_cb, 0, partitioned .and. nbproc_ > 1)
    ce = merge(part_ce, ntot, partitioned .and. nbproc_ > 1)
    nloc = ce - cb
    if (nloc /= nelements) call exit_MPI('lsqr_solve_sensit: nelements does not match the partition!', myrank_, nelements)

    ! ---- classify the constraint rows, nelements_total at a time
    nb = nl_c / ntot
    allocate(is_diag(max(nb, 1)))
    is_diag = .false.
    nblocks = 0
    do b = 1, nb
      r0 = (b - 1) * ntot
      ok = .true.
      c0 = -1
      do i = 1, ntot
        r = r0 + i
        e = matrix_cons%ijl(r + 1) - matrix_cons%ijl(r)
        if (i > cb .and. i <= ce) then
          if (e > 1) then
            ok = .false.
          else if (e == 1) then
            col = matrix_cons%ija(matrix_cons%ijl(r) + 1) - (i - cb)           ! column of the block's first local cell, minus one
            if (c0 < 0) c0 = col
            if (col /= c0) ok = .false.
          endif
        else if (e /= 0) then
          ok = .false.
        endif
        if (.not. ok) exit
      enddo
      if (ok .and. c0 >= 0 .and. nblocks < MAXBLK) then
        is_diag(b) = .true.
        nblocks = nblocks + 1
      endif
    enddo
    allocate(diag(ncolumns, max(nblocks, 1)), rhs(ncolumns, max(nblocks, 1)))
    nblocks = 0
    g_nrows = 0
    g_nnz = 0
    do b = 1, nb
      r0 = (b - 1) * ntot
      if (is_diag(b)) then
        nblocks = nblocks + 1
        diag(:, nblocks) = 0.0
        rhs(:, nblocks) = 0.d0
        c0 = -1
        do i = cb + 1, ce
          r = r0 + i
          if (matrix_cons%ijl(r + 1) > matrix_cons%ijl(r)) then
            col = matrix_cons%ija(matrix_cons%ijl(r) + 1)
            c0 = col - (i - cb)
            diag(col, nblocks) = matrix_cons%sa(matrix_cons%ijl(r) + 1)
          endif
        enddo
        do i = cb + 1, ce                                   ! right-hand side of every local row, stored entry or not
          rhs(c0 + (i - cb), nblocks) = u(nl_s + r0 + i)
        enddo
        dptr(nblocks) = c_loc(diag(1, nblocks))
        rptr(nblocks) = c_loc(rhs(1, nblocks))
      else
        g_nrows = g_nrows + ntot
        g_nnz = g_nnz + matrix_cons%ijl(r0 + ntot + 1) - matrix_cons%ijl(r0 + 1)
      endif
    enddo
    nblk_rows = nb * ntot
    if (nl_c > nblk_rows) then                              ! a tail that is not a whole block
      g_nrows = g_nrows + (nl_c - nblk_rows)
      g_nnz = g_nnz + matrix_cons%ijl(nl_c + 1) - matrix_cons%ijl(nblk_rows + 1)
    endif
    if (g_nrows > 0) then
      allocate(g_rowptr(g_nrows + 1), g_cols(max(g_nnz, 1_c_int64_t)), g_vals(max(g_nnz, 1_c_int64_t)), g_rhs(g_nrows))
      g_rowptr(1) = 0
      pos = 0
      r = 0
      do b = 1, nb + 1
        if (b <= nb) then
          if (is_diag(b)) cycle
          r0 = (b - 1) * ntot
          i = ntot
        else
          r0 = nblk_rows
          i = nl_c - nblk_rows
        endif
        do col = 1, i
          e = matrix_cons%ijl(r0 + col + 1) - matrix_cons%ijl(r0 + col)
          if (e > 0) then
            g_cols(pos + 1:pos + e) = matrix_cons%ija(matrix_cons%ijl(r0 + col) + 1:matrix_cons%ijl(r0 + col + 1))
            g_vals(pos + 1:pos + e) = matrix_cons%sa(matrix_cons%ijl(r0 + col) + 1:matrix_cons%ijl(r0 + col + 1))
            ! the device layout wants ascending columns; the reference's builders add them in stencil order (damping_gradient.F90:190,
            ! cross_gradient.F90:322-369: both problems' columns interleaved)
            call sort_row(g_cols(pos + 1:pos + e), g_vals(pos + 1:pos + e), e)
            pos = pos + e
          endif
          r = r + 1
          g_rowptr(r + 1) = pos
          g_rhs(r) = u(nl_s + r0 + col)
        enddo
      enddo
      call api_check(tfx_cons_upload_csr(ctx, int(g_nrows, c_int64_t), g_rowptr, g_cols, g_vals, g_rhs), 'lsqr_solve_sensit (matrix_cons)', myrank_)
    endif

    ! ---- WAVELET_DOMAIN (joint_inverse_problem.F90:189-198): spatial unknowns -> every product with S goes through the transform
    if (.not. WAVELET_DOMAIN .and. compression_type > 0) then
      call api_check(tfx_lsqr_set_wavelet_domain(ctx, 0_c_int, nx, ny, nz, compression_type), 'WAVELET_DOMAIN', myrank_)
      if (nbproc_ > 1) then
        ncomp_total = ncomponents * count(SOLVE_PROBLEM)
        call api_check(tfx_lsqr_set_partition(ctx, int(cb, c_int64_t), int(ncomp_total, c_int)), 'tfx_lsqr_set_partition', myrank_)
      endif
    else
      call api_check(tfx_lsqr_set_wavelet_domain(ctx, 1_c_int, nx, ny, nz, compression_type), 'WAVELET_DOMAIN', myrank_)
    endif
    call api_check(tfx_select_problem(ctx, 0_c_int), 'tfx_select_problem', myrank_)
    call api_check(tfx_lsqr_solve(ctx, niter, rmin, gamma, target_misfit, u, nblocks, dptr, rptr, x, iters, rr), 'lsqr_solve_sensit', myrank_)
    if (g_nrows > 0) call api_check(tfx_cons_clear(ctx), 'tfx_cons_clear', myrank_)
    if (.not. WAVELET_DOMAIN .and. compression_type > 0) &
      call api_check(tfx_lsqr_set_wavelet_domain(ctx, 1_c_int, nx, ny, nz, compression_type), 'WAVELET_DOMAIN', myrank_)
    u = 0.d0                                                ! consumed, like the reference's in-place use of the right-hand side
    if (myrank_ == 0) print *, 'End of subroutine lsqr_solve_sensit, r =', rr, ' iter =', iters          ! :300-305
  end subroutine lsqr_solve_sensit

  ! Ascending columns inside one constraint row (rows are a handful of entries: insertion sort).  Two entries of one row in the same
  ! column - the reference's format allows them, its products simply add both - become one entry with the fp32 sum of the two.
  ! Rows as tfx_matrix_upload_csr takes them - columns strictly ascending inside a row - from rows as t_sparse_matrix%add builds them
  ! (sparse_matrix.f90:213-229): ANY column order, and the same column more than once (the stencils of cross_gradient.F90 and
  ! damping_gradient.F90 add to one column from several directions).  Per row: a stable insertion sort by column (rows of constraint
  ! matrices hold a handful of entries), entries of one column merged by adding their values in the order they were added (fp32, like
  ! the sort_row of the constraint path).  where(k) = the entry of the result that input entry k went into (for writing values back).
  subroutine api_canonical_csr(nl, rowptr, ija, sa, rp, cols, vals, where)
    integer, intent(in) :: nl
    integer(c_int64_t), intent(in) :: rowptr(nl + 1)
    integer(c_int32_t), intent(in) :: ija(*)
    real(c_float), intent(in) :: sa(*)
    integer(c_int64_t), allocatable, intent(out) :: rp(:)
    integer(c_int32_t), allocatable, intent(out) :: cols(:)
    real(c_float), allocatable, intent(out) :: vals(:)
    integer(c_int64_t), allocatable, intent(out), optional :: where(:)
    integer(c_int64_t), allocatable :: idx(:)
    integer(c_int64_t) :: a, b, i, j, m, n, k, nnz, t
    integer :: r
    nnz = rowptr(nl + 1)
    allocate(rp(nl + 1), cols(max(nnz, 1_c_int64_t)), vals(max(nnz, 1_c_int64_t)), idx(max(nnz, 1_c_int64_t)))
    if (present(where)) allocate(where(max(nnz, 1_c_int64_t)))
    m = 0
    rp(1) = 0
    do r = 1, nl
      a = rowptr(r) + 1; b = rowptr(r + 1); n = b - a + 1
      do i = 1, n
        idx(i) = a + i - 1
      enddo
      do i = 2, n                                    ! stable: equal columns keep the order they were added in
        t = idx(i)
        j = i - 1
        do while (j >= 1)
          if (ija(idx(j)) <= ija(t)) exit
          idx(j + 1) = idx(j)
          j = j - 1
        enddo
        idx(j + 1) = t
      enddo
      do i = 1, n
        k = idx(i)
        if (i > 1) then
          if (ija(k) == cols(m)) then
            vals(m) = vals(m) + sa(k)
            if (present(where)) where(k) = m
            cycle
          endif
        endif
        m = m + 1
        cols(m) = ija(k); vals(m) = sa(k)
        if (present(where)) where(k) = m
      enddo
      rp(r + 1) = m
    enddo
  end subroutine api_canonical_csr

  subroutine sort_row(c, v, n)
    integer(c_int64_t), intent(inout) :: n
    integer(c_int32_t), intent(inout) :: c(n)
    real(c_float), intent(inout) :: v(n)
    integer(c_int64_t) :: i, j, m
    integer(c_int32_t) :: ck
    real(c_float) :: vk
    do i = 2, n
      ck = c(i); vk = v(i)
      j = i - 1
      do while (j >= 1)
        if (c(j) <= ck) exit
        c(j + 1) = c(j); v(j + 1) = v(j)
        j = j - 1
      enddo
      c(j + 1) = ck; v(j + 1) = vk
    enddo
    m = 1
    do i = 2, n
      if (c(i) == c(m)) then
        v(m) = v(m) + v(i)
      else
        m = m + 1
        c(m) = c(i); v(m) = v(i)
      endif
    enddo
    n = m
  end subroutine sort_row

  !=======================================================================================================
  ! t_joint_inversion - src/inversion/joint_inverse_problem.F90
  !=======================================================================================================
  ! joint_inversion_initialize, :124-216: which terms the system has, and the WAVELET_DOMAIN rule (:189-198): the unknowns are wavelet
  ! coefficients unless a constraint acts in space (cross-gradient, clustering, gradient damping, Lp damping, local ADMM bounds,
  ! local damping weights).  The sensitivity matrix itself is allocated by read_sensitivity_kernel (it lives on the device).
  subroutine joint_inversion_initialize(this, par, nnz_sensit, myrank)
    class(t_joint_inversion), intent(inout) :: this
    type(t_parameters_inversion), intent(in) :: par
    integer(c_int64_t), intent(in) :: nnz_sensit
    integer, intent(in) :: myrank
    integer :: i
    this%nelements_total = par%nelements_total
    do i = 1, 2
      this%add_damping(i) = par%problem_weight(i) /= 0.d0 .and. par%alpha(i) /= 0.d0
      this%add_damping_gradient(i) = par%problem_weight(i) /= 0.d0 .and. par%beta(i) /= 0.d0
      this%add_admm(i) = par%problem_weight(i) /= 0.d0 .and. par%admm_type > 0
    enddo
    this%add_cross_grad = par%cross_grad_weight /= 0.d0
    this%add_clustering = any(par%clustering_weight_glob /= 0.d0 .and. par%problem_weight /= 0.d0)
    this%admm_cost = 0.d0
    this%WAVELET_DOMAIN = .true.
    if (this%add_cross_grad .or. this%add_clustering .or. this%add_damping_gradient(1) .or. this%add_damping_gradient(2) .or. &
        par%norm_power /= 2.d0 .or. par%admm_bound_type /= 1 .or. par%apply_local_damping_weight > 0) this%WAVELET_DOMAIN = .false.
    if (myrank == 0) print *, 'WAVELET_DOMAIN =', this%WAVELET_DOMAIN
    if (myrank == 0) print *, 'nnz of the sensitivity kernel =', nnz_sensit
  end subroutine joint_inversion_initialize

  ! joint_inversion_initialize2, :223-359: the remaining allocations, after the kernel has been read (the constraint matrix and the
  ! right-hand side are sized by the first solve, when the general rows of the caller are known)
  subroutine joint_inversion_initialize2(this, par, arr, model, myrank, nbproc)
    class(t_joint_inversion), intent(inout) :: this
    type(t_parameters_inversion), intent(in) :: par
    type(t_inversion_arrays), intent(in) :: arr(2)
    type(t_model), intent(in) :: model(2)
    integer, intent(in) :: myrank, nbproc
    integer :: i
    this%ndata_lines = 0
    do i = 1, 2
      if (par%problem_weight(i) /= 0.d0) this%ndata_lines = this%ndata_lines + par%ndata(i) * par%ndata_components(i)
    enddo
    if (allocated(this%z_admm)) deallocate(this%z_admm, this%u_admm, this%x0_ADMM)
    allocate(this%z_admm(par%nelements, 2), this%u_admm(par%nelements, 2), this%x0_ADMM(par%nelements, 2))
    this%z_admm = 0.d0; this%u_admm = 0.d0; this%x0_ADMM = 0.d0
    if (myrank == 0 .and. nbproc > 1) print *, 'joint inversion: data rows =', this%ndata_lines, ', cells of rank 0 =', model(1)%nelements + model(2)%nelements, &
                                               size(arr(1)%column_weight) + size(arr(2)%column_weight)
  end subroutine joint_inversion_initialize2

  subroutine joint_inversion_reset(this, myrank)                                   ! :366-380
    class(t_joint_inversion), intent(inout) :: this
    integer, intent(in) :: myrank
    if (allocated(this%b_RHS)) this%b_RHS = 0.d0
    if (allocated(this%matrix_cons%ijl)) call this%matrix_cons%reset()
    if (myrank < 0) return
  end subroutine joint_inversion_reset

  pure function joint_inversion_get_admm_cost(this) result(res)                    ! :585-590
    class(t_joint_inversion), intent(in) :: this
    real(dp) :: res(2)
    res = this%admm_cost
  end function joint_inversion_get_admm_cost

  ! Rows of the constraints that are built outside this module (gradient damping, cross-gradient, clustering), for the NEXT solve:
  ! CSR with 1-based local columns of this rank's unknowns, rows replicated on all ranks, and their right-hand side.
  subroutine joint_inversion_set_general_rows(this, nrows, rowptr, cols, vals, rhs)
    class(t_joint_inversion), intent(inout) :: this
    integer(c_int64_t), intent(in) :: nrows
    integer(c_int64_t), intent(in), optional :: rowptr(:)
    integer(c_int32_t), intent(in), optional :: cols(:)
    real(c_float), intent(in), optional :: vals(:)
    real(dp), intent(in), optional :: rhs(:)
    integer(c_int64_t) :: nnz
    this%g_nrows = nrows
    if (allocated(this%g_rowptr)) deallocate(this%g_rowptr, this%g_cols, this%g_vals, this%g_rhs)
    if (nrows <= 0) return
    nnz = rowptr(nrows + 1)
    allocate(this%g_rowptr(nrows + 1), this%g_cols(max(nnz, 1_c_int64_t)), this%g_vals(max(nnz, 1_c_int64_t)), this%g_rhs(nrows))
    this%g_rowptr = rowptr(1:nrows + 1)
    if (nnz > 0) then
      this%g_cols(1:nnz) = cols(1:nnz)
      this%g_vals(1:nnz) = vals(1:nnz)
    endif
    this%g_rhs = rhs(1:nrows)
  end subroutine joint_inversion_set_general_rows

  ! joint_inversion_calculate_matrix_partitioning, :712-739: rows of problem 2 follow problem 1's data rows, its columns follow
  ! problem 1's local unknowns
  subroutine joint_inversion_calculate_matrix_partitioning(par, line_start, line_end, param_shift)
    type(t_parameters_inversion), intent(in) :: par
    integer, intent(out) :: line_start(2), line_end(2), param_shift(2)
    line_start = 0; line_end = 0; param_shift = 0
    if (par%problem_weight(1) /= 0.d0) line_end(1) = par%ndata(1) * par%ndata_components(1)
    if (par%problem_weight(2) /= 0.d0) then
      line_start(2) = line_end(1)
      line_end(2) = line_start(2) + par%ndata(2) * par%ndata_components(2)
      if (par%problem_weight(1) /= 0.d0) param_shift(2) = par%nelements          ! (gravity has one model component)
    endif
  end subroutine joint_inversion_calculate_matrix_partitioning

  ! admm_method_iterate_admm_arrays, src/inversion/admm_method.F90:70-134: projection of x + u onto the union of the cell's intervals
  subroutine iterate_admm_arrays(nel, nlithos, min_bound, max_bound, xm, z, u, x0out)
    integer, intent(in) :: nel, nlithos
    real(dp), intent(in) :: min_bound(nlithos, nel), max_bound(nlithos, nel), xm(nel)
    real(dp), intent(inout) :: z(nel), u(nel)
    real(dp), intent(out) :: x0out(nel)
    integer :: p, j
    real(dp) :: a, mindist, v, closest
    logical :: inside
    do p = 1, nel
      a = xm(p) + u(p)
      inside = .false.
      do j = 1, nlithos
        if (min_bound(j, p) <= a .and. a <= max_bound(j, p)) then
          inside = .true.
          z(p) = a
          exit
        endif
      enddo
      if (.not. inside) then
        mindist = 1.d30
        closest = a
        do j = 1, nlithos
          v = dabs(min_bound(j, p) - a)
          if (v < mindist) then
            mindist = v
            closest = min_bound(j, p)
          endif
          v = dabs(max_bound(j, p) - a)
          if (v < mindist) then
            mindist = v
            closest = max_bound(j, p)
          endif
        enddo
        z(p) = closest
      endif
    enddo
    u = u + xm - z
    x0out = z - u
  end subroutine iterate_admm_arrays

  ! this rank's slice of a model-sized quantity, divided by the FULL column weight (zero guard of damping.F90:129-135), gathered from all
  ! ranks and - in the wavelet domain - transformed (damping.F90:135-150, wavelet_utils.F90:37-72)
  subroutine scaled_full_vector(ip, loc, nloc, full, par, wavelet, myrank, nbproc)
    integer, intent(in) :: ip, nloc, myrank, nbproc
    real(dp), intent(in) :: loc(nloc)
    real(dp), intent(out) :: full(:)
    type(t_parameters_inversion), intent(in) :: par
    logical, intent(in) :: wavelet
    integer :: p
    call get_full_array(loc, nloc, full, myrank, nbproc)
    do p = 1, size(full)
      if (kst(ip)%cw_full(p) /= 0.d0) then
        full(p) = full(p) / kst(ip)%cw_full(p)
      else
        full(p) = 0.d0
      endif
    enddo
    if (wavelet .and. par%compression_type > 0) call forward_wavelet(full, par%nx, par%ny, par%nz, par%compression_type)
  end subroutine scaled_full_vector

  !-------------------------------------------------------------------------------------------------------
  ! joint_inversion_solve, :393-573.  delta_model(nelements, nmodel_components, 2): the update of this rank's cells in model space.
  subroutine joint_inversion_solve(this, par, arr, model, delta_model, memory, myrank, nbproc)
    class(t_joint_inversion), intent(inout) :: this
    type(t_parameters_inversion), intent(in) :: par
    type(t_inversion_arrays), intent(in) :: arr(2)
    type(t_model), intent(inout) :: model(2)
    integer, intent(in) :: myrank, nbproc
    real(dp), intent(out) :: memory
    real(dp), intent(out) :: delta_model(par%nelements, par%nmodel_components, 2)
    logical :: SOLVE_PROBLEM(2)
    integer :: line_start(2), line_end(2), param_shift(2)
    integer :: i, k, p, ntot, nloc, cb, ce, nl_cons, lc, lc0, nc, ndt, kadm, ncols, row
    integer(c_int64_t) :: e8, nnz_cons
    real(dp), allocatable :: full(:), x(:), loc(:)
    real(dp) :: s1, s2, s3, sums(2)

    SOLVE_PROBLEM = par%problem_weight /= 0.d0
    ntot = par%nelements_total
    nloc = par%nelements
    cb = merge(part_cb, 0, partitioned .and. nbproc > 1)
    ce = cb + nloc
    call joint_inversion_calculate_matrix_partitioning(par, line_start, line_end, param_shift)
    ncols = this%matrix_sensit%get_ncolumns()
    ! ---- size of the system (joint_inverse_problem.F90:223-330): one diagonal block of nelements_total rows per damped model
    ! component and per ADMM term, then the caller's general rows
    nl_cons = int(this%g_nrows)
    do i = 1, 2
      if (.not. SOLVE_PROBLEM(i)) cycle
      if (par%alpha(i) /= 0.d0) nl_cons = nl_cons + ntot * model(i)%ncomponents
      if (par%admm_type > 0) nl_cons = nl_cons + ntot
    enddo
    ! (entries: at most one per row of the diagonal blocks + the caller's rows, whose count may change from one major iteration to
    ! the next - the cross-gradient's does - so the row builder grows when it has to)
    nnz_cons = int(nl_cons, c_int64_t)
    if (this%g_nrows > 0) nnz_cons = nnz_cons + this%g_rowptr(this%g_nrows + 1)
    if (.not. allocated(this%b_RHS)) allocate(this%b_RHS(this%ndata_lines + nl_cons))
    if (size(this%b_RHS) /= this%ndata_lines + nl_cons) &
      call exit_MPI('The number of constraint rows changed between major iterations!', myrank, nl_cons)
    if (.not. allocated(this%matrix_cons%ija)) then
      call this%matrix_cons%initialize(nl_cons, ncols, nnz_cons, myrank)
    else if (size(this%matrix_cons%ija, kind=c_int64_t) < nnz_cons) then
      call this%matrix_cons%initialize(nl_cons, ncols, nnz_cons + nnz_cons / 4, myrank)
    endif
    call this%matrix_cons%reset()
    this%b_RHS = 0.d0
    allocate(full(ntot))
    lc = this%ndata_lines                                            ! rows of the constraints start after the data rows
    do i = 1, 2
      if (.not. SOLVE_PROBLEM(i)) cycle
      nc = model(i)%ncomponents
      ndt = par%ndata(i) * par%ndata_components(i)
      lc0 = param_shift(i)                                           ! this rank's unknowns: [m1 cells (cb, ce]; m2 cells (cb, ce]]
      ! the right-hand side of the data rows: problem_weight * residuals (:379-387, :448-455)
      this%b_RHS(line_start(i) + 1:line_start(i) + ndt) = par%problem_weight(i) * reshape(arr(i)%residuals, (/ndt/))
      if (par%alpha(i) /= 0.d0) then                                 ! damping%add, damping.F90:97-234, one block per model component (:456-463)
        do k = 1, nc
          call scaled_full_vector(i, model(i)%val(:, k) - model(i)%val_prior(:, k), nloc, full, par, this%WAVELET_DOMAIN, myrank, nbproc)
          ! value = alpha * pw [* Lp multiplier] [* local weight] in double, ONE cast to the matrix precision in add (damping.F90:160-173,
          ! sparse_matrix.f90:226); right-hand side -alpha * pw * diff [* Lp multiplier] [* local weight] (:218-228)
          do p = 1, ntot
            if (p > cb .and. p <= ce) then
              s3 = full(p)
              s1 = par%alpha(i) * par%problem_weight(i)                  ! matrix value
              s2 = -par%alpha(i) * par%problem_weight(i) * s3            ! right-hand side
              if (par%norm_power /= 2.d0) then                           ! Lp norm multiplier (:250-262)
                if (s3 /= 0.d0) then
                  s3 = (abs(s3))**(par%norm_power / 2.d0 - 1.d0)
                else
                  s3 = 1.d0
                endif
                s1 = s1 * s3
                s2 = s2 * s3
              endif
              if (par%apply_local_damping_weight > 0) then               ! local weight = local alpha (:168-171, :225-228)
                s1 = s1 * model(i)%damping_weight(p - cb)
                s2 = s2 * model(i)%damping_weight(p - cb)
              endif
              call this%matrix_cons%add(s1, lc0 + (k - 1) * nloc + (p - cb), myrank)
              this%b_RHS(lc + p) = s2
            endif
            call this%matrix_cons%new_row(myrank)
          enddo
          lc = lc + ntot
        enddo
      endif
    enddo
    do i = 1, 2                                                      ! ***** ADMM method ***** (:490-527)
      if (.not. SOLVE_PROBLEM(i)) cycle
      if (par%admm_type <= 0) cycle
      nc = model(i)%ncomponents
      lc0 = param_shift(i)
      kadm = merge(1, 3, nc == 1)                                    ! vector model: bounds on Mz (:499-506)
      call iterate_admm_arrays(nloc, model(i)%nlithos, model(i)%min_bound, model(i)%max_bound, model(i)%val(:, kadm), &
                               this%z_admm(:, i), this%u_admm(:, i), this%x0_ADMM(:, i))
      call scaled_full_vector(i, model(i)%val(:, kadm) - this%x0_ADMM(:, i), nloc, full, par, this%WAVELET_DOMAIN, myrank, nbproc)
      do p = 1, ntot
        if (p > cb .and. p <= ce) then                               ! local weight = local rho (damping.F90:177-180, :264-267)
          call this%matrix_cons%add(par%rho_ADMM(i) * par%problem_weight(i) * model(i)%bound_weight(p - cb), &
                                    lc0 + (kadm - 1) * nloc + (p - cb), myrank)
          this%b_RHS(lc + p) = -par%rho_ADMM(i) * par%problem_weight(i) * full(p) * model(i)%bound_weight(p - cb)
        endif
        call this%matrix_cons%new_row(myrank)
      enddo
      lc = lc + ntot
      sums(1) = sum((this%z_admm(:, i) - model(i)%val(:, kadm))**2)       ! costs.f90:38-69, over the cells of all ranks
      sums(2) = sum(this%z_admm(:, i)**2)
      if (nbproc > 1) call allreduce_sum_dp(sums, 2)
      this%admm_cost(i) = 0.d0
      if (sums(2) /= 0.d0) this%admm_cost(i) = sqrt(sums(1) / sums(2))
      if (myrank == 0) print *, 'ADMM cost |x - z| / |z| =', this%admm_cost(i)
    enddo
    do row = 1, int(this%g_nrows)                                    ! the caller's rows (columns of this rank, rows replicated)
      do e8 = this%g_rowptr(row) + 1, this%g_rowptr(row + 1)
        call this%matrix_cons%add(real(this%g_vals(e8), dp), int(this%g_cols(e8)), myrank)
      enddo
      call this%matrix_cons%new_row(myrank)
      this%b_RHS(lc + row) = this%g_rhs(row)
    enddo
    call this%matrix_cons%finalize(myrank)
    ! ---- parallel sparse inversion (:546-552)
    allocate(x(ncols))
    x = 0.d0
    call lsqr_solve_sensit(size(this%b_RHS), ncols, par%niter, par%rmin, par%gamma, par%target_misfit, this%matrix_sensit, &
                           this%matrix_cons, this%b_RHS, x, SOLVE_PROBLEM, par%nelements, par%nx, par%ny, par%nz, &
                           par%nmodel_components, par%compression_type, this%WAVELET_DOMAIN, memory, myrank, nbproc)
    ! ---- back to model space (:556-571): inverse transform of the gathered slices when the unknowns were wavelet coefficients,
    ! then times the column weight (rescale_model, model.F90:312-324)
    delta_model = 0.d0
    allocate(loc(nloc))
    do i = 1, 2
      if (.not. SOLVE_PROBLEM(i)) cycle
      lc0 = param_shift(i)
      do k = 1, model(i)%ncomponents
        loc = x(lc0 + (k - 1) * nloc + 1:lc0 + k * nloc)
        if (par%compression_type > 0 .and. this%WAVELET_DOMAIN) then
          call get_full_array(loc, nloc, full, myrank, nbproc)
          call inverse_wavelet(full, par%nx, par%ny, par%nz, par%compression_type)
          loc = full(cb + 1:ce)
        endif
        delta_model(:, k, i) = loc * arr(i)%column_weight
      enddo
    enddo
    deallocate(full, x, loc)
  end subroutine joint_inversion_solve

end module tfx_reference_api
