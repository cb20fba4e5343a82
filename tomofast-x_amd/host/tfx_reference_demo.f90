!=========================================================================================================
! tfx_reference_demo - the same synthetic problem as tfx_host_demo (SURVEY.md 8d: 16 x 12 x 8 cells of 100 m, 6 x 5
! observations, Haar r = 0.1, alpha = 1e-7), driven ONLY through the reference's own entry points as module
! tfx_reference_api provides them - the call sequence of src/problem_joint_gravmag.F90 (:174, :197, :207, :241, :333) and of
! src/inversion/joint_inverse_problem.F90 (:456-463 damping rows, :549 lsqr_solve_sensit, :559-570 model update):
!   calculate_depth_weight -> calculate_and_write_sensit -> calculate_new_partitioning -> read_sensitivity_kernel ->
!   model%calculate_data -> matrix_cons add / new_row -> lsqr_solve_sensit -> inverse_wavelet -> model%calculate_data
! with a problem weight and data weights that differ from 1, so that read_sensitivity_kernel's row scaling is exercised.
! Prints the fingerprints tests/test_gpu_parity.py compares with the Python host on the same inputs.
!=========================================================================================================
program tfx_reference_demo
  use iso_c_binding
  use tfx_reference_api
  implicit none
  integer, parameter :: nx = 16, ny = 12, nz = 8, ox = 6, oy = 5
  integer, parameter :: N = nx * ny * nz, nd = ox * oy, myrank = 0, nbproc = 1
  real(CUSTOM_REAL), parameter :: h = 100.d0, alpha = 1.d-7, problem_weight = 2.5d0
  type(t_parameters_grav) :: gpar
  type(t_grid) :: grid_full
  type(t_data) :: data
  type(t_model) :: model
  type(t_sparse_matrix) :: matrix_sensit, matrix_cons
  real(CUSTOM_REAL), allocatable :: column_weight_full(:), column_weight(:), mtrue(:), b_RHS(:), delta_model(:), d_obs(:, :), d_calc(:, :)
  real(CUSTOM_REAL) :: memory
  integer :: nelements_at_cpu(nbproc)
  integer(c_int64_t) :: nnz
  logical :: SOLVE_PROBLEM(2)
  integer :: i, j, k, p, a, b
  ! the same major iteration once more through the joint-inversion object (problem_joint_gravmag.F90:236, :263, :497, :500)
  type(t_parameters_inversion) :: ipar
  type(t_inversion_arrays) :: iarr(2)
  type(t_joint_inversion) :: jinv
  type(t_model) :: model2(2)
  real(CUSTOM_REAL), allocatable :: delta3(:, :, :), direct(:)

  gpar%nx = nx; gpar%ny = ny; gpar%nz = nz
  gpar%nelements = N
  gpar%ndata = nd
  gpar%depth_weighting_type = 1
  gpar%depth_weighting_power = 2.d0
  gpar%compression_type = 1
  gpar%compression_rate = 0.1d0
  gpar%sensit_write = 0
  call grid_full%allocate(nx, ny, nz, 1, myrank)
  allocate(mtrue(N), column_weight_full(N), column_weight(N))
  do k = 1, nz
    do j = 1, ny
      do i = 1, nx
        p = i + (j - 1) * nx + (k - 1) * nx * ny          ! src/inversion/grid.F90:409-426
        grid_full%X1(p) = (i - 1) * h; grid_full%X2(p) = i * h
        grid_full%Y1(p) = (j - 1) * h; grid_full%Y2(p) = j * h
        grid_full%Z1(p) = (k - 1) * h; grid_full%Z2(p) = k * h
        mtrue(p) = 0.d0
        if (k - 1 >= nz / 4 .and. k - 1 < nz / 2 .and. j - 1 >= ny / 3 .and. j - 1 < 2 * ny / 3 .and. &
            i - 1 >= nx / 3 .and. i - 1 < 2 * nx / 3) mtrue(p) = 300.d0
      enddo
    enddo
  enddo
  call data%initialize(nd, 1, 1.d0, 1, myrank)
  do b = 0, oy - 1
    do a = 0, ox - 1
      p = a + b * ox + 1
      data%X(p) = (a + 0.5d0) * nx * h / ox + 0.37d0
      data%Y(p) = (b + 0.5d0) * ny * h / oy + 0.41d0
      data%Z(p) = -1.d0
      data%weight(1, p) = 1.d0 + 0.125d0 * mod(p, 4)       ! as if read from a data error file (data_gravmag.f90:243-279)
    enddo
  enddo

  ! (III) problem_joint_gravmag.F90:174-215
  call calculate_depth_weight(gpar, column_weight_full, grid_full, data, myrank, nbproc)
  column_weight_full = 4.d3 * column_weight_full                                           ! :178
  call calculate_and_write_sensit(gpar, grid_full, data, column_weight_full, memory, myrank, nbproc)
  call calculate_new_partitioning(gpar, nnz, nelements_at_cpu, 1, myrank, nbproc)
  gpar%nelements = nelements_at_cpu(myrank + 1)
  ! (IV) :241-248
  call read_sensitivity_kernel(gpar, matrix_sensit, column_weight, problem_weight, data%weight, 1, myrank, nbproc, nelements_at_cpu)
  call matrix_sensit%finalize(myrank)
  print '(a,i0)', 'nnz_total = ', nnz

  ! synthetic data from the true model (:318-345)
  call model%initialize(gpar%nelements, 1, N, myrank)
  model%grid_full%nx = nx; model%grid_full%ny = ny; model%grid_full%nz = nz
  allocate(d_obs(1, nd), d_calc(1, nd))
  model%val(:, 1) = mtrue
  call model%calculate_data(nd, 1, matrix_sensit, problem_weight, column_weight, data%weight, d_obs, gpar%compression_type, 0, 0, myrank, nbproc)

  ! one major iteration from a zero model (joint_inverse_problem.F90:393-573): right-hand side pw * weight * residual, damping rows
  allocate(b_RHS(nd + N), delta_model(N))
  b_RHS(1:nd) = problem_weight * (data%weight(1, :) * d_obs(1, :))
  call matrix_cons%initialize(N, N, int(N, c_int64_t), myrank)
  do i = 1, N                                                                             ! damping%add, damping.F90:158-179
    call matrix_cons%add(alpha * problem_weight, i, myrank)
    call matrix_cons%new_row(myrank)
    b_RHS(nd + i) = 0.d0
  enddo
  call matrix_cons%finalize(myrank)
  SOLVE_PROBLEM = (/.true., .false./)
  call lsqr_solve_sensit(nd + N, N, 20, 1.d-13, 0.d0, 0.d0, matrix_sensit, matrix_cons, b_RHS, delta_model, SOLVE_PROBLEM, &
                         gpar%nelements, nx, ny, nz, 1, gpar%compression_type, .true., memory, myrank, nbproc)
  call inverse_wavelet(delta_model, nx, ny, nz, gpar%compression_type)                     ! :559-567
  model%val(:, 1) = delta_model * column_weight                                           ! :570
  call model%calculate_data(nd, 1, matrix_sensit, problem_weight, column_weight, data%weight, d_calc, gpar%compression_type, 0, 0, myrank, nbproc)
  print '(a,es23.16,a,es23.16)', 'model min = ', minval(model%val), '  max = ', maxval(model%val)
  print '(a,es23.16)', 'data cost = ', norm2(d_calc - d_obs) / norm2(d_obs)
  print '(a,es23.16)', 'u consumed = ', maxval(abs(b_RHS))

  ! ---- the same system assembled and solved by jinv%solve: residuals and column weight in iarr, model and prior in the model object
  allocate(direct(N))
  direct = model%val(:, 1)
  ipar%nx = nx; ipar%ny = ny; ipar%nz = nz
  ipar%nelements = gpar%nelements; ipar%nelements_total = N
  ipar%ndata = (/nd, 0/); ipar%ndata_components = 1; ipar%nmodel_components = 1
  ipar%niter = 20; ipar%rmin = 1.d-13; ipar%gamma = 0.d0; ipar%target_misfit = 0.d0
  ipar%alpha = (/alpha, 0.d0/); ipar%norm_power = 2.d0
  ipar%problem_weight = (/problem_weight, 0.d0/)
  ipar%compression_type = gpar%compression_type
  call iarr(1)%allocate_aux(gpar%nelements, nd, 1, myrank)
  iarr(1)%column_weight = column_weight
  iarr(1)%residuals(1, :) = data%weight(1, :) * (d_obs(1, :) - 0.d0)                       ! :486-491, zero starting model
  call model2(1)%initialize(gpar%nelements, 1, N, myrank)
  call model2(2)%initialize(gpar%nelements, 1, N, myrank)
  call jinv%initialize(ipar, nnz, myrank)
  jinv%matrix_sensit = matrix_sensit                                                       ! (read_sensitivity_kernel filled this handle above)
  call jinv%initialize2(ipar, iarr, model2, myrank, nbproc)
  allocate(delta3(gpar%nelements, 1, 2))
  call jinv%solve(ipar, iarr, model2, delta3, memory, myrank, nbproc)
  call model2(1)%update(delta3(:, :, 1))
  print '(a,es23.16)', 'jinv vs direct = ', norm2(model2(1)%val(:, 1) - direct) / norm2(direct)
  call tfx_api_finalize()
  print '(a)', 'THE END.'
end program tfx_reference_demo
