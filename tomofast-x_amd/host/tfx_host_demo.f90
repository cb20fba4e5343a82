!=========================================================================================================
! tfx_host_demo - a Fortran host driving the MI355X path through the C ABI, the way the reference's
! solve_problem_joint_gravmag (src/problem_joint_gravmag.F90:65-613) drives its Fortran kernels:
! grid -> depth weight -> sensitivity kernel -> synthetic data -> damped LSQR -> inverse wavelet -> model update.
! Synthetic problem of SURVEY.md 8(d): 16 x 12 x 8 cells of 100 m, 6 x 5 observations, Haar r = 0.1, alpha = 1e-7.
! Prints fingerprints that tests/test_gpu_fortran_host.py compares with the Python host on the same inputs.
!=========================================================================================================
program tfx_host_demo
  use iso_c_binding
  use tfx_binding
  implicit none
  integer, parameter :: nx = 16, ny = 12, nz = 8, ox = 6, oy = 5
  integer, parameter :: N = nx * ny * nz, nd = ox * oy
  real(c_double), parameter :: h = 100.d0, alpha = 1.d-7
  real(c_double), allocatable, target :: X1(:), X2(:), Y1(:), Y2(:), Z1(:), Z2(:), cw(:), xd(:), yd(:), zd(:)
  real(c_double), allocatable, target :: mtrue(:), xw(:), d_obs(:), x(:), rhs(:), dm(:), d_calc(:)
  real(c_float), allocatable, target :: diag(:)
  type(c_ptr) :: ctx, dptr(1), rptr(1)
  integer(c_int64_t) :: nnz, nr, nc, nnzm, bytes
  real(c_double) :: err_sum, r
  integer(c_int) :: iters
  integer :: i, j, k, p, a, b

  allocate(X1(N), X2(N), Y1(N), Y2(N), Z1(N), Z2(N), cw(N), mtrue(N), xw(N), x(N), rhs(N), dm(N), diag(N))
  allocate(xd(nd), yd(nd), zd(nd), d_obs(nd), d_calc(nd))
  do k = 1, nz
    do j = 1, ny
      do i = 1, nx
        p = i + (j - 1) * nx + (k - 1) * nx * ny          ! src/inversion/grid.F90:409-426
        X1(p) = (i - 1) * h; X2(p) = i * h
        Y1(p) = (j - 1) * h; Y2(p) = j * h
        Z1(p) = (k - 1) * h; Z2(p) = k * h
        mtrue(p) = 0.d0
        if (k - 1 >= nz / 4 .and. k - 1 < nz / 2 .and. j - 1 >= ny / 3 .and. j - 1 < 2 * ny / 3 .and. &
            i - 1 >= nx / 3 .and. i - 1 < 2 * nx / 3) mtrue(p) = 300.d0
      enddo
    enddo
  enddo
  do b = 0, oy - 1
    do a = 0, ox - 1
      p = a + b * ox + 1
      xd(p) = (a + 0.5d0) * nx * h / ox + 0.37d0
      yd(p) = (b + 0.5d0) * ny * h / oy + 0.41d0
      zd(p) = -1.d0
    enddo
  enddo

  call tfx_check(tfx_create(0_c_int, c_null_ptr, ctx), 'tfx_create')
  call tfx_check(tfx_set_grid(ctx, nx, ny, nz, X1, X2, Y1, Y2, Z1, Z2), 'tfx_set_grid')
  call tfx_check(tfx_column_weight_type1(ctx, 2.d0, 0.d0, 4.d3, cw), 'tfx_column_weight_type1')
  call tfx_check(tfx_build_kernel_grav(ctx, int(nd, c_int64_t), xd, yd, zd, cw, 1_c_int, 0.1d0, 1.d0, c_null_ptr, &
                                       0_c_int64_t, int(N, c_int64_t), nnz, err_sum, c_null_ptr), 'tfx_build_kernel_grav')
  call tfx_check(tfx_matrix_info(ctx, nr, nc, nnzm, bytes), 'tfx_matrix_info')
  print '(a,i0,a,es23.16)', 'nnz_total = ', nnz, '  COMPRESSION ERROR, r = ', err_sum / nd

  ! synthetic data d = S Wav(m_true / cw)   (src/inversion/model.F90:220-307)
  xw = mtrue / cw
  call tfx_check(tfx_wavelet(ctx, xw, nx, ny, nz, 1_c_int64_t, 1_c_int, 1_c_int), 'tfx_wavelet')
  call tfx_check(tfx_calc_data(ctx, xw, 1.d0, c_null_ptr, d_obs), 'tfx_calc_data')

  ! one major iteration from a zero model: [S; alpha I] x = [d_obs; 0]
  diag = real(alpha, c_float)
  rhs = 0.d0
  dptr(1) = c_loc(diag)
  rptr(1) = c_loc(rhs)
  call tfx_check(tfx_lsqr_solve(ctx, 20_c_int, 1.d-13, 0.d0, 0.d0, d_obs, 1_c_int, dptr, rptr, x, iters, r), 'tfx_lsqr_solve')
  dm = x
  call tfx_check(tfx_wavelet(ctx, dm, nx, ny, nz, 1_c_int64_t, 1_c_int, 2_c_int), 'tfx_wavelet (inverse)')
  dm = dm * cw                                              ! src/inversion/joint_inverse_problem.F90:570
  xw = dm / cw
  call tfx_check(tfx_wavelet(ctx, xw, nx, ny, nz, 1_c_int64_t, 1_c_int, 1_c_int), 'tfx_wavelet')
  call tfx_check(tfx_calc_data(ctx, xw, 1.d0, c_null_ptr, d_calc), 'tfx_calc_data')
  print '(a,i0,a,es23.16)', 'lsqr iters = ', iters, '  r = ', r
  print '(a,es23.16,a,es23.16)', 'model min = ', minval(dm), '  max = ', maxval(dm)
  print '(a,es23.16)', 'data cost = ', norm2(d_calc - d_obs) / norm2(d_obs)
  call tfx_check(tfx_destroy(ctx), 'tfx_destroy')
  print '(a)', 'THE END.'
end program tfx_host_demo
