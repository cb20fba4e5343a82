!=========================================================================================================
! tfx_binding - iso_c_binding interface of libtfx.so (include/tfx.h) for a Fortran host.
!
! This is the binding a Tomofast-x maintainer adds to call the MI355X path from the existing Fortran code:
! each procedure below replaces the body of one reference routine (file:line in include/tfx.h and
! INTEGRATION.md).  Arrays are passed as assumed-size, contiguous, exactly as the reference holds them
! (real(8) vectors, real(4) matrix values, 1-based integer(4) column indices).
!=========================================================================================================
module tfx_binding
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: TFX_OK = 0, TFX_E_ARG = -1, TFX_E_HIP = -2, TFX_E_GEOMETRY = -3, &
                               TFX_E_STATE = -4, TFX_E_NUMERIC = -5, TFX_E_COMM = -6

  interface
    integer(c_int) function tfx_create(device, stream, ctx) bind(C, name="tfx_create")
      import :: c_int, c_ptr
      integer(c_int), value :: device
      type(c_ptr), value :: stream
      type(c_ptr), intent(out) :: ctx
    end function

    integer(c_int) function tfx_destroy(ctx) bind(C, name="tfx_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    type(c_ptr) function tfx_last_error() bind(C, name="tfx_last_error")
      import :: c_ptr
    end function

    integer(c_int) function tfx_device_count() bind(C, name="tfx_device_count")
      import :: c_int
    end function

    ! synchronous copy host <-> device, ordered after the ctx stream (staging for a host-side all-reduce hook)
    integer(c_int) function tfx_copy(ctx, dst, src, bytes) bind(C, name="tfx_copy")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_int64_t), value :: bytes
    end function

    integer(c_int) function tfx_set_allreduce(ctx, fn, user, rank, nranks) bind(C, name="tfx_set_allreduce")
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: ctx, user
      type(c_funptr), value :: fn
      integer(c_int), value :: rank, nranks
    end function

    ! t_grid (src/inversion/grid.F90:30-50)
    integer(c_int) function tfx_set_grid(ctx, nx, ny, nz, X1, X2, Y1, Y2, Z1, Z2) bind(C, name="tfx_set_grid")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, nz
      real(c_double), intent(in) :: X1(*), X2(*), Y1(*), Y2(*), Z1(*), Z2(*)
    end function

    ! calculate_depth_weight type 1 (src/forward/gravmag/weights_gravmag.f90:46-196)
    integer(c_int) function tfx_column_weight_type1(ctx, power, Z0, multiplier, cw) bind(C, name="tfx_column_weight_type1")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), value :: power, Z0, multiplier
      real(c_double), intent(out) :: cw(*)
    end function

    ! calculate_depth_weight type 2, distance weighting (src/forward/gravmag/weights_gravmag.f90:81-138)
    integer(c_int) function tfx_column_weight_type2(ctx, ndata, xd, yd, zd, power, beta, multiplier, cw) &
        bind(C, name="tfx_column_weight_type2")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: ndata
      real(c_double), intent(in) :: xd(*), yd(*), zd(*)
      real(c_double), value :: power, beta, multiplier
      real(c_double), intent(out) :: cw(*)
    end function

    ! calculate_depth_weight type 3, minimum-distance weighting (src/forward/gravmag/weights_gravmag.f90:140-162)
    integer(c_int) function tfx_column_weight_type3(ctx, ndata, xd, yd, zd, power, multiplier, cw) &
        bind(C, name="tfx_column_weight_type3")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: ndata
      real(c_double), intent(in) :: xd(*), yd(*), zd(*)
      real(c_double), value :: power, multiplier
      real(c_double), intent(out) :: cw(*)
    end function

    ! graviprism_z (src/forward/gravmag/grav/gravity_field.f90:131-195)
    integer(c_int) function tfx_prism_rows_gz(ctx, ndata, xd, yd, zd, rows) bind(C, name="tfx_prism_rows_gz")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: ndata
      real(c_double), intent(in) :: xd(*), yd(*), zd(*)
      real(c_double), intent(out) :: rows(*)
    end function

    ! forward_wavelet / inverse_wavelet (src/utils/wavelet_transform.F90:37-70)
    integer(c_int) function tfx_wavelet(ctx, s, n1, n2, n3, nvec, wtype, direction) bind(C, name="tfx_wavelet")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: s(*)
      integer(c_int), value :: n1, n2, n3, wtype, direction
      integer(c_int64_t), value :: nvec
    end function

    ! calculate_and_write_sensit + read_sensitivity_kernel (src/forward/gravmag/sensitivity_gravmag.F90:82-410, :648-883)
    integer(c_int) function tfx_build_kernel_grav(ctx, ndata, xd, yd, zd, column_weight, compression_type, rate, &
                                                  problem_weight, data_weight, col_begin, col_end, nnz, error_sum, nnz_hist) &
                                                  bind(C, name="tfx_build_kernel_grav")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: ndata, col_begin, col_end
      real(c_double), intent(in) :: xd(*), yd(*), zd(*), column_weight(*)
      integer(c_int), value :: compression_type
      real(c_double), value :: rate, problem_weight
      type(c_ptr), value :: data_weight      ! c_loc(array) or c_null_ptr
      integer(c_int64_t), intent(out) :: nnz
      real(c_double), intent(out) :: error_sum
      type(c_ptr), value :: nnz_hist         ! c_loc(int32 array of N) or c_null_ptr
    end function

    ! the same for the magnetic problem (magprism, src/forward/gravmag/mag/magnetic_field.f90:118-297)
    integer(c_int) function tfx_build_kernel_mag(ctx, ndata, xd, yd, zd, column_weight, incl, decl, azim, intensity, &
                                                 compression_type, rate, problem_weight, data_weight, col_begin, col_end, &
                                                 nnz, error_sum, nnz_hist) bind(C, name="tfx_build_kernel_mag")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: ndata, col_begin, col_end
      real(c_double), intent(in) :: xd(*), yd(*), zd(*), column_weight(*)
      real(c_double), value :: incl, decl, azim, intensity
      integer(c_int), value :: compression_type
      real(c_double), value :: rate, problem_weight
      type(c_ptr), value :: data_weight
      integer(c_int64_t), intent(out) :: nnz
      real(c_double), intent(out) :: error_sum
      type(c_ptr), value :: nnz_hist
    end function

    ! the general build: gradiometry (data_type 2, 1 | 6 data components), three-component magnetic data, magnetisation-vector
    ! model (3 model components) - the whole loop of src/forward/gravmag/sensitivity_gravmag.F90:189-311.
    ! mag_field = c_loc of (/ incl, decl, azim, intensity /) or c_null_ptr; data_weight(ndata_components, ndata)
    integer(c_int) function tfx_build_kernel(ctx, problem_type, data_type, ndata_components, nmodel_components, ndata, xd, yd, zd, &
                                             column_weight, mag_field, compression_type, rate, problem_weight, data_weight, &
                                             col_begin, col_end, nnz, error_sum, nnz_hist) bind(C, name="tfx_build_kernel")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int), value :: problem_type, data_type, ndata_components, nmodel_components
      integer(c_int64_t), value :: ndata, col_begin, col_end
      real(c_double), intent(in) :: xd(*), yd(*), zd(*), column_weight(*)
      type(c_ptr), value :: mag_field
      integer(c_int), value :: compression_type
      real(c_double), value :: rate, problem_weight
      type(c_ptr), value :: data_weight
      integer(c_int64_t), intent(out) :: nnz
      real(c_double), intent(out) :: error_sum
      type(c_ptr), value :: nnz_hist
    end function

    ! sensit_line(nelements, nmodel_components, ndata_components) of every observation, any generator of :193-220
    integer(c_int) function tfx_prism_rows(ctx, problem_type, data_type, ndata_components, nmodel_components, ndata, xd, yd, zd, &
                                           mag_field, rows) bind(C, name="tfx_prism_rows")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int), value :: problem_type, data_type, ndata_components, nmodel_components
      integer(c_int64_t), value :: ndata
      real(c_double), intent(in) :: xd(*), yd(*), zd(*)
      type(c_ptr), value :: mag_field
      real(c_double), intent(out) :: rows(*)
    end function

    ! multi-rank WAVELET_DOMAIN = F: where this rank's unknowns sit in the full model
    integer(c_int) function tfx_lsqr_set_partition(ctx, col_begin, ncomponents) bind(C, name="tfx_lsqr_set_partition")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: col_begin
      integer(c_int), value :: ncomponents
    end function

    ! ---- multi-GPU build: row-parallel compression + relayout (sensitivity_gravmag.F90:179-189, :795-830 without the files)
    integer(c_int) function tfx_rowstore_build_ex(ctx, problem_type, data_type, ndata_components, ndata, xd, yd, zd, column_weight, &
                                                  mag_field, compression_type, rate, problem_weight, data_weight, nnz, error_sum, &
                                                  nnz_hist) bind(C, name="tfx_rowstore_build_ex")
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int), value :: problem_type, data_type, ndata_components
      integer(c_int64_t), value :: ndata
      real(c_double), intent(in) :: xd(*), yd(*), zd(*), column_weight(*)
      type(c_ptr), value :: mag_field
      integer(c_int), value :: compression_type
      real(c_double), value :: rate, problem_weight
      type(c_ptr), value :: data_weight
      integer(c_int64_t), intent(out) :: nnz
      real(c_double), intent(out) :: error_sum
      type(c_ptr), value :: nnz_hist
    end function

    ! counts(d, r) = entries of local row r with column in [bounds(d), bounds(d+1))   (C layout counts[r*nparts + d])
    integer(c_int) function tfx_rowstore_build_comp(ctx, problem_type, data_type, ndata_components, nmodel_components, ndata, xd, yd, zd, &
                                                    column_weight, mag_field, compression_type, rate, problem_weight, data_weight, &
                                                    nnz_out, error_sum_out, nnz_hist_out) bind(C, name="tfx_rowstore_build_comp")
      import :: c_ptr, c_int, c_int64_t, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: problem_type, data_type, ndata_components, nmodel_components, compression_type
      integer(c_int64_t), value :: ndata
      real(c_double), intent(in) :: xd(*), yd(*), zd(*), column_weight(*)
      type(c_ptr), value :: mag_field, data_weight, nnz_hist_out
      real(c_double), value :: rate, problem_weight
      integer(c_int64_t), intent(out) :: nnz_out
      real(c_double), intent(out) :: error_sum_out
    end function tfx_rowstore_build_comp
    integer(c_int) function tfx_rowstore_counts(ctx, nparts, bounds, counts) bind(C, name="tfx_rowstore_counts")
      import :: c_int, c_ptr, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int), value :: nparts
      integer(c_int64_t), intent(in) :: bounds(*)
      integer(c_int32_t), intent(out) :: counts(*)
    end function

    ! columns [col_begin, col_end) of local rows [row_begin, row_begin + nrows) packed into DEVICE buffers
    integer(c_int) function tfx_rowstore_pack(ctx, row_begin, nrows, col_begin, col_end, cols_dev, vals_dev, capacity, n_out) &
        bind(C, name="tfx_rowstore_pack")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx, cols_dev, vals_dev
      integer(c_int64_t), value :: row_begin, nrows, col_begin, col_end, capacity
      integer(c_int64_t), intent(out) :: n_out
    end function

    integer(c_int) function tfx_rowstore_free(ctx) bind(C, name="tfx_rowstore_free")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function tfx_matrix_begin(ctx, nrows, ncols, nnz_upper) bind(C, name="tfx_matrix_begin")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: nrows, ncols, nnz_upper
    end function

    ! rows [row_begin, row_begin + nr) (row_begin a multiple of 2048), packed in DEVICE buffers, nel(r) entries each
    integer(c_int) function tfx_matrix_append_rows(ctx, row_begin, nr, cols_dev, vals_dev, nel) bind(C, name="tfx_matrix_append_rows")
      import :: c_int, c_ptr, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx, cols_dev, vals_dev
      integer(c_int64_t), value :: row_begin, nr
      integer(c_int32_t), intent(in) :: nel(*)
    end function

    integer(c_int) function tfx_matrix_finish(ctx) bind(C, name="tfx_matrix_finish")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function tfx_device_malloc(ctx, bytes, ptr) bind(C, name="tfx_device_malloc")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: bytes
      type(c_ptr), intent(out) :: ptr
    end function

    integer(c_int) function tfx_device_free(ctx, ptr) bind(C, name="tfx_device_free")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, ptr
    end function

    ! matrix_cons built on the host (cross-gradient, clustering, gradient damping, local bounds: joint_inverse_problem.F90:466-544)
    ! as CSR with its right-hand side; columns span all problems; consumed by the next tfx_lsqr_solve
    integer(c_int) function tfx_cons_upload_csr(ctx, nrows, rowptr, cols, vals, rhs) bind(C, name="tfx_cons_upload_csr")
      import :: c_int, c_ptr, c_float, c_double, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: nrows
      integer(c_int64_t), intent(in) :: rowptr(*)
      integer(c_int32_t), intent(in) :: cols(*)
      real(c_float), intent(in) :: vals(*)
      real(c_double), intent(in) :: rhs(*)
    end function

    integer(c_int) function tfx_cons_clear(ctx) bind(C, name="tfx_cons_clear")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! WAVELET_DOMAIN (joint_inverse_problem.F90:189-198): 0 = spatial unknowns, S applied through the n1 x n2 x n3 transform
    integer(c_int) function tfx_lsqr_set_wavelet_domain(ctx, wavelet_domain, n1, n2, n3, wavelet_type) &
        bind(C, name="tfx_lsqr_set_wavelet_domain")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int), value :: wavelet_domain, n1, n2, n3, wavelet_type
    end function

    ! joint inversion: slot 0 / 1 = which problem's sensitivity matrix the build / matrix / product / calc_data calls act on;
    ! LSQR solves with blockdiag(slot 0, slot 1) once slot 1 holds a matrix (src/inversion/joint_inverse_problem.F90:712-739)
    integer(c_int) function tfx_select_problem(ctx, slot) bind(C, name="tfx_select_problem")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int), value :: slot
    end function

    ! t_sparse_matrix add_row/new_row/finalize (src/inversion/sparse_matrix.f90:213-293)
    integer(c_int) function tfx_matrix_upload_csr(ctx, nrows, ncols, rowptr, cols, vals) bind(C, name="tfx_matrix_upload_csr")
      import :: c_int, c_ptr, c_float, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: nrows, ncols
      integer(c_int64_t), intent(in) :: rowptr(*)
      integer(c_int32_t), intent(in) :: cols(*)
      real(c_float), intent(in) :: vals(*)
    end function

    ! the matrix back as CSR (what the reference writes to its SENSIT files): rowptr(nrows+1) 0-based offsets, 1-based columns
    integer(c_int) function tfx_matrix_download_csr(ctx, rowptr, cols, vals) bind(C, name="tfx_matrix_download_csr")
      import :: c_int, c_ptr, c_float, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), intent(out) :: rowptr(*)
      integer(c_int32_t), intent(out) :: cols(*)
      real(c_float), intent(out) :: vals(*)
    end function

    integer(c_int) function tfx_matrix_info(ctx, nrows, ncols, nnz, device_bytes) bind(C, name="tfx_matrix_info")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), intent(out) :: nrows, ncols, nnz, device_bytes
    end function

    ! get_load_balancing_nelements (src/forward/gravmag/sensitivity_gravmag.F90:470-524)
    integer(c_int) function tfx_partition_columns(nnz_hist, N, nparts, nel_at, nnz_at) bind(C, name="tfx_partition_columns")
      import :: c_int, c_int64_t, c_int32_t
      integer(c_int32_t), intent(in) :: nnz_hist(*)
      integer(c_int64_t), value :: N
      integer(c_int), value :: nparts
      integer(c_int32_t), intent(out) :: nel_at(*)
      integer(c_int64_t), intent(out) :: nnz_at(*)
    end function

    ! mult_vector / add_mult_vector (src/inversion/sparse_matrix.f90:298-329)
    integer(c_int) function tfx_spmv(ctx, x, b, add) bind(C, name="tfx_spmv")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: x(*)
      real(c_double), intent(inout) :: b(*)
      integer(c_int), value :: add
    end function

    ! trans_mult_vector / add_trans_mult_vector (src/inversion/sparse_matrix.f90:373-405)
    integer(c_int) function tfx_spmtv(ctx, x, b, add) bind(C, name="tfx_spmtv")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: x(*)
      real(c_double), intent(inout) :: b(*)
      integer(c_int), value :: add
    end function

    ! lsqr_solve_sensit (src/inversion/lsqr_solver2.F90:47-308); diag / rhs_blocks: arrays of c_loc() pointers
    integer(c_int) function tfx_lsqr_solve(ctx, niter, rmin, gamma, target_misfit, b_data, nblocks, diag, rhs_blocks, &
                                           x, iters, r) bind(C, name="tfx_lsqr_solve")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: niter, nblocks
      real(c_double), value :: rmin, gamma, target_misfit
      real(c_double), intent(in) :: b_data(*)
      type(c_ptr), intent(in) :: diag(*), rhs_blocks(*)
      real(c_double), intent(out) :: x(*)
      integer(c_int), intent(out) :: iters
      real(c_double), intent(out) :: r
    end function

    ! model_calculate_data (src/inversion/model.F90:220-307), after un-weighting + wavelet
    integer(c_int) function tfx_calc_data(ctx, xw, problem_weight, data_weight, data_calc) bind(C, name="tfx_calc_data")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: xw(*)
      real(c_double), value :: problem_weight
      type(c_ptr), value :: data_weight
      real(c_double), intent(out) :: data_calc(*)
    end function

    ! read_sensitivity_kernel's row scaling (sensitivity_gravmag.F90:834-843): row r *= real(scale(r), 4)
    integer(c_int) function tfx_matrix_scale_rows(ctx, scale) bind(C, name="tfx_matrix_scale_rows")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: scale(*)
    end function

    ! entry bound of the next kernel build into the selected slot (a column range whose count is known from the histogram)
    integer(c_int) function tfx_matrix_reserve(ctx, nnz_upper) bind(C, name="tfx_matrix_reserve")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: nnz_upper
    end function

    ! t_sparse_matrix%normalize_columns (sparse_matrix.f90:414-443)
    integer(c_int) function tfx_matrix_normalize_columns(ctx, column_norm) bind(C, name="tfx_matrix_normalize_columns")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(out) :: column_norm(*)
    end function

    integer(c_int) function tfx_matrix_free(ctx) bind(C, name="tfx_matrix_free")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! ---- RCCL inside libtfx.so (include/tfx.h "RCCL inside the library")
    integer(c_int) function tfx_comm_abort(ctx) bind(C, name="tfx_comm_abort")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function tfx_comm_info(ctx, nranks_seen, rank_seen, device_seen, version, path, path_len) bind(C, name="tfx_comm_info")
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ctx
      integer(c_int), intent(out) :: nranks_seen, rank_seen, device_seen, version
      character(kind=c_char), intent(out) :: path(*)
      integer(c_int), value :: path_len
    end function

    integer(c_int) function tfx_comm_unique_id(id) bind(C, name="tfx_comm_unique_id")
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: id(128)
    end function

    integer(c_int) function tfx_comm_init_rccl(ctx, id, rank, nranks) bind(C, name="tfx_comm_init_rccl")
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ctx
      character(kind=c_char), intent(in) :: id(128)
      integer(c_int), value :: rank, nranks
    end function

    integer(c_int) function tfx_comm_destroy(ctx) bind(C, name="tfx_comm_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    ! dtype: 0 = real(8), 1 = integer(4), 2 = integer(8); dev_buf is a DEVICE pointer (tfx_device_malloc)
    integer(c_int) function tfx_comm_allreduce(ctx, dev_buf, n, dtype) bind(C, name="tfx_comm_allreduce")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx, dev_buf
      integer(c_int64_t), value :: n
      integer(c_int), value :: dtype
    end function

    ! MPI_Allgatherv on DEVICE buffers of real(8) (wavelet_utils.F90:37-72): counts / displs are host arrays of nranks entries
    integer(c_int) function tfx_comm_allgatherv(ctx, dev_send, dev_recv, counts, displs) bind(C, name="tfx_comm_allgatherv")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx, dev_send, dev_recv
      integer(c_int64_t), intent(in) :: counts(*), displs(*)
    end function

    integer(c_int) function tfx_comm_group_begin(ctx) bind(C, name="tfx_comm_group_begin")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function tfx_comm_group_end(ctx) bind(C, name="tfx_comm_group_end")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function

    integer(c_int) function tfx_comm_send(ctx, dev_buf, bytes, peer) bind(C, name="tfx_comm_send")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx, dev_buf
      integer(c_int64_t), value :: bytes
      integer(c_int), value :: peer
    end function

    integer(c_int) function tfx_comm_recv(ctx, dev_buf, bytes, peer) bind(C, name="tfx_comm_recv")
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: ctx, dev_buf
      integer(c_int64_t), value :: bytes
      integer(c_int), value :: peer
    end function

    integer(c_int) function tfx_comm_barrier(ctx) bind(C, name="tfx_comm_barrier")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
  end interface

contains

  ! The reference's error convention (src/utils/mpi_tools.F90:29-53): print the message and stop.
  subroutine tfx_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    integer :: i
    if (rc == 0) return
    call c_f_pointer(tfx_last_error(), msg, [1024])
    write(*, '(a)', advance='no') 'tfx error in '//where//': '
    do i = 1, 1024
      if (msg(i) == c_null_char) exit
      write(*, '(a)', advance='no') msg(i)
    enddo
    write(*, *)
    stop 1
  end subroutine tfx_check

end module tfx_binding
