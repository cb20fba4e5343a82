/*
 * tfx.h - C ABI of the MI355X-native Tomofast-x sensitivity-kernel hot path (libtfx.so).
 *
 * Drop-in boundary.  The reference (Tomofast-x, Fortran 2008 + MPI) has no FFI: its "plugin interface" for
 * this path is the set of Fortran procedures that src/problem_joint_gravmag.F90 and
 * src/inversion/joint_inverse_problem.F90 call.  Each entry point below replaces one of them; a Fortran host
 * binds them with iso_c_binding (interface blocks in INTEGRATION.md, module tomofast-x_amd/host/tfx_binding.f90).
 *
 * Conventions
 *  - Plain C: pointers + sizes, scalars by value, no C++/torch types.  All entry points return 0 on success,
 *    <0 on error (TFX_E_*); tfx_last_error() gives the message.  The reference's convention is
 *    print-banner + MPI_Abort (src/utils/mpi_tools.F90:29-53); the host turns a non-zero status into that.
 *  - One tfx_ctx per GPU (one process per GPU, or one host thread per ctx).  Not thread-safe per ctx.
 *  - Vector arguments may be HOST or DEVICE pointers (unified addressing, hipMemcpyDefault); they are
 *    borrowed for the duration of the call.  Device memory is owned by the ctx.
 *  - Column indices crossing the ABI are 1-based int32, as in the reference's SENSIT files and `ija`
 *    (src/inversion/sparse_matrix.f90:57).  Values are fp32 (MATRIX_PRECISION, src/global_typedefs.F90:42),
 *    vectors fp64 (CUSTOM_REAL, :31).  Cell order is i-fastest (src/inversion/grid.F90:409-426).
 *  - Multi-GPU: the ctx of rank r holds a contiguous column range of S (reference: nelements_at_cpu,
 *    src/forward/gravmag/sensitivity_gravmag.F90:470-524).  The two per-iteration reductions of LSQR
 *    (src/inversion/lsqr_solver2.F90:214, :514) are RCCL all-reduces on the ctx stream once the ctx has a communicator
 *    (tfx_comm_init_rccl); hosts without RCCL (test boxes) supply them through the hook (tfx_set_allreduce).
 */
#ifndef TFX_H
#define TFX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfx_ctx tfx_ctx;

enum {
    TFX_OK = 0,
    TFX_E_ARG = -1,        /* bad argument / size mismatch                                    */
    TFX_E_HIP = -2,        /* HIP runtime error (no GPU, out of memory, launch failure)       */
    TFX_E_GEOMETRY = -3,   /* "Data coordinate coincides with model grid boundary"            */
                           /*   (src/forward/gravmag/grav/gravity_field.f90:176-181)          */
    TFX_E_STATE = -4,      /* call order (no grid / no matrix / no solve in progress)         */
    TFX_E_NUMERIC = -5,    /* zero norm / zero weight where the reference aborts              */
    TFX_E_COMM = -6        /* the all-reduce hook reported failure                            */
};

/* ---- context ------------------------------------------------------------------------------------------- */
/* Creates a context on HIP device `device`.  `stream` = hipStream_t to launch on (NULL = null stream).      */
int tfx_create(int device, void *stream, tfx_ctx **ctx);
int tfx_destroy(tfx_ctx *ctx);
const char *tfx_last_error(void);
/* Library / device facts for logs: fills name[len], returns CU count (or <0). */
int tfx_device_info(tfx_ctx *ctx, char *name, int len, int64_t *hbm_bytes);

/* HIP devices visible to the process (a host that starts one process per GPU maps rank -> device with it).           */
int tfx_device_count(void);
/* Synchronous copy host <-> device (any direction), ordered after the work queued on the ctx stream: lets a host-language
 * all-reduce hook stage the device buffer through MPI when it has no device-aware collective at hand.                   */
int tfx_copy(tfx_ctx *ctx, void *dst, const void *src, int64_t bytes);

/* Raw device buffers for a host that moves packed matrix pieces itself (tfx_rowstore_pack -> send -> tfx_matrix_append_rows). */
int tfx_device_malloc(tfx_ctx *ctx, int64_t bytes, void **ptr_out);
int tfx_device_free(tfx_ctx *ctx, void *ptr);

/* All-reduce hook (sum, fp64, in place on a DEVICE buffer of n doubles, enqueued on `stream`).
 * NULL = single rank.  rank/nranks tell LSQR who adds the -alpha*u term (lsqr_solver2.F90:194-198).         */
typedef int (*tfx_allreduce_fn)(void *user, double *dev_buf, int64_t n, void *stream);
int tfx_set_allreduce(tfx_ctx *ctx, tfx_allreduce_fn fn, void *user, int rank, int nranks);
/* Optional companion hook with MPI_Allgatherv semantics on DEVICE buffers (same `user` as the all-reduce hook): rank r's nsend =
 * counts[r] doubles land at dev_recv + displs[r] on every rank.  Used for the column slices of multi-rank WAVELET_DOMAIN = F
 * (apply_wavelet_transform, src/inversion/wavelet_utils.F90:37-72); without it the slices travel as a zero-padded sum.       */
typedef int (*tfx_allgatherv_fn)(void *user, const double *dev_send, int64_t nsend, double *dev_recv, const int64_t *counts,
                                 const int64_t *displs, void *stream);
int tfx_set_allgatherv(tfx_ctx *ctx, tfx_allgatherv_fn fn);

/* ---- RCCL inside the library (the production multi-GPU path; one process per GPU) -----------------------------------
 * With a communicator the collectives of the path are ncclAllReduce / grouped ncclBroadcast / ncclSend / ncclRecv on the ctx
 * stream, queued between the kernels that produce and consume the buffers: no host callback, no stream synchronisation.
 *   lsqr_solver2.F90:214 and :511-515 (the two reductions of an LSQR iteration), model.F90:288-293 (predicted data),
 *   wavelet_utils.F90:37-72 (slices of the model vector), sensitivity_gravmag.F90:322 (nnz histogram), :795-830 (relayout).
 * Rank 0 calls tfx_comm_unique_id, the host language carries the 128 bytes to the other ranks (MPI_Bcast, torch store, file),
 * every rank calls tfx_comm_init_rccl (collective).  The hooks above remain for hosts / test boxes without RCCL.            */
#define TFX_COMM_ID_BYTES 128
int tfx_comm_unique_id(char *id_out /* [TFX_COMM_ID_BYTES] */);
int tfx_comm_init_rccl(tfx_ctx *ctx, const char *unique_id /* [TFX_COMM_ID_BYTES] */, int rank, int nranks);
int tfx_comm_destroy(tfx_ctx *ctx);
/* Start-up that can not strand a rank.  (i) tfx_comm_init_rccl itself gives up: the blocking rendezvous (ncclCommInitRank) runs
 * on a helper thread inside the library and the call returns TFX_E_COMM when it has not completed within the timeout (120 s;
 * environment TFX_COMM_INIT_TIMEOUT or tfx_debug_set "comm_init_timeout_s"; <= 0 waits for ever) - a peer that died BEFORE the
 * rendezvous costs the timeout, in every host language.  (ii) The hosts wrap the call in "try - agree over the control channel
 * (MPI / gloo) - fall back to the hooks"; a rank whose own call succeeded while another rank's failed drops its half-open
 * communicator with tfx_comm_abort (ncclCommAbort: no hand-shake with the peers).  tfx_comm_abort from another thread while a
 * rendezvous of the ctx is in flight cancels exactly that attempt (it never installs its communicator).                        */
int tfx_comm_abort(tfx_ctx *ctx);
/* Facts for the bench line / logs: ranks the communicator itself counts (ncclCommCount; 0 without a communicator), this rank's
 * index and device in it, ncclGetVersion, and the file the nccl* symbols come from (libtfx.so depends on librccl.so.1: a process
 * that has already mapped an RCCL of that soname - PyTorch bundles one - keeps that single copy, otherwise /opt/rocm/lib's).   */
int tfx_comm_info(tfx_ctx *ctx, int *nranks_seen, int *rank_seen, int *device_seen, int *version, char *path, int path_len);
/* Collectives for the host's own exchange steps, on DEVICE buffers, queued on the ctx stream.                              */
enum { TFX_F64 = 0, TFX_I32 = 1, TFX_I64 = 2 };
int tfx_comm_allreduce(tfx_ctx *ctx, void *dev_buf, int64_t n, int dtype);           /* sum, in place                       */
/* MPI_Allgatherv on DEVICE buffers (apply_wavelet_transform's exchange, src/inversion/wavelet_utils.F90:37-72): rank r's counts[r]
 * doubles land at dev_recv + displs[r] on every rank; counts / displs are host arrays of nranks entries; counts may differ and may be
 * zero.  RCCL: ONE group of ncclBroadcast calls, one per contributing rank; without a communicator the tfx_allgatherv_fn hook.  The call
 * LSQR makes itself for multi-rank WAVELET_DOMAIN = F, exposed for the host's own exchange steps and the N-GPU self-test of bench.py. */
int tfx_comm_allgatherv(tfx_ctx *ctx, const double *dev_send, double *dev_recv, const int64_t *counts, const int64_t *displs);
int tfx_comm_group_begin(tfx_ctx *ctx);                                               /* ncclGroupStart / End around a set   */
int tfx_comm_group_end(tfx_ctx *ctx);                                                 /*   of sends and receives             */
int tfx_comm_send(tfx_ctx *ctx, const void *dev_buf, int64_t bytes, int peer);
int tfx_comm_recv(tfx_ctx *ctx, void *dev_buf, int64_t bytes, int peer);
int tfx_comm_barrier(tfx_ctx *ctx);                                                   /* 1-element all-reduce + stream sync  */

/* ---- model grid ------------------------------------------------------------------------------------------
 * Replaces t_grid (src/inversion/grid.F90:30-50): six fp64 arrays of nx*ny*nz cell bounds, uploaded once.   */
int tfx_set_grid(tfx_ctx *ctx, int nx, int ny, int nz, const double *X1, const double *X2, const double *Y1,
                 const double *Y2, const double *Z1, const double *Z2);

/* calculate_depth_weight type 1 (src/forward/gravmag/weights_gravmag.f90:71-79,170-250) followed by
 * column_weight *= multiplier (src/problem_joint_gravmag.F90:178).  cw_out: N doubles.                      */
int tfx_column_weight_type1(tfx_ctx *ctx, double power, double Z0, double multiplier, double *cw_out);

/* calculate_depth_weight type 2, distance weighting (weights_gravmag.f90:81-138, the reference's default type), with the
 * same tail (sqrt(volume), max-normalise, invert, multiplier).                                                   */
int tfx_column_weight_type2(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double power,
                            double beta, double multiplier, double *cw_out);

/* calculate_depth_weight type 3, minimum-distance weighting (weights_gravmag.f90:140-162): w = sqrt(1 / (min over the data of the
 * distance from the cell centre + 0.01)^power), same tail.                                                        */
int tfx_column_weight_type3(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double power,
                            double multiplier, double *cw_out);

/* ---- unit-level kernels (also what the parity tests call) ------------------------------------------------ */
/* graviprism_z (src/forward/gravmag/grav/gravity_field.f90:131-195): ndata rows of N, rows_out[ndata*N].    */
int tfx_prism_rows_gz(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd,
                      double *rows_out);
/* magnetic_field_magprism + sharmbox + dircos (src/forward/gravmag/mag/magnetic_field.f90:118-297, :321-457, :91-110)
 * for the scalar-susceptibility model and TMI data (nModelComponents = nDataComponents = 1): ndata rows of N.
 * incl / decl / azim in degrees, intensity in nT (parfile keys forward.magneticField.*).  Observations strictly
 * inside a cell take the reference's 6-sub-box split (:139-226).                                             */
int tfx_prism_rows_mag(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double incl,
                       double decl, double azim, double intensity, double *rows_out);
/* Any row generator of the build loop (src/forward/gravmag/sensitivity_gravmag.F90:193-220):
 *   problem_type 1, data_type 1, 1 component     graviprism_z       gravity_field.f90:131-195
 *   problem_type 1, data_type 1, 3 components    graviprism_full    gravity_field.f90:41-126    (gx, gy, gz = LineX, LineY, LineZ;
 *                                                the gz rows have the bits of graviprism_z's.  Public in the reference but without a
 *                                                call site there - data_type 1 maps to graviprism_z only, sensitivity_gravmag.F90:195)
 *   problem_type 1, data_type 2, 1 component     gradiprism_zz      gravity_field.f90:315-362
 *   problem_type 1, data_type 2, 6 components    gradiprism_full    gravity_field.f90:207-310   (XX, YY, ZZ, XY, YZ, ZX)
 *   problem_type 2, nmodel_components 1|3 (susceptibility | magnetisation vector), ndata_components 1|3 (TMI | Bx, By, Bz)
 *                                                magprism           magnetic_field.f90:118-297, mag_field = incl, decl, azim, nT
 * rows_out[ndata][ndata_components][nmodel_components][N] = sensit_line(:, k, d) of every observation.            */
int tfx_prism_rows(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int nmodel_components, int64_t ndata,
                   const double *xd, const double *yd, const double *zd, const double *mag_field, double *rows_out);
/* forward_wavelet / inverse_wavelet (src/utils/wavelet_transform.F90:37-70), in place on nvec arrays of
 * n1*n2*n3 doubles stored back to back.  type 1 Haar, 2 D4; direction 1 forward, 2 inverse.                 */
int tfx_wavelet(tfx_ctx *ctx, double *s, int n1, int n2, int n3, int64_t nvec, int type, int direction);
/* Threshold + compaction of one row of wavelet coefficients (sensitivity_gravmag.F90:230-295).
 * cols_out/vals_out need room for K entries.  Returns nel in *nel_out.                                      */
int tfx_compress_row(tfx_ctx *ctx, const double *row, int64_t N, int64_t K, int32_t *cols_out, float *vals_out,
                     int64_t *nel_out, double *thr_out, double *cost_discarded_out);

/* ---- sensitivity matrix ----------------------------------------------------------------------------------
 * calculate_and_write_sensit + read_sensitivity_kernel (sensitivity_gravmag.F90:82-410, :648-883) without the
 * disk round trip: builds rows [0, ndata) for the observation points, keeps columns [col_begin, col_end)
 * (0-based, half-open; whole matrix = [0, N)) and stores S device-resident.  Values are scaled by
 * (float)(problem_weight * data_weight[i]) like :834-843 (data_weight NULL = 1).
 * compression_type 0 none / 1 Haar / 2 D4; K = int(rate*N) (:64-77).
 * Outputs: nnz kept in this ctx, sum over rows of the compression error r_i (:283) (divide by ndata for the
 * reference's "COMPRESSION ERROR"), and optionally the per-column nnz histogram over ALL N columns (:267),
 * which is what calculate_new_partitioning consumes.                                                        */
int tfx_build_kernel_grav(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd,
                          const double *column_weight, int compression_type, double rate, double problem_weight,
                          const double *data_weight, int64_t col_begin, int64_t col_end, int64_t *nnz_out,
                          double *error_sum_out, int32_t *nnz_hist_out);

/* The same for the magnetic problem (problem_type 2, sensitivity_gravmag.F90:215-219): rows from magprism.  */
int tfx_build_kernel_mag(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd,
                         const double *column_weight, double incl, double decl, double azim, double intensity,
                         int compression_type, double rate, double problem_weight, const double *data_weight,
                         int64_t col_begin, int64_t col_end, int64_t *nnz_out, double *error_sum_out,
                         int32_t *nnz_hist_out);

/* Storage bound of the NEXT tfx_build_kernel* into the selected slot: at most nnz_upper entries will be kept (0 = no bound given: rows x K,
 * K = int(rate * N), sensitivity_gravmag.F90:64-77).  For a build restricted to a column range [col_begin, col_end) whose entry count is
 * known from a counting pass (the sum of nnz_hist over the range - what a rank of the column-partitioned system holds): without it the
 * range is given the storage of the whole kernel.  A build that finds more entries than reserved fails ("capacity exceeded").  The
 * reservation is consumed by the next tfx_build_kernel* / tfx_rowstore_build* call whatever its outcome (also one that fails, builds into
 * a row store or keeps no columns) and is ignored by a build into the other slot (tfx_select_problem).                                  */
int tfx_matrix_reserve(tfx_ctx *ctx, int64_t nnz_upper);

/* General form: any data type / component counts the reference's build loop handles (:193-311).  The matrix has
 * ndata*ndata_components rows (row = idata*ndata_components + d, like read_sensitivity_kernel's new_row per (i, d), :855)
 * and nmodel_components*(col_end - col_begin) columns: model component k occupies columns
 * k*(col_end - col_begin) + (cell - col_begin) (:829-846).  Every (i, d, k) line is weighted, transformed and thresholded
 * on its own (:222-295).  data_weight[ndata*ndata_components], d fastest = data_weight(d, i); nnz_hist_out over the N cells
 * counts every line (:267).  mag_field = incl, decl, azim, intensity_nT (problem_type 2) or NULL.                    */
int tfx_build_kernel(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int nmodel_components, int64_t ndata,
                     const double *xd, const double *yd, const double *zd, const double *column_weight, const double *mag_field,
                     int compression_type, double rate, double problem_weight, const double *data_weight, int64_t col_begin,
                     int64_t col_end, int64_t *nnz_out, double *error_sum_out, int32_t *nnz_hist_out);

/* Joint inversion of two problems on one grid (gravity + magnetic, src/inversion/joint_inverse_problem.F90:712-739): slot 0 / 1
 * selects the sensitivity matrix that tfx_build_kernel*, tfx_matrix_*, tfx_rowstore_*, tfx_spmv / tfx_spmtv and tfx_calc_data act
 * on (default 0).  When slot 1 holds a matrix, tfx_lsqr_* solve with S = blockdiag(slot 0, slot 1): b_data = [rows of problem 1;
 * rows of problem 2], x = [model 1; model 2] (the reference's line_start / param_shift layout), diagonal blocks and the general
 * constraint matrix span both column blocks.  tfx_select_problem(ctx, 1) + tfx_matrix_free drops the second problem.        */
int tfx_select_problem(tfx_ctx *ctx, int slot);

/* Alternative to building: upload a CSR (what read_sensitivity_kernel assembles from SENSIT files;
 * t_sparse_matrix add_row/new_row/finalize, src/inversion/sparse_matrix.f90:213-293).  rowptr: nrows+1
 * 0-based offsets; cols 1-based local column indices in [1, ncols], ascending within a row.                 */
int tfx_matrix_upload_csr(tfx_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr,
                          const int32_t *cols, const float *vals);
/* Size query, then download as CSR (for SENSIT-format writers and tests).                                   */
int tfx_matrix_info(tfx_ctx *ctx, int64_t *nrows, int64_t *ncols, int64_t *nnz, int64_t *device_bytes);
int tfx_matrix_download_csr(tfx_ctx *ctx, int64_t *rowptr, int32_t *cols, float *vals);
int tfx_matrix_free(tfx_ctx *ctx);
/* Row scaling of the selected matrix: entry (r, c) *= (float)scale[r], in fp32 - what read_sensitivity_kernel applies when it
 * re-loads a kernel that was written unscaled (sensit_compressed * real(problem_weight * data_weight(d, i), MATRIX_PRECISION),
 * sensitivity_gravmag.F90:834-843).  A host that keeps calculate_and_write_sensit and read_sensitivity_kernel as two steps
 * builds with problem_weight = 1, data_weight = NULL and calls this from the second.  scale: nrows doubles.                */
int tfx_matrix_scale_rows(tfx_ctx *ctx, const double *scale);
/* t_sparse_matrix%normalize_columns (src/inversion/sparse_matrix.f90:414-443): column_norm_out[j] = sqrt(sum of the fp32 squares of
 * column j, added in fp64) and every entry of a column with a non-zero norm becomes (float)(value / column_norm[j]); zero columns stay.
 * The column sums are formed exactly (integer accumulation), i.e. with the same bits on every run; they differ from the reference's
 * sequential fp64 sums by at most nrows * 2^-53 relative.  Acts on the selected matrix (and its transposed copy).  ncols doubles.   */
int tfx_matrix_normalize_columns(tfx_ctx *ctx, double *column_norm_out);

/* General constraint rows: matrix_cons as the cross-gradient / clustering / gradient-damping / local-bound ADMM
 * builders assemble it on the host (src/inversion/joint_inverse_problem.F90:332, :466-544), uploaded as CSR with its
 * right-hand side (the b_RHS(lc:) slice).  Same layout and kernels as S; consumed by tfx_lsqr_* together with the
 * diagonal blocks.  Multi-rank: the rows are replicated, each rank uploads its column range (local 1-based columns),
 * like S.  tfx_cons_clear drops it.                                                                          */
int tfx_cons_upload_csr(tfx_ctx *ctx, int64_t nrows, const int64_t *rowptr, const int32_t *cols, const float *vals,
                        const double *rhs);
int tfx_cons_clear(tfx_ctx *ctx);

/* ---- multi-GPU build: row-parallel compression + relayout (what the reference does through SENSIT files and a
 * rank-0 MPI_Scatterv per row, sensitivity_gravmag.F90:179-189 and :795-830).
 * (One row store per problem slot, tfx_select_problem: a joint run partitions on the counts of both kernels first.)
 * tfx_rowstore_build: like tfx_build_kernel_* for THIS rank's share of the observations, but the compressed rows stay
 *   row-major on the device with all their columns (problem_type 1 grav / 2 magn with mag_field = incl, decl, azim, nT).
 * tfx_rowstore_counts: counts_out[r*nparts + d] = entries of local row r with column in [bounds[d], bounds[d+1]).
 * tfx_rowstore_pack: segment [col_begin, col_end) of local rows [row_begin, +nrows) packed row after row into DEVICE
 *   buffers (columns re-based to col_begin) - the piece to send to the owner of that column range.
 * tfx_matrix_begin / _append_rows / _finish: the receiving side assembles its column range from such pieces
 *   (row_begin must be a multiple of 2048 = the row-block size; pieces arrive in row-block order).               */
int tfx_rowstore_build(tfx_ctx *ctx, int problem_type, int64_t ndata, const double *xd, const double *yd, const double *zd,
                       const double *column_weight, const double *mag_field, int compression_type, double rate,
                       double problem_weight, const double *data_weight, int64_t *nnz_out, double *error_sum_out,
                       int32_t *nnz_hist_out);
/* ... with a data type / several data components (one model component); local row = idata_local*ndata_components + d. */
int tfx_rowstore_build_ex(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int64_t ndata, const double *xd,
                          const double *yd, const double *zd, const double *column_weight, const double *mag_field,
                          int compression_type, double rate, double problem_weight, const double *data_weight, int64_t *nnz_out,
                          double *error_sum_out, int32_t *nnz_hist_out);
/* ... and several model components (magnetisation vector): a stored row holds component k at columns k*N + cell; counts / pack
 * take CELL ranges and cut every component, the packed piece has component k at k*(col_end - col_begin) + cell - col_begin.  */
int tfx_rowstore_build_comp(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int nmodel_components, int64_t ndata,
                            const double *xd, const double *yd, const double *zd, const double *column_weight,
                            const double *mag_field, int compression_type, double rate, double problem_weight,
                            const double *data_weight, int64_t *nnz_out, double *error_sum_out, int32_t *nnz_hist_out);
int tfx_rowstore_counts(tfx_ctx *ctx, int nparts, const int64_t *bounds, int32_t *counts_out);
int tfx_rowstore_pack(tfx_ctx *ctx, int64_t row_begin, int64_t nrows, int64_t col_begin, int64_t col_end,
                      int32_t *cols_dev_out, float *vals_dev_out, int64_t capacity, int64_t *n_out);
int tfx_rowstore_free(tfx_ctx *ctx);
/* How the selected matrix is stored: bytes of the entry streams per stored entry (what one product streams per entry), stored
 * entries (non-zeros + empty-row markers + pad entries), bytes of those streams, adjoint on a transposed copy (0 / 1).        */
int tfx_matrix_format(tfx_ctx *ctx, double *bytes_per_entry, int64_t *stored_entries, int64_t *stream_bytes, int *adjoint_copy);
int tfx_matrix_begin(tfx_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz_upper);
int tfx_matrix_append_rows(tfx_ctx *ctx, int64_t row_begin, int64_t nr, const int32_t *cols_dev, const float *vals_dev,
                           const int32_t *nel_host);
int tfx_matrix_finish(tfx_ctx *ctx);

/* get_load_balancing_nelements (sensitivity_gravmag.F90:470-524): host-side, exact integer rule.            */
int tfx_partition_columns(const int32_t *nnz_hist, int64_t N, int nparts, int32_t *nel_at_part,
                          int64_t *nnz_at_part);

/* t_sparse_matrix%mult_vector / add_mult_vector (sparse_matrix.f90:298-329): b (+)= S x.  x: ncols, b: nrows. */
int tfx_spmv(tfx_ctx *ctx, const double *x, double *b, int add);
/* trans_mult_vector / add_trans_mult_vector (sparse_matrix.f90:373-405): b (+)= S^T x.  x: nrows, b: ncols.
 * With a transposed copy (the default when it fits) the sums are fp64 like the forward product's.  Without one the products of a
 * tile group are rounded to a fixed-point grid of 2^-60..2^-61 x (largest column sum of |value| in the group's tiles x max |x| of the
 * group's rows) and added exactly in 64-bit integers: the error bound is NORM-WISE per tile group (a column whose sum is 2^-k of the
 * group's largest keeps 2^-(60-k) relative accuracy), not component-wise like an fp64 sum; a non-finite x poisons every column of the
 * tile group.  tfx_matrix_format's adjoint_copy output says which of the two a matrix uses.                                      */
int tfx_spmtv(tfx_ctx *ctx, const double *x, double *b, int add);

/* ---- LSQR ------------------------------------------------------------------------------------------------
 * lsqr_solve_sensit (src/inversion/lsqr_solver2.F90:47-308) over [S; C] where C is a stack of `nblocks`
 * diagonal blocks diag(diag[b]) (the damping / ADMM blocks that damping%add builds, src/inversion/damping.F90:
 * 97-201; values are fp32 like matrix%add stores them, sparse_matrix.f90:226), never materialised as CSR.
 *   b_data: nrows (replicated on all ranks), rhs_blocks[b]: ncols (local), x_out: ncols (local).
 * In the multi-rank case the constraint rows stay rank-local and only ||u_cons||^2 joins the all-reduce
 * (mathematically identical to the reference's all-reduce over all rows, DESIGN.md "Multi-GPU").
 * gamma != 0 enables soft thresholding (:478-494); target_misfit > 0 the misfit exit (:168-189).
 * Returns iterations done and the final relative residual r = phibar / |b|.                                 */
int tfx_lsqr_solve(tfx_ctx *ctx, int niter, double rmin, double gamma, double target_misfit,
                   const double *b_data, int nblocks, const float *const *diag, const double *const *rhs_blocks,
                   double *x_out, int *iters_out, double *r_out);

/* WAVELET_DOMAIN switch of the reference (src/inversion/joint_inverse_problem.F90:189-198).  1 (default): the unknowns
 * live in the wavelet domain and S is applied as stored.  0: spatial unknowns (needed by constraints that act in space:
 * cross-gradient, clustering, gradient damping, local bounds) - every product with S goes through the 3-D transform
 * (lsqr_solver2.F90:200-206, :228-234).  Single rank: ncolumns = ncomponents * n1*n2*n3; multi-rank: see below.    */
int tfx_lsqr_set_wavelet_domain(tfx_ctx *ctx, int wavelet_domain, int n1, int n2, int n3, int wavelet_type);
/* Multi-rank WAVELET_DOMAIN = 0: this rank's unknowns are the cells [col_begin, col_begin + ncolumns/ncomponents) of each of
 * the ncomponents model components (all problems of a joint run counted).  Every product with S then gathers the slices of all
 * ranks through the all-reduce hook (disjoint supports), transforms the full vector on every rank and keeps its slice
 * (apply_wavelet_transform, src/inversion/wavelet_utils.F90:37-72: gather to rank 0, transform, scatter).                    */
int tfx_lsqr_set_partition(tfx_ctx *ctx, int64_t col_begin, int ncomponents);

/* The same solver in three steps, so that a caller (bench.py) can time exactly k iterations with everything
 * resident: begin = lines :120-157 (x=0, normalise u, v = A^T u, ...), iterate = k passes of the loop body
 * :163-290 (stops early on the reference's exit conditions; returns iterations actually done), end = copy x. */
int tfx_lsqr_begin(tfx_ctx *ctx, double rmin, double gamma, double target_misfit, const double *b_data,
                   int nblocks, const float *const *diag, const double *const *rhs_blocks);
int tfx_lsqr_iterate(tfx_ctx *ctx, int k, int *done_out, double *r_out);
int tfx_lsqr_end(tfx_ctx *ctx, double *x_out);

/* model_calculate_data (src/inversion/model.F90:220-307) after the model has been un-weighted and wavelet-
 * transformed by the caller or by tfx_model_to_wavelet: data_calc[i] = (S xw)[i] / problem_weight /
 * data_weight[i]; this is tfx_spmv + the two divisions, all-reduced through the hook.                       */
int tfx_calc_data(tfx_ctx *ctx, const double *xw_local, double problem_weight, const double *data_weight,
                  double *data_calc);

/* ---- timing (HIP events on the ctx stream; what bench.py brackets the timed region with) ----------------- */
int tfx_timer_start(tfx_ctx *ctx);
int tfx_timer_stop_ms(tfx_ctx *ctx, double *ms_out);       /* synchronises                                  */
/* Per-kernel accumulated GPU time (HIP events around every launch of the two matrix kernels) since the last
 * reset: which = 0 SpMV, 1 SpMtV, 2 the in-stream all-reduces (RCCL or hook).  Enabled with tfx_profile_enable(ctx, 1). */
int tfx_profile_enable(tfx_ctx *ctx, int on);
int tfx_profile_get(tfx_ctx *ctx, int which, double *total_ms, int64_t *launches);

/* Test / diagnostics switchboard.  key "force_general_prism" (value 0/1): always use the six-array prism kernel even
 * when the grid is a tensor product; key "tensor_grid": returns 1 when the tensor-product fast path is active;
 * key "band_select_min_cells" (value): grids of at least that many cells find the row thresholds by the sample-bracketed
 * band select instead of the full radix select (default 2^20; < 0: never) - both give the exact order statistic;
 * keys "band_batches" / "band_fallbacks": return how many row batches used it / fell back to the full select;
 * key "deterministic": accepted and ignored (older hosts set it) - the two matrix products are always reproducible: every fp64 sum
 * is formed in an order fixed by the matrix and its work lists (forward kernel), or exactly in integers (adjoint without a copy);
 * key "lsqr_merge_tail" (0 / 1, default 1; environment TFX_LSQR_MERGE_TAIL): 1 = the x / w update of an LSQR iteration also does the next
 *     iteration's u = -alpha u and constraint forward step (one launch instead of three); both forms give identical bits;
 * key "fwd_group" (0 = automatic, 1, 2; larger values are clamped to 2): row blocks that share one staged x tile in the forward product;
 * key "fwd_run" (1..16, default 2; environment TFX_FWD_RUN): consecutive chunks a wave of the forward kernel takes at a time;
 * key "adj_copy" (0 never / 1 always / 2 automatic, default 2; environment TFX_ADJ_COPY): matrices finished from now on get a
 * transposed copy of their tiles so that the adjoint product runs as the forward kernel on S^T (twice the matrix memory, a longer
 * build - DESIGN.md 3); automatic = matrices of at least "adj_copy_min_nnz" stored entries (default 0) whenever the device has
 * room; 1 = a copy that does not fit is an error; without a copy the adjoint runs on the tiles of S (exact integer accumulation);
 * "tr_panel_entries" / "tr_pos_budget" (0 = default) size the panels the copy is built in (tests force many small panels of either
 * shape), "drop_adj_copy" gives the copy of the selected matrix up;
 * "has_adj_copy" queries the selected matrix, "adj_copy_build_ms" returns the wall clock its copy took to build (ms);
 * key "build_overlap" (0/1/2, default 1): the kernel build runs its row generator (VALU-bound) on a second stream one batch ahead of
 * the wavelet / threshold / compaction kernels (HBM-bound) of the main stream and reads each batch's statistics one batch late
 * (three row buffers); 0 = one stream, one row buffer; 1 = overlapped when the kernel has at least 8 batches of rows; 2 = overlapped
 * always (what the tests use to drive the deferred-statistics path on small kernels) - the three give the same bits;
 * key "chain_under_wavelet" (0/1, default 1): overlapped build - the threshold / compaction chain of a batch runs on a third stream
 * beside the wavelet passes of the next batch (the generator leaves no registers for it); 0 = on the main stream; same bits;
 * key "chunk_exponent_span" (value): diagnostics - per mille of the stored 512-entry chunks whose non-zero values span at most
 * `value` binades (prints the histogram on stderr);
 * key "force_collectives" (0/1): issue the collectives of the multi-rank path even with one rank - with a world-size-1
 * communicator this runs the real ncclAllReduce / ncclBroadcast calls on a single-GPU box;
 * key "comm_init_timeout_s" (seconds, default 120; environment TFX_COMM_INIT_TIMEOUT; <= 0 waits for ever): how long tfx_comm_init_rccl
 *     waits for the rendezvous before it returns TFX_E_COMM;
 * key "items_per_cu" (default 16; bench.py: environment TFX_ITEMS_PER_CU): work items per CU the tile lists of matrices finished from
 *     now on are cut into (largest first); key "refinish": rebuilds the work lists of the selected matrix with the current
 *     "items_per_cu" / "fwd_group" (tools/ab_products.py sweeps them on one resident matrix);
 * key "gen_after_wavelet" (0..3, default 3; environment TFX_GEN_AFTER_WAVELET): overlapped build - the generator of the next batch is
 *     queued behind that many axis passes of the current batch's wavelet transform (0: at the batch start, beside all three);
 * key "gen_wgs_per_cu" (default 0 = one workgroup per tile; environment TFX_GEN_WGS_PER_CU): overlapped build - the gravity generator
 *     runs as a persistent grid of that many workgroups per CU that walk the tiles (leaves registers / LDS to the main stream's kernels);
 * key "wave_pipe" (0..8, default 0; environment TFX_WAVE_PIPE): the axis passes of the wavelet transform run as the software-pipelined
 *     persistent kernel (k_wavelet_axis_pipe: the next tile's loads in flight over the lifting of the current one) with that many
 *     workgroups per CU, where its shape conditions hold; returns the value set.  Same bits; measured in round 6 and not the default
 *     (profiles/README.md round 6: 3.6 TB/s at two workgroups per CU against 5.0 for one workgroup per tile).
 * Every key is read by name in csrc/api.hip; tests/test_cabi_exports.py checks that each one the library accepts, and each one any file
 * of this repository passes, is described here.                                                                              */
int tfx_debug_set(tfx_ctx *ctx, const char *key, int value);

/* Diagnostics: evaluates the device build of the prism kernels' fp64 log / atan2 (csrc/fastmath.h: table-reduced replacements of the
 * device libm calls behind gravity_field.f90:165-186 and magnetic_field.f90:376-399) on host arrays: out_log[i] = log(a[i]),
 * out_atan2[i] = atan2(a[i], b[i]).  Tests compare them with the host libm.                                                 */
int tfx_fastmath_eval(tfx_ctx *ctx, int64_t n, const double *a, const double *b, double *out_log, double *out_atan2);

#ifdef __cplusplus
}
#endif
#endif
