/*
 * tfx_oracle.h - CPU restatement of the Tomofast-x sensitivity-kernel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (tomofast-x_amd/, libtfx.so) never
 * links, loads or calls anything in oracle/.
 *
 * Every function restates one routine of the reference (cited file:line, paths relative to the
 * reference root).  Plain C, scalar, fp64 with strict IEEE semantics (built with -ffp-contract=off,
 * no -ffast-math) so that results match the reference's x86-64 build bit-for-bit where the
 * operation order is fixed (prism rows, wavelets, threshold, compaction, partition) - and, since round 6,
 * THROUGH THE SOLVER: norm2(u) is evaluated the way the reference's Fortran runtime (LLVM flang) does
 * (norm2_flang in tfx_oracle.c) and sum(v**2) sequentially, with which LSQR (all 12 runs of
 * tests/golden/lsqr.npz incl. exit iterations and residuals) and whole inversions (config 1 = 60 x 100
 * iterations + ADMM; Haar / D4 / uncompressed, magnetic, multi-component, joint, cross-gradient,
 * gradient damping, Lp, clustering: every end-to-end fixture) reproduce the reference's final models
 * bit for bit (tests/test_oracle_golden.py).
 *
 * Pinned against: oracle/_ref (the unmodified reference compiled by oracle/ref_build.sh) through the
 * golden vectors in tests/golden/ (made by tests/golden/make_golden.py), and against the reference's
 * own known-answer unit tests (src/tests/tests_lsqr.f90, tests_wavelet_compression.f90,
 * tests_sparse_matrix.f90) restated in tests/test_oracle_kat.py.
 */
#ifndef TFX_ORACLE_H
#define TFX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/forward/gravmag/grav/gravity_field.f90:131-195 (graviprism_z). Returns 0, or -1/-2 when
 * R+X<=0 / R+Y<=0 ("Data coordinate coincides with model grid boundary", :176-181). */
int orc_graviprism_z(int64_t n, const double *X1, const double *X2, const double *Y1, const double *Y2,
                     const double *Z1, const double *Z2, double xd, double yd, double zd, double *line);

/* src/forward/gravmag/grav/gravity_field.f90:41-126 (graviprism_full): lines[c*n + i], c = X, Y, Z.  Returns 0, or -1 / -2 / -3
 * when R+X<=0 / R+Y<=0 / R+Z<=0 (:96-104). */
int orc_graviprism_full(int64_t n, const double *X1, const double *X2, const double *Y1, const double *Y2,
                        const double *Z1, const double *Z2, double xd, double yd, double zd, double *lines);

/* src/forward/gravmag/mag/magnetic_field.f90: dircos (:91-110) -> magv[3]; magprism (:118-297) + sharmbox (:321-457)
 * for scalar susceptibility and TMI data (1 model component, 1 data component), incl. the in-cell 6-sub-box split.
 * Returns 0 or -1 / -2 (model grid X / Y boundary coincides with the data position, :345-354). */
void orc_dircos(double incl, double decl, double azim, double *magv);
int orc_magprism_tmi(int64_t n, const double *X1, const double *X2, const double *Y1, const double *Y2,
                     const double *Z1, const double *Z2, double xd, double yd, double zd, const double *magv,
                     double intensity, double *line);
/* General form: ncm = nmodel_components (1 susceptibility | 3 magnetisation vector), ncd = ndata_components (1 TMI | 3).
 * line[i + n*(k + ncm*d)] = sensit_line(i, k, d) (magnetic_field.f90:243-295). -3: wrong component counts. */
int orc_magprism(int64_t n, int ncm, int ncd, const double *X1, const double *X2, const double *Y1, const double *Y2,
                 const double *Z1, const double *Z2, double xd, double yd, double zd, const double *magv,
                 double intensity, double *line);

/* src/forward/gravmag/grav/gravity_field.f90:207-310 (gradiprism_full) / :315-362 (gradiprism_zz, only_zz != 0).
 * lines[c*n + i], c in the order the build stores the components (sensitivity_gravmag.F90:210-212):
 * XX, YY, ZZ, XY, YZ, ZX.  Returns 0, -4 (zero denominator) or -5 (bad log argument). */
int orc_gradiprism(int64_t n, int only_zz, const double *X1, const double *X2, const double *Y1, const double *Y2,
                   const double *Z1, const double *Z2, double xd, double yd, double zd, double *lines);

/* One (data, data-component, model-component) line of the build loop (sensitivity_gravmag.F90:222-311):
 * column weight -> cost_full -> wavelet -> threshold -> compaction.  line (N) is overwritten.  Returns nel. */
int64_t orc_compress_line(int64_t N, int nx, int ny, int nz, const double *cw, int compression_type, int64_t K,
                          double *line, int32_t *cols, float *vals, double *error_r);

int64_t orc_build_row_mag(int64_t N, int nx, int ny, int nz, const double *X1, const double *X2,
                          const double *Y1, const double *Y2, const double *Z1, const double *Z2,
                          const double *cw, double xd, double yd, double zd, const double *magv, double intensity,
                          int compression_type, int64_t K, double *work, int32_t *cols, float *vals, double *error_r,
                          int *ierr);

/* src/forward/gravmag/weights_gravmag.f90:71-79,170-195,204-250 (depth weighting type 1) followed by
 * src/problem_joint_gravmag.F90:178 (column_weight *= multiplier). z_axis: cell centre = (Z1+Z2)/2.
 * Returns 0 or -1 (non-positive depth) / -2 (zero weight). */
int orc_column_weight_type1(int64_t n, const double *X1, const double *X2, const double *Y1, const double *Y2,
                            const double *Z1, const double *Z2, double power, double Z0, double multiplier,
                            double *cw);

/* Distance weighting, type 2 (src/forward/gravmag/weights_gravmag.f90:81-138, then the same tail as type 1). */
int orc_column_weight_type2(int64_t n, const double *X1, const double *X2, const double *Y1, const double *Y2,
                            const double *Z1, const double *Z2, int64_t ndata, const double *xd, const double *yd,
                            const double *zd, double power, double beta, double multiplier, double *cw);

/* Minimum-distance weighting, type 3 (src/forward/gravmag/weights_gravmag.f90:140-162, then the same tail as type 1). */
int orc_column_weight_type3(int64_t n, const double *X1, const double *X2, const double *Y1, const double *Y2,
                            const double *Z1, const double *Z2, int64_t ndata, const double *xd, const double *yd,
                            const double *zd, double power, double multiplier, double *cw);

/* src/utils/wavelet_transform.F90:37-70. type 1 = Haar (:75-236), 2 = Daubechies D4 (:243-498).
 * s is (n1,n2,n3) with n1 fastest (Fortran order). Returns 0 or -1 (unknown type). */
int orc_forward_wavelet(double *s, int n1, int n2, int n3, int type);
int orc_inverse_wavelet(double *s, int n1, int n2, int n3, int type);

/* src/forward/gravmag/sensitivity_gravmag.F90:230-295 (+ src/utils/sort.f90:36-60 for the order
 * statistic).  row holds the wavelet coefficients (N); K = nel_compressed.  Outputs: cols (1-based,
 * ascending), vals (fp32), *thr, *cost_discarded.  Returns nel. */
int64_t orc_compress_row(const double *row, int64_t N, int64_t K, int32_t *cols, float *vals,
                         double *thr, double *cost_discarded);

/* One full sensitivity row: prism -> column weight -> cost_full -> wavelet -> compress
 * (sensitivity_gravmag.F90:189-311). work: N doubles. compression_type 0 => all N entries, thr=-1. */
int64_t orc_build_row_grav(int64_t N, int nx, int ny, int nz, const double *X1, const double *X2,
                           const double *Y1, const double *Y2, const double *Z1, const double *Z2,
                           const double *cw, double xd, double yd, double zd, int compression_type,
                           int64_t K, double *work, int32_t *cols, float *vals, double *error_r, int *ierr);

/* src/forward/gravmag/sensitivity_gravmag.F90:470-524 (nnz-balanced contiguous column partition). */
void orc_partition(const int32_t *nnz, int64_t N, int P, int32_t *nel_at_cpu, int64_t *nnz_at_cpu);

/* src/inversion/sparse_matrix.f90:322-327 / :397-403.  CSR with int64 row pointers (0-based offsets,
 * nrows+1), 1-based int32 columns, fp32 values; fp64 accumulation in the reference's order. */
void orc_spmv_add(int64_t nrows, const int64_t *rowptr, const int32_t *cols, const float *vals,
                  const double *x, double *b);
void orc_spmtv_add(int64_t nrows, const int64_t *rowptr, const int32_t *cols, const float *vals,
                   const double *x, double *b);

/* src/inversion/lsqr_solver2.F90:47-308 (lsqr_solve_sensit), single rank, WAVELET_DOMAIN = true or
 * compression off (no per-iteration transform).  A = [S; C], both CSR as above.  u (nl_s+nl_c) is
 * consumed, x (ncols) is overwritten.  target_misfit <= 0 disables the misfit exit.
 * Returns the number of iterations performed; *r_out = final relative residual. */
int orc_lsqr_solve_sensit(int64_t nl_s, int64_t nl_c, int64_t ncols, int niter, double rmin, double gamma,
                          double target_misfit,
                          const int64_t *s_rowptr, const int32_t *s_cols, const float *s_vals,
                          const int64_t *c_rowptr, const int32_t *c_cols, const float *c_vals,
                          double *u, double *x, double *r_out);

/* The same with WAVELET_DOMAIN = false when wavelet_type > 0: spatial unknowns, every product with S goes through the
 * n1 x n2 x n3 transform of each model component (lsqr_solver2.F90:137-145, :171-176, :200-206, :228-234). */
int orc_lsqr_solve_sensit_wd(int64_t nl_s, int64_t nl_c, int64_t ncols, int niter, double rmin, double gamma,
                             double target_misfit,
                             const int64_t *s_rowptr, const int32_t *s_cols, const float *s_vals,
                             const int64_t *c_rowptr, const int32_t *c_cols, const float *c_vals,
                             double *u, double *x, double *r_out, int wavelet_type, int n1, int n2, int n3);

/* src/inversion/lsqr_solver2.F90:478-494 */
void orc_normalize_columns(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, float *vals, double *column_norm);
void orc_soft_threshold(double *x, int64_t n, double gamma);

/* src/inversion/model.F90:220-307 (model_calculate_data), single rank, one model component.
 * data_calc[i] = (S (Wav(model/cw)))[i] / problem_weight / data_weight[i].  work: N doubles. */
int orc_calc_data(int64_t N, int nx, int ny, int nz, int64_t ndata, const double *model, const double *cw,
                  int compression_type, const int64_t *s_rowptr, const int32_t *s_cols, const float *s_vals,
                  double problem_weight, const double *data_weight, double *work, double *data_calc);

#ifdef __cplusplus
}
#endif
#endif
