// lsqr.hip - device-resident LSQR over [S; C]: lsqr_solve_sensit, src/inversion/lsqr_solver2.F90:47-308.
//
// S is the tiled sensitivity matrix (matrix.hip); C is a stack of diagonal blocks (the damping / ADMM blocks that
// damping%add builds row by row, src/inversion/damping.F90:158-179) applied on the fly.  All vectors and the
// Golub-Kahan scalars live on the device; per iteration the host reads back one 48-byte record to evaluate the
// reference's exit conditions (:163, :251-254, :286-289).
//
// Multi-rank (column-partitioned S, one rank per GPU): u_data is replicated, v/w/x and the constraint rows are local.
//   reduction 1 (lsqr_solver2.F90:214): all-reduce of [S_loc v ; ||u_cons,loc||^2]   (nrows + 1 doubles)
//   reduction 2 (:511-515):             all-reduce of ||v_loc||^2                     (1 double)
// The reference all-reduces the N constraint rows too; they are block-diagonal by construction (each rank fills only
// its own rows, damping.F90:158-179), so keeping them local and reducing only their squared norm is identical math.
#include "common.h"
#include <cmath>

namespace tfx {

struct Scalars {
    double alpha, beta, rhobar, phibar, b1, r, t1, t2;
    double sum_u, sum_uc, sum_v, misfit_ss;
    double inv_alpha;      // 1/alpha kept apart from t2 when the rotation runs in the same launch
    double rmin;           // stop test of the iteration loop, evaluated on the device (lsqr_solver2.F90:163)
    int32_t stop, skip, iters, pad2;   // stop: no further iteration may change x / the scalars; skip: this iteration is void
    int32_t rho_zero, u_zero, v_zero, pad;
};

struct LsqrState {
    int64_t nrows = 0, ncols = 0;   // nrows = data rows + rows of the general constraint matrix (if any)
    int64_t nrows_data = 0;
    int nblocks = 0;
    double rmin = 0, gamma = 0, target_misfit = 0;
    DBuf<double> u;        // nrows + 1 (last = local ||u_cons||^2, rides along in reduction 1)
    DBuf<double> v, w, x;
    DBuf<double> uc;       // nblocks * ncols
    DBuf<float> diag;      // nblocks * ncols
    DBuf<double> b0, sx;   // target-misfit only
    DBuf<double> tw;       // WAVELET_DOMAIN = F: wavelet-domain image of v / x, or S^T u before the inverse transform
    DBuf<double> twf;      //   multi-rank: the full-length vector (all ranks' slices) the transform runs on
    std::vector<int64_t> g_counts, g_displs;   //   cells per rank and first cell of every rank (all-gather of the slices)
    DBuf<double> red;      // block partial sums
    DBuf<unsigned int> cnt;   // arrival counters of the single-launch reductions (zero between launches)
    bool next_ready = false;  // the tail launch of the last iteration has already done the next iteration's u = -alpha u and constraint forward step
    DBuf<Scalars> sc;
    Scalars *h_sc = nullptr;   // pinned
    int iter = 0;          // iterations completed
    double r = 1.0;
    bool active = false, exact = false, finished = false;
};

__device__ __forceinline__ void set_beta(Scalars *sc, double uc_total);
__device__ __forceinline__ void set_alpha(Scalars *sc);
__device__ __forceinline__ void rotate(Scalars *sc);

constexpr int RED_BLOCKS = 1024;
constexpr int RED_THREADS = 256;

__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double sm[RED_THREADS / 64];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < RED_THREADS / 64; ++i) s += sm[i];
    return s;   // valid in thread 0
}

// Grid-wide reduction in one launch: every block leaves its partial sum in red[], the block that finishes last adds them up in
// index order - the same arithmetic as a separate k_final_sum launch, one launch less in a chain that is bound by
// launch-to-launch latency on small systems.  counter must be zero on entry; it is zero again on exit.
__device__ __forceinline__ bool last_block_done(unsigned int *counter)
{
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(counter, 1u);
        s_last = (t == gridDim.x - 1);
        if (s_last) *counter = 0;
    }
    __syncthreads();
    return s_last != 0;
}

__device__ __forceinline__ double final_sum_dev(const double *red, int n)       // valid in thread 0
{
    const volatile double *vr = red;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += vr[i];
    return block_sum(s);
}

// red[block] = sum over the block's grid-stride range of x[i]^2
__global__ void k_sumsq(const double *__restrict__ x, int64_t n, double *__restrict__ red)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s = fma(x[i], x[i], s);
    s = block_sum(s);
    if (threadIdx.x == 0) red[blockIdx.x] = s;
}

// ||u||^2 over the rows and beta in one launch (norm_u)
__global__ void k_sumsq_beta(const double *__restrict__ x, int64_t n, double *red, unsigned int *counter, Scalars *sc,
                             const double *uc_total)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s = fma(x[i], x[i], s);
    s = block_sum(s);
    if (threadIdx.x == 0) red[blockIdx.x] = s;
    if (last_block_done(counter)) {
        const double tot = final_sum_dev(red, (int)gridDim.x);
        if (threadIdx.x == 0) {
            sc->sum_u = tot;
            set_beta(sc, *uc_total);
        }
    }
}

// The same finalisations as separate one-block launches, for grids beyond ONE_LAUNCH_MAX_BLOCKS: there the arrival counter (one
// same-address atomic per block, ~50 ns each) costs more than the launch it saves.
constexpr int ONE_LAUNCH_MAX_BLOCKS = 64;
__global__ void k_final_sum_beta(const double *__restrict__ red, int n, Scalars *sc, const double *uc_total)
{
    const double tot = final_sum_dev(red, n);
    if (threadIdx.x == 0) {
        sc->sum_u = tot;
        set_beta(sc, *uc_total);
    }
}

__global__ void k_final_sum_v(const double *__restrict__ red, int n, Scalars *sc, int mode)
{
    const double tot = final_sum_dev(red, n);
    if (threadIdx.x == 0) {
        sc->sum_v = tot;
        if (mode == 2) { set_alpha(sc); rotate(sc); }
    }
}

// *dst = sum of red[0..n) in index order (deterministic)
__global__ void k_final_sum(const double *__restrict__ red, int n, double *dst)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += red[i];
    s = block_sum(s);
    if (threadIdx.x == 0) *dst = s;
}

__global__ void k_scale(double *__restrict__ x, int64_t n, const double *factor_ptr, int negate_or_zero)
{
    // negate_or_zero: 0 -> x *= f ; 1 -> x = -f * x ; 2 -> x = 0
    const double f = *factor_ptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (negate_or_zero == 0) x[i] = f * x[i];
        else if (negate_or_zero == 1) x[i] = -f * x[i];
        else x[i] = 0.0;
    }
}

// u_cons[b] = -alpha * u_cons[b] + diag[b] .* v  ; red[block] = partial ||u_cons||^2      (lsqr_solver2.F90:194-211)
// dst != null: the last block also writes the total (index-order sum of the partials) to *dst
__global__ void k_cons_forward(double *__restrict__ uc, const float *__restrict__ diag, const double *__restrict__ v,
                               int64_t ncols, int nblocks, const Scalars *sc, double *red, unsigned int *counter, double *dst)
{
    const double alpha = sc->alpha;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncols; i += (int64_t)gridDim.x * blockDim.x) {
        const double vi = v[i];
        for (int b = 0; b < nblocks; ++b) {
            const int64_t k = (int64_t)b * ncols + i;
            const double t = -alpha * uc[k] + (double)diag[k] * vi;
            uc[k] = t;
            s = fma(t, t, s);
        }
    }
    s = block_sum(s);
    if (threadIdx.x == 0) red[blockIdx.x] = s;
    if (dst && last_block_done(counter)) {
        const double tot = final_sum_dev(red, (int)gridDim.x);
        if (threadIdx.x == 0) *dst = tot;
    }
}

// v += sum_b diag[b] .* u_cons[b] ; red[block] = partial ||v||^2                          (lsqr_solver2.F90:236-241)
// mode 0: partials only; 1: the last block writes sum_v; 2: ... and alpha and the plane rotation (single rank)
__global__ void k_cons_adjoint(double *__restrict__ v, const float *__restrict__ diag, const double *__restrict__ uc,
                               int64_t ncols, int nblocks, double *red, unsigned int *counter, Scalars *sc, int mode)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncols; i += (int64_t)gridDim.x * blockDim.x) {
        double t = v[i];
        for (int b = 0; b < nblocks; ++b) {
            const int64_t k = (int64_t)b * ncols + i;
            t = fma((double)diag[k], uc[k], t);
        }
        v[i] = t;
        s = fma(t, t, s);
    }
    s = block_sum(s);
    if (threadIdx.x == 0) red[blockIdx.x] = s;
    if (mode != 0 && last_block_done(counter)) {
        const double tot = final_sum_dev(red, (int)gridDim.x);
        if (threadIdx.x == 0) {
            sc->sum_v = tot;
            if (mode == 2) { set_alpha(sc); rotate(sc); }
        }
    }
}

// beta = sqrt(sum_u + sum_uc); scale factor 1/beta (normalize, lsqr_solver2.F90:501-530)
__device__ __forceinline__ void set_beta(Scalars *sc, double uc_total)
{
    const double ss = sc->sum_u + uc_total;
    const double beta = sqrt(ss);
    sc->beta = beta;
    sc->u_zero = (beta == 0.0);
    sc->t1 = (beta != 0.0) ? 1.0 / beta : 1.0;     // t1 doubles as the scale factor until the rotation overwrites it
}

__device__ __forceinline__ void set_alpha(Scalars *sc)
{
    const double alpha = sqrt(sc->sum_v);
    sc->alpha = alpha;
    sc->v_zero = (alpha == 0.0);
    sc->t2 = (alpha != 0.0) ? 1.0 / alpha : 1.0;
    sc->inv_alpha = sc->t2;
}

__global__ void k_alpha(Scalars *sc) { set_alpha(sc); }

// first-iteration initialisation (lsqr_solver2.F90:134, :155-157)
__global__ void k_init_scalars(Scalars *sc, double rmin)
{
    sc->rmin = rmin;
    sc->b1 = sc->beta;
    sc->rhobar = sc->alpha;
    sc->phibar = sc->beta;
    sc->r = 1.0;
    sc->rho_zero = 0;
    sc->stop = 0;
    sc->skip = 0;
    sc->iters = 0;
}

// plane rotation (lsqr_solver2.F90:248-266, :277-280)
__device__ __forceinline__ void rotate(Scalars *sc)
{
    // The host queues several iterations before it looks at the scalars again; an iteration that starts after the loop's exit
    // condition was met (:163, :251-254, :286-289) must leave x, w and the scalars alone: it is marked void here, k_update_xw obeys.
    sc->skip = sc->stop;
    if (sc->stop) return;
    const double alpha = sc->alpha, beta = sc->beta;
    const double rho = sqrt(sc->rhobar * sc->rhobar + beta * beta);
    if (rho == 0.0) { sc->rho_zero = 1; sc->t1 = 0.0; sc->t2 = 0.0; sc->stop = 1; sc->skip = 1; return; }   // exits before the x / w update and the soft threshold (:251-254)
    const double rho_inv = 1.0 / rho;
    const double c = sc->rhobar * rho_inv;
    const double s = beta * rho_inv;
    const double theta = s * alpha;
    sc->rhobar = -c * alpha;
    const double phi = c * sc->phibar;
    sc->phibar = s * sc->phibar;
    sc->t1 = phi * rho_inv;
    sc->t2 = -theta * rho_inv;
    sc->r = sc->phibar / sc->b1;
    sc->iters += 1;
    if (fabs(sc->rhobar) < 1.e-30 || !(sc->r > sc->rmin)) sc->stop = 1;
}

__global__ void k_rotate(Scalars *sc) { rotate(sc); }
// alpha from the all-reduced |v|^2 and the plane rotation in one launch (multi-rank iterations: the normalisation of v is left to
// k_update_xw, like on a single rank)
__global__ void k_alpha_rotate(Scalars *sc) { set_alpha(sc); rotate(sc); }

// u *= t1 ; u_cons *= t1 ; v = -beta v   (normalisation of u and the first half of the adjoint step, :218-225) in one launch
__global__ void k_scale_u_uc_v(double *__restrict__ u, int64_t nu, double *__restrict__ uc, int64_t nuc, double *__restrict__ v,
                               int64_t nv, const Scalars *sc)
{
    const double f = sc->t1, beta = sc->beta;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < nu; i += st) u[i] = f * u[i];
    for (int64_t i = i0; i < nuc; i += st) uc[i] = f * uc[i];
    for (int64_t i = i0; i < nv; i += st) v[i] = -beta * v[i];
}

// v *= 1/alpha (normalize) ; x = t1*w + x ; w = t2*w + v ; soft threshold            (lsqr_solver2.F90:241, :269-274)
__global__ void k_update_xw(double *__restrict__ v, double *__restrict__ w, double *__restrict__ x, int64_t n,
                            const Scalars *sc, double inv_alpha_known, double gamma)
{
    // inv_alpha_known != 0: v has not been normalised yet (single-rank fused path): v = (1/alpha) v first, as k_scale would
    if (sc->skip) return;
    const double t1 = sc->t1, t2 = sc->t2, ia = sc->inv_alpha;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double wi = w[i];
        double xi = t1 * wi + x[i];
        double vi = v[i];
        if (inv_alpha_known != 0.0) { vi = ia * vi; v[i] = vi; }
        w[i] = t2 * wi + vi;
        if (gamma != 0.0) {                                    // :478-494
            if (fabs(xi) <= gamma) xi = 0.0;
            else if (xi <= -gamma) xi = xi + gamma;
            else if (xi >= gamma) xi = xi - gamma;
        }
        x[i] = xi;
    }
}

// k_update_xw, then - elementwise on the same index, with the v it has just normalised - the constraint forward step of the NEXT iteration
// (k_cons_forward: u_cons = -alpha u_cons + diag .* v, partial |u_cons|^2, lsqr_solver2.F90:194-211), then the next iteration's
// u = -alpha u (k_scale; u_mode 1: rank 0, 2: zero on the other ranks, :194-198).  One launch instead of three and one pass over v instead
// of two (2 % of an iteration of the reduced workloads - the launches of the chain pipeline, profiles/README.md round 5); the arithmetic
// and the partial sums are those of the separate kernels (same grid, same mapping), so the bits of a solve do not depend on which form ran.  A void iteration (Scalars::skip) leaves x, w and v alone as
// k_update_xw does; what it does to u and u_cons is what the separate kernels of the following (equally void) iteration would have done.
__global__ void k_update_xw_next(double *__restrict__ v, double *__restrict__ w, double *__restrict__ x, int64_t n, Scalars *sc, double gamma,
                                 double *__restrict__ uc, const float *__restrict__ diag, int nblocks, double *red, unsigned int *counter,
                                 double *dst, double *__restrict__ u, int64_t nr, int u_mode)
{
    const bool skip = sc->skip != 0;
    const double t1 = sc->t1, t2 = sc->t2, ia = sc->inv_alpha, alpha = sc->alpha;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double vi = v[i];
        if (!skip) {
            const double wi = w[i];
            double xi = t1 * wi + x[i];
            vi = ia * vi;
            v[i] = vi;
            w[i] = t2 * wi + vi;
            if (gamma != 0.0) {                                    // :478-494
                if (fabs(xi) <= gamma) xi = 0.0;
                else if (xi <= -gamma) xi = xi + gamma;
                else if (xi >= gamma) xi = xi - gamma;
            }
            x[i] = xi;
        }
        for (int b = 0; b < nblocks; ++b) {
            const int64_t k = (int64_t)b * n + i;
            const double t = -alpha * uc[k] + (double)diag[k] * vi;
            uc[k] = t;
            s = fma(t, t, s);
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += (int64_t)gridDim.x * blockDim.x)
        u[i] = (u_mode == 1) ? -alpha * u[i] : 0.0;
    s = block_sum(s);
    if (threadIdx.x == 0) red[blockIdx.x] = s;
    if (dst && last_block_done(counter)) {
        const double tot = final_sum_dev(red, (int)gridDim.x);
        if (threadIdx.x == 0) *dst = tot;
    }
}

// y += x
__global__ void k_axpy1(double *__restrict__ y, const double *__restrict__ x, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += x[i];
}

__global__ void k_copy(double *__restrict__ dst, const double *__restrict__ src, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// red[block] = partial sum (sx - b0)^2                                                 (lsqr_solver2.F90:183)
__global__ void k_misfit(const double *__restrict__ sx, const double *__restrict__ b0, int64_t n, double *__restrict__ red)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = sx[i] - b0[i];
        s = fma(d, d, s);
    }
    s = block_sum(s);
    if (threadIdx.x == 0) red[blockIdx.x] = s;
}

static inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(RED_BLOCKS, (n + RED_THREADS - 1) / RED_THREADS)); }

#define LAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(RED_THREADS), 0, s, __VA_ARGS__)

// S = blockdiag(slot 0, slot 1): the second problem of a joint inversion owns the rows after the first problem's data and the
// columns after its model (joint_inverse_problem.F90:712-739: line_start / param_shift)
static int S_forward(tfx_ctx *ctx, const double *x, double *b, int add)
{
    TFX_TRY(spmv_dev(ctx, ctx->mat, x, b, add));
    if (ctx->mat2.valid) TFX_TRY(spmv_dev(ctx, ctx->mat2, x + ctx->mat.ncols, b + ctx->mat.nrows, add));
    return 0;
}

static int S_adjoint(tfx_ctx *ctx, const double *x, double *b, int add)
{
    TFX_TRY(spmtv_dev(ctx, ctx->mat, x, b, add));
    if (ctx->mat2.valid) TFX_TRY(spmtv_dev(ctx, ctx->mat2, x + ctx->mat.nrows, b + ctx->mat.ncols, add));
    return 0;
}

// sum over the ranks: ncclAllReduce on the ctx stream when the ctx has a communicator, else the host's hook (comm.hip)
static int allreduce(tfx_ctx *ctx, double *buf, int64_t n)
{
    if (!ctx->multi()) return 0;
    return comm_allreduce_f64(ctx, buf, n);
}

static int read_scalars(tfx_ctx *ctx, LsqrState *L)
{
    TFX_HIP(hipMemcpyAsync(L->h_sc, L->sc.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    TFX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ||u||: sum over u_data (replicated) + all-reduced local ||u_cons||^2 which sits in u[nrows]
static int norm_u(tfx_ctx *ctx, LsqrState *L)
{
    hipStream_t s = ctx->stream;
    const int g = grid_for(L->nrows);
    if (g <= ONE_LAUNCH_MAX_BLOCKS) {
        LAUNCH(k_sumsq_beta, g, L->u.p, L->nrows, L->red.p, L->cnt.p, L->sc.p, L->u.p + L->nrows);
    } else {
        LAUNCH(k_sumsq, g, L->u.p, L->nrows, L->red.p);
        LAUNCH(k_final_sum_beta, 1, L->red.p, g, L->sc.p, L->u.p + L->nrows);
    }
    TFX_HIP(hipGetLastError());
    return 0;
}

static int scale_u(tfx_ctx *ctx, LsqrState *L)
{
    hipStream_t s = ctx->stream;
    LAUNCH(k_scale, grid_for(L->nrows), L->u.p, L->nrows, &L->sc.p->t1, 0);
    if (L->nblocks > 0) LAUNCH(k_scale, grid_for((int64_t)L->nblocks * L->ncols), L->uc.p, (int64_t)L->nblocks * L->ncols, &L->sc.p->t1, 0);
    TFX_HIP(hipGetLastError());
    return 0;
}

// full[k*N + c0 + i] = loc[k*nloc + i]  /  loc[k*nloc + i] = full[k*N + c0 + i]
__global__ void k_place_slice(double *__restrict__ full, const double *__restrict__ loc, int ncomp, int64_t N, int64_t c0, int64_t nloc)
{
    const int64_t n = (int64_t)ncomp * nloc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / nloc, j = i - k * nloc;
        full[k * N + c0 + j] = loc[i];
    }
}

__global__ void k_take_slice(double *__restrict__ loc, const double *__restrict__ full, int ncomp, int64_t N, int64_t c0, int64_t nloc)
{
    const int64_t n = (int64_t)ncomp * nloc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / nloc, j = i - k * nloc;
        loc[i] = full[k * N + c0 + j];
    }
}

// WAVELET_DOMAIN = F (lsqr_solver2.F90:200-206, :228-234 -> apply_wavelet_transform, wavelet_utils.F90:37-72).
// L->tw holds this rank's slice of a vector; it is replaced by the same slice of the transformed vector (dir 1 forward,
// 2 inverse).  Single rank: the slice is the whole vector.  Multi-rank: the slices are gathered into the full vector (disjoint
// supports, so the sum all-reduce of the hook is a gather), every rank transforms redundantly and keeps its slice (SURVEY 8e;
// the reference gathers to rank 0, transforms there and scatters).
static int transform_slice(tfx_ctx *ctx, LsqrState *L, int dir)
{
    hipStream_t s = ctx->stream;
    if (!ctx->multi())
        return wavelet_dev(ctx, L->tw.p, ctx->wd_n1, ctx->wd_n2, ctx->wd_n3, ctx->wd_nvec, ctx->wd_type, dir);
    const int64_t N = (int64_t)ctx->wd_n1 * ctx->wd_n2 * ctx->wd_n3;
    const int ncomp = ctx->wd_ncomp;
    const int64_t nloc = L->ncols / ncomp, full = (int64_t)ncomp * N;
    // all-gather of the slices, one model component after the other (N doubles each land on every rank: the reference moves the
    // same N doubles to rank 0 and back, wavelet_utils.F90:46-70)
    bool gathered = true;
    for (int k = 0; k < ncomp && gathered; ++k) {
        const int rc = comm_allgatherv_f64(ctx, L->tw.p + (int64_t)k * nloc, L->twf.p + (int64_t)k * N, L->g_counts.data(), L->g_displs.data());
        if (rc < 0) return rc;
        if (rc == 1) gathered = false;
    }
    if (!gathered) {
        // a host hook without all-gather: disjoint supports, so the sum all-reduce of the zero-padded full vector is a gather
        TFX_HIP(hipMemsetAsync(L->twf.p, 0, (size_t)full * sizeof(double), s));
        LAUNCH(k_place_slice, grid_for(L->ncols), L->twf.p, L->tw.p, ncomp, N, ctx->wd_col_begin, nloc);
        TFX_HIP(hipGetLastError());
        TFX_TRY(allreduce(ctx, L->twf.p, full));
    }
    TFX_TRY(wavelet_dev(ctx, L->twf.p, ctx->wd_n1, ctx->wd_n2, ctx->wd_n3, ncomp, ctx->wd_type, dir));
    LAUNCH(k_take_slice, grid_for(L->ncols), L->tw.p, L->twf.p, ncomp, N, ctx->wd_col_begin, nloc);
    TFX_HIP(hipGetLastError());
    return 0;
}

// v (+)= S^T u_data + C^T u_cons ; alpha = ||v|| ; v /= alpha
// fuse_rotate (iterations on a single rank): alpha and the plane rotation come out of the final-sum launch and the
// normalisation of v is left to k_update_xw
// defer_scale (iterations on several ranks): after the all-reduce of |v|^2 one launch sets alpha and rotates; v / alpha again in k_update_xw
static int adjoint_and_alpha(tfx_ctx *ctx, LsqrState *L, bool fuse_rotate = false, bool defer_scale = false)
{
    hipStream_t s = ctx->stream;
    if (ctx->spatial_unknowns) {                                         // lsqr_solver2.F90:137-145, :228-236
        TFX_TRY(S_adjoint(ctx, L->u.p, L->tw.p, 0));
        TFX_TRY(transform_slice(ctx, L, 2));
        LAUNCH(k_axpy1, grid_for(L->ncols), L->v.p, L->tw.p, L->ncols);
    } else {
        TFX_TRY(S_adjoint(ctx, L->u.p, L->v.p, 1));
    }
    if (ctx->cons.valid) TFX_TRY(spmtv_dev(ctx, ctx->cons, L->u.p + L->nrows_data, L->v.p, 1));     // lsqr_solver2.F90:147, :238
    const int g = grid_for(L->ncols);
    const int vmode = fuse_rotate ? 2 : 1;
    if (g <= ONE_LAUNCH_MAX_BLOCKS) {
        LAUNCH(k_cons_adjoint, g, L->v.p, L->diag.p, L->uc.p, L->ncols, L->nblocks, L->red.p, L->cnt.p + 1, L->sc.p, vmode);
    } else {
        LAUNCH(k_cons_adjoint, g, L->v.p, L->diag.p, L->uc.p, L->ncols, L->nblocks, L->red.p, L->cnt.p + 1, L->sc.p, 0);
        LAUNCH(k_final_sum_v, 1, L->red.p, g, L->sc.p, vmode);
    }
    TFX_HIP(hipGetLastError());
    if (fuse_rotate) return 0;
    TFX_TRY(allreduce(ctx, &L->sc.p->sum_v, 1));
    if (defer_scale) {
        LAUNCH(k_alpha_rotate, 1, L->sc.p);
    } else {
        LAUNCH(k_alpha, 1, L->sc.p);
        LAUNCH(k_scale, g, L->v.p, L->ncols, &L->sc.p->t2, 0);
    }
    TFX_HIP(hipGetLastError());
    return 0;
}

extern "C" int lsqr_free(tfx_ctx *ctx)
{
    if (ctx && ctx->lsqr) {
        if (ctx->lsqr->h_sc) (void)hipHostFree(ctx->lsqr->h_sc);
        delete ctx->lsqr;
        ctx->lsqr = nullptr;
    }
    return 0;
}

}  // namespace tfx

using namespace tfx;

extern "C" {

int tfx_lsqr_begin(tfx_ctx *ctx, double rmin, double gamma, double target_misfit, const double *b_data, int nblocks,
                   const float *const *diag, const double *const *rhs_blocks)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !b_data) return fail(TFX_E_ARG, "tfx_lsqr_begin: null argument");
    TiledMatrix &m = ctx->mat;
    if (!m.valid) return fail(TFX_E_STATE, "tfx_lsqr_begin: no matrix");
    if (nblocks < 0 || (nblocks > 0 && (!diag || !rhs_blocks))) return fail(TFX_E_ARG, "bad constraint blocks");
    TFX_HIP(hipSetDevice(ctx->device));
    if (!ctx->lsqr) {
        ctx->lsqr = new LsqrState();
        TFX_HIP(hipHostMalloc((void **)&ctx->lsqr->h_sc, sizeof(Scalars), hipHostMallocDefault));
    }
    LsqrState *L = ctx->lsqr;
    hipStream_t s = ctx->stream;
    const bool have_cons = ctx->cons.valid;
    const int64_t s_rows = ctx->total_rows(), s_cols = ctx->total_cols();       // both problems of a joint inversion
    if (have_cons && ctx->cons.ncols != s_cols) return fail(TFX_E_STATE, "constraint matrix has %lld columns, S has %lld",
                                                            (long long)ctx->cons.ncols, (long long)s_cols);
    L->nrows_data = s_rows;
    L->nrows = s_rows + (have_cons ? ctx->cons.nrows : 0);
    L->ncols = s_cols;
    L->nblocks = nblocks;
    L->rmin = rmin;
    L->gamma = gamma;
    L->target_misfit = target_misfit;
    L->iter = 0;
    L->r = 1.0;
    L->active = true;
    L->exact = false;
    L->finished = false;
    L->next_ready = false;
    const int64_t nc = L->ncols, nr = L->nrows;
    TFX_TRY(L->u.ensure((size_t)nr + 1));
    TFX_TRY(L->v.ensure((size_t)nc));
    TFX_TRY(L->w.ensure((size_t)nc));
    TFX_TRY(L->x.ensure((size_t)nc));
    TFX_TRY(L->uc.ensure((size_t)std::max<int64_t>(1, nblocks * nc)));
    TFX_TRY(L->diag.ensure((size_t)std::max<int64_t>(1, nblocks * nc)));
    TFX_TRY(L->red.ensure(RED_BLOCKS));
    TFX_TRY(L->cnt.ensure(4));
    TFX_HIP(hipMemsetAsync(L->cnt.p, 0, 4 * sizeof(unsigned int), ctx->stream));
    if (ctx->spatial_unknowns) {
        const int64_t n123 = (int64_t)ctx->wd_n1 * ctx->wd_n2 * ctx->wd_n3;
        if (ctx->multi()) {
            // this rank's unknowns are the cells [col_begin, col_begin + nloc) of every model component.  A partition that is wrong on
            // THIS rank only (not set, not matching the local matrix) is not an early return: the rank publishes a "bad" flag in the
            // same all-reduce that builds the table, and all ranks fail together after it - nobody is left inside the collective
            const bool local_bad = ctx->wd_col_begin < 0 || ctx->wd_ncomp <= 0 || nc % std::max(1, ctx->wd_ncomp) != 0 ||
                                   ctx->wd_col_begin + nc / std::max(1, ctx->wd_ncomp) > n123;
            // first cell and cell count of every rank: each rank puts its own pair into a zero vector, the sum is the table (exact:
            // integers < 2^53).  EVERY rank validates the WHOLE table - the same data on all of them, so a range that is out of order,
            // overlapping or does not tile the model makes all ranks fail together instead of leaving some inside a collective with a
            // negative length
            const int P = ctx->nranks;
            TFX_TRY(ctx->vx.ensure((size_t)std::max<int64_t>(3 * P, nc)));
            std::vector<double> hb((size_t)(3 * P), 0.0);
            if (!local_bad) {
                hb[(size_t)ctx->rank] = (double)ctx->wd_col_begin;
                hb[(size_t)(P + ctx->rank)] = (double)(nc / ctx->wd_ncomp);
            }
            hb[(size_t)(2 * P + ctx->rank)] = local_bad ? 1.0 : 0.0;
            TFX_TRY(copy_any(ctx->vx.p, hb.data(), hb.size() * sizeof(double), s));
            TFX_TRY(allreduce(ctx, ctx->vx.p, 3 * P));
            TFX_TRY(copy_any(hb.data(), ctx->vx.p, hb.size() * sizeof(double), s));
            for (int r = 0; r < P; ++r)
                if (hb[(size_t)(2 * P + r)] != 0.0)
                    return fail(TFX_E_STATE, "multi-rank WAVELET_DOMAIN = F: rank %d has no column partition that matches its matrix "
                                "(tfx_lsqr_set_partition not called, or %lld local columns do not divide into its components / exceed the grid)",
                                r, (long long)nc);
            TFX_TRY(L->twf.ensure((size_t)(ctx->wd_ncomp * n123)));
            L->g_counts.assign((size_t)P, 0);
            L->g_displs.assign((size_t)P, 0);
            int64_t expect = 0;
            for (int r = 0; r < P; ++r) {
                L->g_displs[(size_t)r] = (int64_t)hb[(size_t)r];
                L->g_counts[(size_t)r] = (int64_t)hb[(size_t)(P + r)];
                if (L->g_counts[(size_t)r] < 0 || L->g_displs[(size_t)r] != expect)
                    return fail(TFX_E_STATE, "the ranks' column ranges do not tile the model: rank %d starts at cell %lld with %lld cells, expected start %lld",
                                r, (long long)L->g_displs[(size_t)r], (long long)L->g_counts[(size_t)r], (long long)expect);
                expect += L->g_counts[(size_t)r];
            }
            if (expect != n123)
                return fail(TFX_E_STATE, "the ranks' column ranges cover %lld of %lld cells", (long long)expect, (long long)n123);
        } else {
            if (n123 <= 0 || nc % n123 != 0)   // ncolumns = nmodel_components * nelements (wavelet_utils.F90:37-72 loops the components)
                return fail(TFX_E_STATE, "WAVELET_DOMAIN = F needs the whole model on this rank (ncolumns %lld is not a multiple of n1*n2*n3)", (long long)nc);
            ctx->wd_nvec = nc / n123;
        }
        TFX_TRY(L->tw.ensure((size_t)nc));
    }
    TFX_TRY(L->sc.ensure(1));
    TFX_HIP(hipMemsetAsync(L->sc.p, 0, sizeof(Scalars), s));
    TFX_HIP(hipMemsetAsync(L->x.p, 0, (size_t)nc * sizeof(double), s));                  // :120
    TFX_HIP(hipMemsetAsync(L->v.p, 0, (size_t)nc * sizeof(double), s));
    TFX_HIP(hipMemcpyAsync(L->u.p, b_data, (size_t)L->nrows_data * sizeof(double), hipMemcpyDefault, s));
    if (have_cons)
        TFX_HIP(hipMemcpyAsync(L->u.p + L->nrows_data, ctx->cons_rhs.p, (size_t)ctx->cons.nrows * sizeof(double), hipMemcpyDeviceToDevice, s));
    for (int b = 0; b < nblocks; ++b) {
        if (!diag[b] || !rhs_blocks[b]) return fail(TFX_E_ARG, "null constraint block %d", b);
        TFX_HIP(hipMemcpyAsync(L->diag.p + (size_t)b * nc, diag[b], (size_t)nc * sizeof(float), hipMemcpyDefault, s));
        TFX_HIP(hipMemcpyAsync(L->uc.p + (size_t)b * nc, rhs_blocks[b], (size_t)nc * sizeof(double), hipMemcpyDefault, s));
    }
    if (target_misfit > 0.0) {                                                            // :98-106
        TFX_TRY(L->b0.ensure((size_t)L->nrows_data));
        TFX_TRY(L->sx.ensure((size_t)L->nrows_data));
        TFX_HIP(hipMemcpyAsync(L->b0.p, L->u.p, (size_t)L->nrows_data * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    // ||u_cons,loc||^2 -> u[nrows], summed over ranks
    {
        const int64_t n = (int64_t)nblocks * nc;
        const int g = grid_for(std::max<int64_t>(1, n));
        LAUNCH(k_sumsq, g, L->uc.p, n, L->red.p);
        LAUNCH(k_final_sum, 1, L->red.p, g, L->u.p + nr);
        TFX_HIP(hipGetLastError());
        TFX_TRY(allreduce(ctx, L->u.p + nr, 1));
    }
    TFX_TRY(norm_u(ctx, L));                                                             // :123-129
    TFX_TRY(read_scalars(ctx, L));
    if (L->h_sc->beta == 0.0) {                                                           // |b| = 0: the model is exact
        L->exact = true;
        L->finished = true;
        L->r = 0.0;
        return 0;
    }
    TFX_TRY(scale_u(ctx, L));
    TFX_TRY(adjoint_and_alpha(ctx, L));                                                  // :137-150
    LAUNCH(k_init_scalars, 1, L->sc.p, L->rmin);                                         // :134, :155-156
    LAUNCH(k_copy, grid_for(nc), L->w.p, L->v.p, nc);                                    // :157
    TFX_HIP(hipGetLastError());
    TFX_TRY(read_scalars(ctx, L));
    if (L->h_sc->v_zero) return fail(TFX_E_NUMERIC, "Could not normalize initial v, zero denominator!");
    return 0;
}

// WAVELET_DOMAIN (joint_inverse_problem.F90:189-198): 1 = unknowns in the wavelet domain (default; S applied as is),
// 0 = spatial unknowns: every product with S goes through the 3-D wavelet transform (lsqr_solver2.F90:200-206, :228-234)
int tfx_lsqr_set_wavelet_domain(tfx_ctx *ctx, int wavelet_domain, int n1, int n2, int n3, int wavelet_type)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    if (wavelet_domain || wavelet_type == 0) {
        ctx->spatial_unknowns = false;
        return 0;
    }
    if (wavelet_type != 1 && wavelet_type != 2) return fail(TFX_E_ARG, "Unknown wavelet type!");
    if (n1 <= 0 || n2 <= 0 || n3 <= 0) return fail(TFX_E_ARG, "bad grid size");
    ctx->spatial_unknowns = true;
    ctx->wd_n1 = n1; ctx->wd_n2 = n2; ctx->wd_n3 = n3; ctx->wd_type = wavelet_type;
    return 0;
}

// Multi-rank WAVELET_DOMAIN = F: where this rank's unknowns sit in the full model (cells [col_begin, col_begin + nloc) of each of
// the ncomponents model components, all problems counted)
int tfx_lsqr_set_partition(tfx_ctx *ctx, int64_t col_begin, int ncomponents)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    if (col_begin < 0 || ncomponents < 1) return fail(TFX_E_ARG, "tfx_lsqr_set_partition: bad arguments");
    ctx->wd_col_begin = col_begin;
    ctx->wd_ncomp = ncomponents;
    return 0;
}

int tfx_lsqr_iterate(tfx_ctx *ctx, int k, int *done_out, double *r_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !ctx->lsqr || !ctx->lsqr->active) return fail(TFX_E_STATE, "tfx_lsqr_iterate: call tfx_lsqr_begin first");
    LsqrState *L = ctx->lsqr;
    hipStream_t s = ctx->stream;
    TFX_HIP(hipSetDevice(ctx->device));
    const int64_t nc = L->ncols, nr = L->nrows;
    int done = 0;
    // Iterations are queued LSQR_CHUNK at a time; the scalars come back to the host once per chunk.  The loop's exit tests
    // (:163 r <= rmin, :251-254 rho = 0, :286-289 small rhobar) run on the device inside the rotation, and an iteration queued
    // behind a met exit test leaves x, w and the scalars untouched (Scalars::stop / skip), so the result is the one of the
    // iteration-by-iteration loop - without a host round trip in every iteration of a launch-bound system.  With a target
    // misfit (:168-189, a host decision per iteration) the chunk is one iteration.
    constexpr int LSQR_CHUNK = 16;
    while (done < k && !L->finished && L->r > L->rmin) {                                  // :163
        const int chunk = L->target_misfit > 0.0 ? 1 : std::min(k - done, LSQR_CHUNK);
        if (L->target_misfit > 0.0) {                                                     // :168-189
            const int64_t nd = L->nrows_data;
            if (ctx->spatial_unknowns) {                                                  // :171-176
                LAUNCH(k_copy, grid_for(nc), L->tw.p, L->x.p, nc);
                TFX_TRY(transform_slice(ctx, L, 1));
                TFX_TRY(S_forward(ctx, L->tw.p, L->sx.p, 0));
            } else
            TFX_TRY(S_forward(ctx, L->x.p, L->sx.p, 0));
            TFX_TRY(allreduce(ctx, L->sx.p, nd));
            const int g = grid_for(nd);
            LAUNCH(k_misfit, g, L->sx.p, L->b0.p, nd, L->red.p);
            LAUNCH(k_final_sum, 1, L->red.p, g, &L->sc.p->misfit_ss);
            TFX_TRY(read_scalars(ctx, L));
            if (std::sqrt(L->h_sc->misfit_ss / (double)nd) <= L->target_misfit) { L->finished = true; break; }
        }
        for (int j = 0; j < chunk; ++j) {
            // u = -alpha u (rank 0) | 0 (others), then u += S_loc v                       :194-209
            const bool prepared = L->next_ready;          // the previous iteration's tail launch has done :194-198 and :211 already
            L->next_ready = false;
            if (!prepared) LAUNCH(k_scale, grid_for(nr), L->u.p, nr, &L->sc.p->alpha, ctx->rank == 0 ? 1 : 2);
            if (ctx->spatial_unknowns) {                                                  // :200-209
                LAUNCH(k_copy, grid_for(nc), L->tw.p, L->v.p, nc);
                TFX_TRY(transform_slice(ctx, L, 1));
                TFX_TRY(S_forward(ctx, L->tw.p, L->u.p, 1));
            } else {
                TFX_TRY(S_forward(ctx, L->v.p, L->u.p, 1));
            }
            if (ctx->cons.valid) TFX_TRY(spmv_dev(ctx, ctx->cons, L->v.p, L->u.p + L->nrows_data, 1));   // :211 (general C rows)
            if (!prepared) {                                                              // :211 (diagonal blocks, local)
                const int g = grid_for(nc);
                if (g <= ONE_LAUNCH_MAX_BLOCKS) {
                    LAUNCH(k_cons_forward, g, L->uc.p, L->diag.p, L->v.p, nc, L->nblocks, L->sc.p, L->red.p, L->cnt.p + 2, L->u.p + nr);
                } else {
                    LAUNCH(k_cons_forward, g, L->uc.p, L->diag.p, L->v.p, nc, L->nblocks, L->sc.p, L->red.p, L->cnt.p + 2, (double *)nullptr);
                    LAUNCH(k_final_sum, 1, L->red.p, g, L->u.p + nr);
                }
            }
            TFX_HIP(hipGetLastError());
            TFX_TRY(allreduce(ctx, L->u.p, nr + 1));                                      // :214
            TFX_TRY(norm_u(ctx, L));                                                      // :218
            {                                                                             // u, u_cons /= beta ; v = -beta v   (:218-225)
                const int64_t nuc = (int64_t)L->nblocks * nc;
                LAUNCH(k_scale_u_uc_v, grid_for(std::max(std::max(nr, nuc), nc)), L->u.p, nr, L->uc.p, nuc, L->v.p, nc, L->sc.p);
            }
            const bool fused = !ctx->multi();                      // no reduction between sum_v and alpha
            TFX_TRY(adjoint_and_alpha(ctx, L, fused, !fused));                            // :228-241, :248-266 (alpha and the rotation)
            if (ctx->lsqr_merge_tail) {           // v / alpha, :241, :269-274 and the next iteration's :194-198, :211 in one launch
                const int g = grid_for(nc);
                const int u_mode = ctx->rank == 0 ? 1 : 2;
                if (g <= ONE_LAUNCH_MAX_BLOCKS) {
                    LAUNCH(k_update_xw_next, g, L->v.p, L->w.p, L->x.p, nc, L->sc.p, L->gamma, L->uc.p, L->diag.p, L->nblocks, L->red.p, L->cnt.p + 2,
                           L->u.p + nr, L->u.p, nr, u_mode);
                } else {
                    LAUNCH(k_update_xw_next, g, L->v.p, L->w.p, L->x.p, nc, L->sc.p, L->gamma, L->uc.p, L->diag.p, L->nblocks, L->red.p, L->cnt.p + 2,
                           (double *)nullptr, L->u.p, nr, u_mode);
                    LAUNCH(k_final_sum, 1, L->red.p, g, L->u.p + nr);
                }
                L->next_ready = true;
            } else {
                LAUNCH(k_update_xw, grid_for(nc), L->v.p, L->w.p, L->x.p, nc, L->sc.p, 1.0, L->gamma);   // v / alpha, :241, :269-274
            }
            TFX_HIP(hipGetLastError());
        }
        TFX_TRY(read_scalars(ctx, L));
        const int did = L->h_sc->iters - L->iter;                                         // iterations that counted
        L->iter = L->h_sc->iters;
        done += did;
        L->r = L->h_sc->r;
        if (L->h_sc->rho_zero) { L->finished = true; break; }                             // :251-254
        if (std::fabs(L->h_sc->rhobar) < 1.e-30) { L->finished = true; break; }           // :286-289
        if (did < chunk) break;                                                           // r <= rmin inside the chunk
    }
    if (done_out) *done_out = done;
    if (r_out) *r_out = L->r;
    return 0;
}

int tfx_lsqr_end(tfx_ctx *ctx, double *x_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !ctx->lsqr || !ctx->lsqr->active) return fail(TFX_E_STATE, "tfx_lsqr_end: no solve in progress");
    LsqrState *L = ctx->lsqr;
    if (x_out) TFX_TRY(copy_any(x_out, L->x.p, (size_t)L->ncols * sizeof(double), ctx->stream));
    L->active = false;
    return 0;
}

int tfx_lsqr_solve(tfx_ctx *ctx, int niter, double rmin, double gamma, double target_misfit, const double *b_data,
                   int nblocks, const float *const *diag, const double *const *rhs_blocks, double *x_out, int *iters_out,
                   double *r_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(tfx_lsqr_begin(ctx, rmin, gamma, target_misfit, b_data, nblocks, diag, rhs_blocks));
    int done = 0;
    double r = 1.0;
    TFX_TRY(tfx_lsqr_iterate(ctx, niter, &done, &r));
    if (iters_out) *iters_out = done;
    if (r_out) *r_out = r;
    return tfx_lsqr_end(ctx, x_out);
}

// data_calc = S xw / problem_weight / data_weight, the divisions on the device (model.F90:295-302)
__global__ void k_calc_data_finish(double *__restrict__ d, int64_t n, double problem_weight, const double *__restrict__ dw)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = d[i] / problem_weight;
        if (dw) v = v / dw[i];
        d[i] = v;
    }
}

int tfx_calc_data(tfx_ctx *ctx, const double *xw_local, double problem_weight, const double *data_weight,
                  double *data_calc)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !xw_local || !data_calc) return fail(TFX_E_ARG, "tfx_calc_data: null argument");
    TiledMatrix &m = ctx->selmat();        // the selected problem: part_mult_vector with its line_start / param_shift
    if (!m.valid) return fail(TFX_E_STATE, "tfx_calc_data: no matrix");
    if (problem_weight == 0.0) return fail(TFX_E_NUMERIC, "Zero problem weight in model_calculate_data!");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // scratch kept by the ctx: x, b and the data weights (no allocation per call once they have their size)
    TFX_TRY(ctx->vx.ensure((size_t)m.ncols));
    TFX_TRY(ctx->vb.ensure((size_t)m.nrows));
    TFX_HIP(hipMemcpyAsync(ctx->vx.p, xw_local, (size_t)m.ncols * sizeof(double), hipMemcpyDefault, s));
    const double *dw = nullptr;
    if (data_weight) {
        TFX_TRY(ctx->vw.ensure((size_t)m.nrows));
        TFX_HIP(hipMemcpyAsync(ctx->vw.p, data_weight, (size_t)m.nrows * sizeof(double), hipMemcpyDefault, s));
        dw = ctx->vw.p;
    }
    TFX_TRY(spmv_dev(ctx, ctx->vx.p, ctx->vb.p, 0));                                      // model.F90:285-286
    TFX_TRY(allreduce(ctx, ctx->vb.p, m.nrows));                                          // model.F90:290
    LAUNCH(k_calc_data_finish, grid_for(m.nrows), ctx->vb.p, m.nrows, problem_weight, dw);
    TFX_HIP(hipGetLastError());
    TFX_TRY(copy_any(data_calc, ctx->vb.p, (size_t)m.nrows * sizeof(double), s));
    return 0;
}

}  // extern "C"
