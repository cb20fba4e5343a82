// build.hip - sensitivity-kernel build pipeline on the device:
//   prism rows   graviprism_z                 src/forward/gravmag/grav/gravity_field.f90:131-195
//   weights      calculate_depth_weight (1)   src/forward/gravmag/weights_gravmag.f90:71-79,170-250
//   wavelets     Haar3D / DaubD43D (+inverse) src/utils/wavelet_transform.F90:75-498
//   threshold    quicksort + order statistic  src/forward/gravmag/sensitivity_gravmag.F90:240-256, src/utils/sort.f90
//   compaction   keep |c| > thr, fp32 cast    src/forward/gravmag/sensitivity_gravmag.F90:258-272
// and tfx_build_kernel_grav = calculate_and_write_sensit + read_sensitivity_kernel without the disk round trip
// (sensitivity_gravmag.F90:82-410, :648-883).
//
// Compiled with -ffp-contract=off: the lifting steps and the prism polynomial are evaluated with the reference's
// operation order (separate multiply / add), so wavelets are bit-identical to the reference and prism rows differ
// only through the device libm (atan2 / log, <= 2 ulp).
#include "common.h"
#include "fastmath.h"
#include <algorithm>
#include <chrono>
#include <cmath>

namespace tfx {

// =============================================================================================================
// prism rows
// =============================================================================================================
// gravity_field.f90:26 - `G_grav = 6.674e-11` is a default-real literal: the value used is (double)(float)6.674e-11.
__device__ __forceinline__ double g_grav() { return (double)6.674e-11f; }

// The reduction tables of fastmath.h live in the workgroup's LDS: one static array shared by every device function of a kernel
// (init_math_tables() at kernel entry, then a __syncthreads() before the first dlog / datan2).
__device__ __forceinline__ FastMathTables math_tables()
{
    __shared__ double s_math_tab[FASTMATH_TABLE_DOUBLES];
    return FastMathTables{s_math_tab, s_math_tab + 3 * TFX_LOG_TAB_N};
}
__device__ __forceinline__ void init_math_tables()
{
    double *lds = const_cast<double *>(math_tables().logt);
    for (int i = threadIdx.x; i < 3 * TFX_LOG_TAB_N; i += blockDim.x) lds[i] = tfx_log_tab[i];
    for (int i = threadIdx.x; i < 8 * TFX_ATAN_TAB_N; i += blockDim.x) lds[3 * TFX_LOG_TAB_N + i] = tfx_atan_tab[i];
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also makes the workgroup's GLOBAL stores visible, i.e. it waits for
// every outstanding store (s_waitcnt vmcnt(0)): in the row generators that drained the non-temporal stores of an observation's rows -
// which nothing in the kernel ever reads back - at the first barrier of the next observation, so the HBM write latency of each cell phase
// was exposed instead of running under the next node phase.
// ONLY LDS IS ORDERED: a global store issued before it may still be in flight after it.  Nothing in these kernels reads back what it
// stored; a change that does (rows, sumsq) needs __syncthreads() at that point.  The two instructions are the gfx9 encoding (CDNA:
// lgkmcnt counts LDS operations, s_barrier is the workgroup barrier); common.h refuses to compile the device code for anything else.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#ifdef TFX_LIBM_TRANSCENDENTALS          // A/B builds only (make EXTRA=-DTFX_LIBM_TRANSCENDENTALS): the device libm instead of fastmath.h
__device__ __forceinline__ double dlog(double x) { return log(x); }
__device__ __forceinline__ double datan2(double y, double x) { return atan2(y, x); }
#else
__device__ __forceinline__ double dlog(double x) { return fast_log(x, math_tables()); }
__device__ __forceinline__ double datan2(double y, double x) { return fast_atan2(y, x, math_tables()); }
#endif

// One corner of the prism integral (gravity_field.f90:165-186): returns ZZ*atan2'(XX*YY, ZZ*R) - XX*log(R+YY) - YY*log(R+XX)
// and flags R+XX <= 0 / R+YY <= 0 (:176-181).  Shared by the general and the tensor-grid kernel so both produce the
// same bits.
// log / atan2: fastmath.h (table-reduced, <= 1 ulp class like the device libm, a third of its instructions).
__device__ __forceinline__ double corner_term(double XX, double YY, double ZZ, int &bad)
{
    const double twopi = 2.0 * 3.14159265358979323846;
    const double Rs = sqrt(XX * XX + YY * YY + ZZ * ZZ);                                            // :165
    double arg3 = datan2(XX * YY, ZZ * Rs);                                                         // :167
    if (arg3 < 0) arg3 = arg3 + twopi;
    double arg4 = Rs + XX;
    double arg5 = Rs + YY;
    if (arg4 <= 0.) bad |= 1;
    if (arg5 <= 0.) bad |= 2;
    arg4 = dlog(arg4);
    arg5 = dlog(arg5);
    return ZZ * arg3 - XX * arg5 - YY * arg4;                                                       // :186
}

constexpr int PRISM_MAX_BATCH = 64;      // observations per launch (bounded by the build's batch size, <= 32)

// General grid (six independent arrays): one thread per cell, loops over the observation batch (coordinates are
// wave-uniform scalar loads).  rows[o*N + p] = G*gz (* cw[p] when cw != null: apply_column_weight,
// sensitivity_gravmag.F90:1042-1054).
__global__ __launch_bounds__(256) void k_prism_gz(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                                                  const double *__restrict__ Y1, const double *__restrict__ Y2,
                                                  const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                  int nobs, const double *__restrict__ xd, const double *__restrict__ yd,
                                                  const double *__restrict__ zd, const double *__restrict__ cw,
                                                  double *__restrict__ rows, int *__restrict__ err,
                                                  double *__restrict__ sumsq /* [nobs][gridDim.x] or null */)
{
    __shared__ double s_sq[PRISM_MAX_BATCH];
    init_math_tables();
    if (sumsq)
        for (int o = threadIdx.x; o < nobs; o += blockDim.x) s_sq[o] = 0.0;
    __syncthreads();
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x; p0 < N; p0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = p0 + threadIdx.x;
        const bool active = p < N;
        const int64_t pc = active ? p : N - 1;
        const double x1 = X1[pc], x2 = X2[pc], y1 = Y1[pc], y2 = Y2[pc], z1 = Z1[pc], z2 = Z2[pc];
        const double w = cw ? cw[pc] : 1.0;
        for (int o = 0; o < nobs; ++o) {
            double XX[2], YY[2], ZZ[2];
            XX[0] = xd[o] - x1; XX[1] = xd[o] - x2;                     // :151-156
            YY[0] = yd[o] - y1; YY[1] = yd[o] - y2;
            ZZ[0] = zd[o] - z1; ZZ[1] = zd[o] - z2;
            double gz = 0.0;
            int bad = 0;
#pragma unroll
            for (int K = 0; K < 2; ++K)
#pragma unroll
                for (int L = 0; L < 2; ++L)
#pragma unroll
                    for (int M = 0; M < 2; ++M) {
                        const double dmu = ((K + L + M) & 1) ? 1.0 : -1.0;   // signo(K)*signo(L)*signo(M), signo = (-1, +1)
                        gz = gz + dmu * corner_term(XX[K], YY[L], ZZ[M], bad);
                    }
            if (bad && active) atomicOr(err, bad);
            double v = g_grav() * gz;                                                               // :192
            if (cw) v = v * w;
            if (active) rows[(int64_t)o * N + p] = v;
            if (sumsq) {                                    // cost_full (sensitivity_gravmag.F90:234), diagnostic only
                double sq = active ? v * v : 0.0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) sq += __shfl_down(sq, d);
                if ((threadIdx.x & 63) == 0) atomicAdd(&s_sq[o], sq);
            }
        }
    }
    if (sumsq) {
        __syncthreads();
        for (int o = threadIdx.x; o < nobs; o += blockDim.x) sumsq[(int64_t)o * gridDim.x + blockIdx.x] = s_sq[o];
    }
}

// Tensor-product grid (X1/X2 depend on i only, Y on j, Z on k, and neighbouring cells share their faces bit-for-bit -
// what the reference's grid files describe in practice): the corner term depends only on the NODE, and a node is
// shared by up to 8 cells.  A workgroup evaluates the (TX+1)(TY+1)(TZ+1) nodes of its TX x TY x TZ cell tile once into
// LDS (1.3 transcendental evaluations per cell instead of 8) and then forms each cell's 8-term sum in the reference's
// order (K outer, L, M inner), so the result is bit-identical to k_prism_gz.
constexpr int PT_X = 32, PT_Y = 8, PT_Z = 8;
constexpr int PT_NODES = (PT_X + 1) * (PT_Y + 1) * (PT_Z + 1);
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_prism_gz_tensor(int nx, int ny, int nz, const double *__restrict__ xe,
                                                         const double *__restrict__ ye, const double *__restrict__ ze,
                                                         int nobs, const double *__restrict__ xd,
                                                         const double *__restrict__ yd, const double *__restrict__ zd,
                                                         const double *__restrict__ cw, double *__restrict__ rows,
                                                         int *__restrict__ err, double *__restrict__ sumsq, int ntiles)
{
    __shared__ double T[PT_NODES];
    __shared__ double s_w[4];
    init_math_tables();
    // the node / cell index arithmetic does not depend on the observation: done once per workgroup, kept in LDS
    __shared__ double s_xe[PT_X + 1], s_ye[PT_Y + 1], s_ze[PT_Z + 1];
    __shared__ int s_node[PT_NODES];                      // LDS slot | a << 12 | b << 18 | c << 22
    const int tiles_x = (nx + PT_X - 1) / PT_X, tiles_y = (ny + PT_Y - 1) / PT_Y;
    int bad = 0;
    // a launch of fewer workgroups than tiles (the build's overlap mode: two per CU, leaving registers and LDS to the HBM-bound
    // kernels of the main stream) walks the tiles with the grid's stride
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (tile != (int)blockIdx.x) lds_barrier();           // the previous tile's T / s_node are still being read
    const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, bz = tile / (tiles_x * tiles_y);
    const int i0 = bx * PT_X, j0 = by * PT_Y, k0 = bz * PT_Z;
    const int cx = min(PT_X, nx - i0), cy = min(PT_Y, ny - j0), cz = min(PT_Z, nz - k0);
    const int64_t N = (int64_t)nx * ny * nz;
    const int nnode = (cx + 1) * (cy + 1) * (cz + 1);
    for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
        const int a = n % (cx + 1), b = (n / (cx + 1)) % (cy + 1), c = n / ((cx + 1) * (cy + 1));
        s_node[n] = ((c * (PT_Y + 1) + b) * (PT_X + 1) + a) | (a << 12) | (b << 18) | (c << 22);
    }
    if (threadIdx.x <= cx) s_xe[threadIdx.x] = xe[i0 + threadIdx.x];
    if (threadIdx.x <= cy) s_ye[threadIdx.x] = ye[j0 + threadIdx.x];
    if (threadIdx.x <= cz) s_ze[threadIdx.x] = ze[k0 + threadIdx.x];
    // cells of this thread: the column (ta, tb) of the tile, one cell per z layer - the LDS slot of a cell's (0, 0, 0) node and its
    // matrix column are affine in the layer, so nothing per cell stays in registers over the observation loop (the kernel's
    // occupancy is set by its VGPRs: 4 waves per SIMD at <= 128)
    static_assert(PT_X * PT_Y == 256, "one thread per (x, y) column of the tile");
    constexpr int LAYER = (PT_Y + 1) * (PT_X + 1);
    const int ta = threadIdx.x % PT_X, tb = threadIdx.x / PT_X;
    const bool col_ok = ta < cx && tb < cy;
    const int slot0 = tb * (PT_X + 1) + ta;
    const int64_t col0 = ((int64_t)k0 * ny + (j0 + tb)) * nx + (i0 + ta), lay = (int64_t)ny * nx;
    for (int o = 0; o < nobs; ++o) {
        const double xo = xd[o], yo = yd[o], zo = zd[o];
        lds_barrier();
        for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
            const int code = s_node[n];
            T[code & 4095] = corner_term(xo - s_xe[(code >> 12) & 63], yo - s_ye[(code >> 18) & 15], zo - s_ze[code >> 22], bad);
        }
        lds_barrier();
        double sq = 0.0;
        double *out = rows + (int64_t)o * N;
        const double *cwo = cw;
        asm volatile("" : "+s"(cwo));       // the weights are re-read per observation (L2 hits) instead of held in 16 VGPRs over the node loop
        if (col_ok) {
#pragma unroll
            for (int j = 0; j < PT_Z; ++j) {
                if (j >= cz) break;
                const double *t0 = T + slot0 + j * LAYER;
                double gz = 0.0;
#pragma unroll
                for (int K = 0; K < 2; ++K)
#pragma unroll
                    for (int L = 0; L < 2; ++L)
#pragma unroll
                        for (int M = 0; M < 2; ++M) {
                            const double dmu = ((K + L + M) & 1) ? 1.0 : -1.0;
                            gz = gz + dmu * t0[(M * (PT_Y + 1) + L) * (PT_X + 1) + K];
                        }
                double v = g_grav() * gz;
                const int64_t col = col0 + j * lay;
                if (cw) v = v * cwo[col];
                __builtin_nontemporal_store(v, &out[col]);       // 2 GB per batch: the next reader (wavelet x pass) finds nothing in L2 anyway
                sq = fma(v, v, sq);
            }
        }
        if (sumsq) {                                        // cost_full (sensitivity_gravmag.F90:234), fixed order
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) sq += __shfl_down(sq, d);
            if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sq;
            lds_barrier();
            if (threadIdx.x == 0) sumsq[(int64_t)o * ntiles + tile] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        }
    }
    }
    if (bad) atomicOr(err, bad);
}

// =============================================================================================================
// graviprism_full (gravity_field.f90:41-126): the three components of the attraction, sub-rows X, Y, Z per observation
// (LineX, LineY, LineZ).  The reference declares it public and never calls it (sensitivity_gravmag.F90:193-213 maps data_type 1 to
// graviprism_z); here it is tfx_prism_rows / tfx_build_kernel with (problem_type 1, data_type 1, ndata_components 3).
// =============================================================================================================
// One corner (:77-112): the three bracketed terms, and the three abort tests R+X <= 0 (1), R+Y <= 0 (2), R+Z <= 0 (64) (:96-104).
// tz is corner_term()'s expression operation by operation, so the Z sub-row has the bits of the graviprism_z row.
__device__ __forceinline__ void corner_term3(double XX, double YY, double ZZ, double &tx, double &ty, double &tz, int &bad)
{
    const double twopi = 2.0 * 3.14159265358979323846;
    const double Rs = sqrt(XX * XX + YY * YY + ZZ * ZZ);                                            // :77
    double arg1 = datan2(YY * ZZ, XX * Rs);                                                         // :79-81
    double arg2 = datan2(XX * ZZ, YY * Rs);
    double arg3 = datan2(XX * YY, ZZ * Rs);
    if (arg1 < 0) arg1 = arg1 + twopi;                                                              // :83-91
    if (arg2 < 0) arg2 = arg2 + twopi;
    if (arg3 < 0) arg3 = arg3 + twopi;
    double arg4 = Rs + XX;                                                                          // :93-95
    double arg5 = Rs + YY;
    double arg6 = Rs + ZZ;
    if (arg4 <= 0.) bad |= 1;                                                                       // :96-104
    if (arg5 <= 0.) bad |= 2;
    if (arg6 <= 0.) bad |= 64;
    arg4 = dlog(arg4);
    arg5 = dlog(arg5);
    arg6 = dlog(arg6);
    tx = XX * arg1 - YY * arg6 - ZZ * arg5;                                                         // :110-112
    ty = YY * arg2 - ZZ * arg4 - XX * arg6;
    tz = ZZ * arg3 - XX * arg5 - YY * arg4;
}

// General grid (six arrays): one thread per cell; rows[(o*3 + c)*N + p], c = X, Y, Z.
__global__ __launch_bounds__(256) void k_prism_g3(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                                                  const double *__restrict__ Y1, const double *__restrict__ Y2,
                                                  const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                  int nobs, const double *__restrict__ xd, const double *__restrict__ yd,
                                                  const double *__restrict__ zd, const double *__restrict__ cw,
                                                  double *__restrict__ rows, int *__restrict__ err,
                                                  double *__restrict__ sumsq /* [nobs*3][gridDim.x] or null */)
{
    __shared__ double s_sq[PRISM_MAX_BATCH];
    init_math_tables();
    if (sumsq)
        for (int o = threadIdx.x; o < nobs * 3; o += blockDim.x) s_sq[o] = 0.0;
    __syncthreads();
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x; p0 < N; p0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = p0 + threadIdx.x;
        const bool active = p < N;
        const int64_t pc = active ? p : N - 1;
        const double x1 = X1[pc], x2 = X2[pc], y1 = Y1[pc], y2 = Y2[pc], z1 = Z1[pc], z2 = Z2[pc];
        const double w = cw ? cw[pc] : 1.0;
        for (int o = 0; o < nobs; ++o) {
            double XX[2], YY[2], ZZ[2];
            XX[0] = xd[o] - x1; XX[1] = xd[o] - x2;                     // :61-66
            YY[0] = yd[o] - y1; YY[1] = yd[o] - y2;
            ZZ[0] = zd[o] - z1; ZZ[1] = zd[o] - z2;
            double g3[3] = {0.0, 0.0, 0.0};
            int bad = 0;
#pragma unroll
            for (int K = 0; K < 2; ++K)
#pragma unroll
                for (int L = 0; L < 2; ++L)
#pragma unroll
                    for (int M = 0; M < 2; ++M) {
                        const double dmu = ((K + L + M) & 1) ? 1.0 : -1.0;
                        double tx, ty, tz;
                        corner_term3(XX[K], YY[L], ZZ[M], tx, ty, tz, bad);
                        g3[0] = g3[0] + dmu * tx;
                        g3[1] = g3[1] + dmu * ty;
                        g3[2] = g3[2] + dmu * tz;
                    }
            if (bad && active) atomicOr(err, bad);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double v = g_grav() * g3[c];                                                        // :118-120
                if (cw) v = v * w;
                const int sub = o * 3 + c;
                if (active) rows[(int64_t)sub * N + p] = v;
                if (sumsq) {
                    double sq = active ? v * v : 0.0;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) sq += __shfl_down(sq, d);
                    if ((threadIdx.x & 63) == 0) atomicAdd(&s_sq[sub], sq);
                }
            }
        }
    }
    if (sumsq) {
        __syncthreads();
        for (int o = threadIdx.x; o < nobs * 3; o += blockDim.x) sumsq[(int64_t)o * gridDim.x + blockIdx.x] = s_sq[o];
    }
}

// Tensor-product grid: like k_prism_gz_tensor the three corner terms depend on the node only; a workgroup evaluates the
// (G3_X+1)(G3_Y+1)(G3_Z+1) nodes of its tile once (1.4 x {sqrt, 3 atan2, 3 log} per cell instead of 8) into three LDS planes and every
// cell sums its 8 nodes per component in the reference's order -> the bits of k_prism_g3, and the Z sub-row those of k_prism_gz(_tensor).
constexpr int G3_X = 32, G3_Y = 8, G3_Z = 4;
constexpr int G3_NODES = (G3_X + 1) * (G3_Y + 1) * (G3_Z + 1);
__global__ __launch_bounds__(256) void k_prism_g3_tensor(int nx, int ny, int nz, const double *__restrict__ xe,
                                                         const double *__restrict__ ye, const double *__restrict__ ze, int nobs,
                                                         const double *__restrict__ xd, const double *__restrict__ yd,
                                                         const double *__restrict__ zd, const double *__restrict__ cw,
                                                         double *__restrict__ rows, int *__restrict__ err, double *__restrict__ sumsq)
{
    __shared__ double T[3][G3_NODES];
    __shared__ double s_w[4];
    init_math_tables();                                   // (published by the first barrier of the observation loop)
    __shared__ double s_xe[G3_X + 1], s_ye[G3_Y + 1], s_ze[G3_Z + 1];
    __shared__ int s_node[G3_NODES];                      // LDS slot | a << 12 | b << 18 | c << 22
    const int tiles_x = (nx + G3_X - 1) / G3_X, tiles_y = (ny + G3_Y - 1) / G3_Y;
    const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y, bz = blockIdx.x / (tiles_x * tiles_y);
    const int i0 = bx * G3_X, j0 = by * G3_Y, k0 = bz * G3_Z;
    const int cx = min(G3_X, nx - i0), cy = min(G3_Y, ny - j0), cz = min(G3_Z, nz - k0);
    const int64_t N = (int64_t)nx * ny * nz;
    const int nnode = (cx + 1) * (cy + 1) * (cz + 1);
    int bad = 0;
    for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
        const int a = n % (cx + 1), b = (n / (cx + 1)) % (cy + 1), c = n / ((cx + 1) * (cy + 1));
        s_node[n] = ((c * (G3_Y + 1) + b) * (G3_X + 1) + a) | (a << 12) | (b << 18) | (c << 22);
    }
    if (threadIdx.x <= cx) s_xe[threadIdx.x] = xe[i0 + threadIdx.x];
    if (threadIdx.x <= cy) s_ye[threadIdx.x] = ye[j0 + threadIdx.x];
    if (threadIdx.x <= cz) s_ze[threadIdx.x] = ze[k0 + threadIdx.x];
    // one thread per (x, y) column of the tile, one cell per z layer (as in k_prism_gz_tensor)
    static_assert(G3_X * G3_Y == 256, "one thread per (x, y) column of the tile");
    constexpr int LAYER = (G3_Y + 1) * (G3_X + 1);
    const int ta = threadIdx.x % G3_X, tb = threadIdx.x / G3_X;
    const bool col_ok = ta < cx && tb < cy;
    const int slot0 = tb * (G3_X + 1) + ta;
    const int64_t col0 = ((int64_t)k0 * ny + (j0 + tb)) * nx + (i0 + ta), lay = (int64_t)ny * nx;
    for (int o = 0; o < nobs; ++o) {
        const double xo = xd[o], yo = yd[o], zo = zd[o];
        lds_barrier();
        for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
            const int code = s_node[n];
            double tx, ty, tz;
            corner_term3(xo - s_xe[(code >> 12) & 63], yo - s_ye[(code >> 18) & 15], zo - s_ze[code >> 22], tx, ty, tz, bad);
            const int id = code & 4095;
            T[0][id] = tx;
            T[1][id] = ty;
            T[2][id] = tz;
        }
        lds_barrier();
        double sq[3] = {0.0, 0.0, 0.0};
        if (col_ok) {
#pragma unroll
            for (int j = 0; j < G3_Z; ++j) {
                if (j >= cz) break;
                const int64_t col = col0 + j * lay;
                const double w = cw ? cw[col] : 1.0;
#pragma unroll
                for (int comp = 0; comp < 3; ++comp) {
                    const double *t0 = &T[comp][slot0 + j * LAYER];
                    double gsum = 0.0;
#pragma unroll
                    for (int K = 0; K < 2; ++K)
#pragma unroll
                        for (int L = 0; L < 2; ++L)
#pragma unroll
                            for (int M = 0; M < 2; ++M) {
                                const double dmu = ((K + L + M) & 1) ? 1.0 : -1.0;
                                gsum = gsum + dmu * t0[(M * (G3_Y + 1) + L) * (G3_X + 1) + K];
                            }
                    double v = g_grav() * gsum;                                                     // :118-120
                    if (cw) v = v * w;
                    __builtin_nontemporal_store(v, &rows[(int64_t)(o * 3 + comp) * N + col]);
                    sq[comp] = fma(v, v, sq[comp]);
                }
            }
        }
        if (sumsq) {                                        // cost_full (sensitivity_gravmag.F90:234), fixed order
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double t = sq[i];
#pragma unroll
                for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_down(t, dd);
                lds_barrier();
                if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
                lds_barrier();
                if (threadIdx.x == 0) sumsq[(int64_t)(o * 3 + i) * gridDim.x + blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
            }
        }
    }
    if (bad) atomicOr(err, bad);
}

// =============================================================================================================
// magnetic rows: magprism + sharmbox (src/forward/gravmag/mag/magnetic_field.f90:118-297, :321-457),
// scalar susceptibility model, TMI data
// =============================================================================================================
struct MagField { double magv[3]; double intensity; };

// magnetic tensor of one box; bad |= 4 / 8 when the X / Y grid boundary coincides with the data position (:345-354)
__device__ __forceinline__ void sharmbox_dev(double x0, double y0, double z0, double x1, double y1, double z1, double x2,
                                             double y2, double z2, double *tx, double *ty, double *tz, int &bad)
{
    const double eps = 0.;
    const double rx1 = x1 - x0 + eps, rx2 = x2 - x0 + eps;                 // :336-341
    const double ry1 = y1 - y0 + eps, ry2 = y2 - y0 + eps;
    const double rz1 = z1 - z0 + eps, rz2 = z2 - z0 + eps;
    if (rx1 == 0. || rx2 == 0.) bad |= 4;
    if (ry1 == 0. || ry2 == 0.) bad |= 8;
    const double rx1sq = rx1 * rx1, rx2sq = rx2 * rx2, ry1sq = ry1 * ry1, ry2sq = ry2 * ry2, rz1sq = rz1 * rz1, rz2sq = rz2 * rz2;
    double R1 = ry2sq + rx2sq, R2 = ry2sq + rx1sq, R3 = ry1sq + rx2sq, R4 = ry1sq + rx1sq;        // :361-364
    double a1 = sqrt(rz2sq + R2), a2 = sqrt(rz2sq + R1), a3 = sqrt(rz1sq + R1), a4 = sqrt(rz1sq + R2);
    double a5 = sqrt(rz2sq + R3), a6 = sqrt(rz2sq + R4), a7 = sqrt(rz1sq + R4), a8 = sqrt(rz1sq + R3);
    tx[0] = datan2(ry1 * rz2, (rx2 * a5 + eps)) - datan2(ry2 * rz2, (rx2 * a2 + eps)) + datan2(ry2 * rz1, (rx2 * a3 + eps)) -
            datan2(ry1 * rz1, (rx2 * a8 + eps)) + datan2(ry2 * rz2, (rx1 * a1 + eps)) - datan2(ry1 * rz2, (rx1 * a6 + eps)) +
            datan2(ry1 * rz1, (rx1 * a7 + eps)) - datan2(ry2 * rz1, (rx1 * a4 + eps));                     // :376-383
    ty[0] = dlog((rz2 + a2 + eps) / (rz1 + a3 + eps)) - dlog((rz2 + a1 + eps) / (rz1 + a4 + eps)) +
            dlog((rz2 + a6 + eps) / (rz1 + a7 + eps)) - dlog((rz2 + a5 + eps) / (rz1 + a8 + eps));        // :386-389
    ty[1] = datan2(rx1 * rz2, (ry2 * a1 + eps)) - datan2(rx2 * rz2, (ry2 * a2 + eps)) + datan2(rx2 * rz1, (ry2 * a3 + eps)) -
            datan2(rx1 * rz1, (ry2 * a4 + eps)) + datan2(rx2 * rz2, (ry1 * a5 + eps)) - datan2(rx1 * rz2, (ry1 * a6 + eps)) +
            datan2(rx1 * rz1, (ry1 * a7 + eps)) - datan2(rx2 * rz1, (ry1 * a8 + eps));                     // :392-399
    R1 = ry2sq + rz1sq; R2 = ry2sq + rz2sq; R3 = ry1sq + rz1sq; R4 = ry1sq + rz2sq;                    // :404-407
    a1 = sqrt(rx1sq + R1); a2 = sqrt(rx2sq + R1); a3 = sqrt(rx1sq + R2); a4 = sqrt(rx2sq + R2);
    a5 = sqrt(rx1sq + R3); a6 = sqrt(rx2sq + R3); a7 = sqrt(rx1sq + R4); a8 = sqrt(rx2sq + R4);
    ty[2] = dlog((rx1 + a1 + eps) / (rx2 + a2 + eps)) - dlog((rx1 + a3 + eps) / (rx2 + a4 + eps)) +
            dlog((rx1 + a7 + eps) / (rx2 + a8 + eps)) - dlog((rx1 + a5 + eps) / (rx2 + a6 + eps));        // :419-422
    R1 = rx2sq + rz1sq; R2 = rx2sq + rz2sq; R3 = rx1sq + rz1sq; R4 = rx1sq + rz2sq;                    // :424-427
    a1 = sqrt(ry1sq + R1); a2 = sqrt(ry2sq + R1); a3 = sqrt(ry1sq + R2); a4 = sqrt(ry2sq + R2);
    a5 = sqrt(ry1sq + R3); a6 = sqrt(ry2sq + R3); a7 = sqrt(ry1sq + R4); a8 = sqrt(ry2sq + R4);
    tx[2] = dlog((ry1 + a1 + eps) / (ry2 + a2 + eps)) - dlog((ry1 + a3 + eps) / (ry2 + a4 + eps)) +
            dlog((ry1 + a7 + eps) / (ry2 + a8 + eps)) - dlog((ry1 + a5 + eps) / (ry2 + a6 + eps));        // :439-442
    tz[2] = -1 * (tx[0] + ty[1]);                                                                       // :446
    tz[1] = ty[2];
    tx[1] = ty[0];
    tz[0] = tx[2];
}

// magnetic tensor of a cell seen from (xo, yo, zo), incl. the 6-sub-box split for an observation strictly inside the cell
// (magnetic_field.f90:139-240)
__device__ __forceinline__ void mag_cell_tensor(double x1, double x2, double y1, double y2, double z1, double z2, double xo, double yo,
                                                double zo, double *tx, double *ty, double *tz, int &bad)
{
    if (x1 < xo && x2 > xo && y1 < yo && y2 > yo && z1 < zo && z2 > zo) {                          // :139-141
        double width = (double)0.1f;                                                               // :144
        const double min_clr = fmin(fmin(fmin(fabs(xo - x1), fabs(xo - x2)), fmin(fabs(yo - y1), fabs(yo - y2))),
                                    fmin(fabs(zo - z1), fabs(zo - z2)));
        if (width > min_clr) width = 0.5 * min_clr;                                                // :153
        tx[0] = tx[1] = tx[2] = ty[0] = ty[1] = ty[2] = tz[0] = tz[1] = tz[2] = 0.0;
        for (int j = 0; j < 6; ++j) {                                                              // :157-226
            double bx1 = x1, bx2 = x2, by1 = y1, by2 = y2, bz1 = zo - width, bz2 = zo + width;
            if (j == 0) { bz1 = z1; bz2 = zo - width; }
            else if (j == 1) { bz1 = zo + width; bz2 = z2; }
            else if (j == 2) { bx2 = xo - width; }
            else if (j == 3) { bx1 = xo + width; }
            else if (j == 4) { bx1 = xo - width; bx2 = xo + width; by2 = yo - width; }
            else { bx1 = xo - width; bx2 = xo + width; by1 = yo + width; }
            double sx[3], sy[3], sz[3];
            sharmbox_dev(xo, yo, zo, bx1, by1, bz1, bx2, by2, bz2, sx, sy, sz, bad);
            for (int k = 0; k < 3; ++k) { tx[k] = tx[k] + sx[k]; ty[k] = ty[k] + sy[k]; tz[k] = tz[k] + sz[k]; }
        }
    } else {
        sharmbox_dev(xo, yo, zo, x1, y1, z1, x2, y2, z2, tx, ty, tz, bad);                         // :230-240
    }
}

// projection of the tensor on the field direction / components and the unit scaling (magnetic_field.f90:243-295):
// out[d][k] = sensit_line(:, k, d) of this cell
template <int NCM, int NCD>
__device__ __forceinline__ void mag_project(const double *tx, const double *ty, const double *tz, const MagField &mf, double out[NCD][NCM])
{
    const double PI = 3.14159265358979323846;
    const double mu0 = 4.0 * PI * 1.e-7, T2nT = 1.e+9;                                                     // :31-34
    if (NCM == 1) {
        const double mx = (tx[0] * mf.magv[0] + tx[1] * mf.magv[1]) + tx[2] * mf.magv[2];                  // :246-248
        const double my = (ty[0] * mf.magv[0] + ty[1] * mf.magv[1]) + ty[2] * mf.magv[2];
        const double mz = (tz[0] * mf.magv[0] + tz[1] * mf.magv[1]) + tz[2] * mf.magv[2];
        if (NCD == 1) out[0][0] = mx * mf.magv[0] + my * mf.magv[1] + mz * mf.magv[2];                     // :251
        else { out[0][0] = mx; out[NCD > 1 ? 1 : 0][0] = my; out[NCD > 2 ? 2 : 0][0] = mz; }               // :254-256
    } else {
#pragma unroll
        for (int k = 0; k < NCM; ++k) {
            if (NCD == 1) out[0][k] = tx[k] * mf.magv[0] + ty[k] * mf.magv[1] + tz[k] * mf.magv[2];        // :268
            else { out[0][k] = tx[k]; out[NCD > 1 ? 1 : 0][k] = ty[k]; out[NCD > 2 ? 2 : 0][k] = tz[k]; } // :273-275
        }
    }
#pragma unroll
    for (int d = 0; d < NCD; ++d)
#pragma unroll
        for (int k = 0; k < NCM; ++k) {
            double v = out[d][k];
            v = (NCM == 1) ? mf.intensity * v : (mu0 * T2nT) * v;                                          // :286-291
            out[d][k] = v / (4.0 * PI);                                                                    // :295
        }
}

// NCM model components (1 susceptibility | 3 magnetisation vector), NCD data components (1 TMI | 3): sub-row
// (o*NCD + d)*NCM + k of the output holds sensit_line(:, k, d) of observation o (magnetic_field.f90:243-295).
template <int NCM, int NCD>
__global__ __launch_bounds__(256) void k_magprism(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                                                  const double *__restrict__ Y1, const double *__restrict__ Y2,
                                                  const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                  int nobs, const double *__restrict__ xd, const double *__restrict__ yd,
                                                  const double *__restrict__ zd, const double *__restrict__ cw,
                                                  MagField mf, double *__restrict__ rows, int *__restrict__ err,
                                                  double *__restrict__ sumsq)
{
    constexpr int NSUB = NCM * NCD;
    __shared__ double s_sq[PRISM_MAX_BATCH];
    init_math_tables();
    if (sumsq)
        for (int o = threadIdx.x; o < nobs * NSUB; o += blockDim.x) s_sq[o] = 0.0;
    __syncthreads();
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x; p0 < N; p0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = p0 + threadIdx.x;
        const bool active = p < N;
        const int64_t pc = active ? p : N - 1;
        const double x1 = X1[pc], x2 = X2[pc], y1 = Y1[pc], y2 = Y2[pc], z1 = Z1[pc], z2 = Z2[pc];
        const double w = cw ? cw[pc] : 1.0;
        for (int o = 0; o < nobs; ++o) {
            const double xo = xd[o], yo = yd[o], zo = zd[o];
            double tx[3], ty[3], tz[3];
            int bad = 0;
            mag_cell_tensor(x1, x2, y1, y2, z1, z2, xo, yo, zo, tx, ty, tz, bad);
            double out[NCD][NCM];
            mag_project<NCM, NCD>(tx, ty, tz, mf, out);
            if (bad && active) atomicOr(err, bad);
#pragma unroll
            for (int d = 0; d < NCD; ++d)
#pragma unroll
                for (int k = 0; k < NCM; ++k) {
                    double v = out[d][k];
                    if (cw) v = v * w;
                    const int sub = (o * NCD + d) * NCM + k;
                    if (active) rows[(int64_t)sub * N + p] = v;
                    if (sumsq) {
                        double sq = active ? v * v : 0.0;
#pragma unroll
                        for (int dd = 32; dd > 0; dd >>= 1) sq += __shfl_down(sq, dd);
                        if ((threadIdx.x & 63) == 0) atomicAdd(&s_sq[sub], sq);
                    }
                }
        }
    }
    if (sumsq) {
        __syncthreads();
        for (int o = threadIdx.x; o < nobs * NSUB; o += blockDim.x) sumsq[(int64_t)o * gridDim.x + blockIdx.x] = s_sq[o];
    }
}

// Tensor-product grid: of the 24 atan2 + 12 log + 24 sqrt of a cell's tensor, the two atan2 families and the three distances
// (one per summation order of the squares, so that the bits are sharmbox's) depend on one NODE only.  A workgroup evaluates them
// once for the (MT_X+1)(MT_Y+1)(MT_Z+1) nodes of its tile into LDS (1.4 nodes per cell: 2.8 atan2 + 4.2 sqrt per cell instead of
// 24 + 24); the 12 logs of corner-pair ratios belong to the EDGES of the node lattice (3.7 per cell, evaluated in a second phase).
// Same operations in the same order as sharmbox_dev -> same bits as k_magprism.  The one cell per observation that may CONTAIN
// the observation (6 sub-boxes, a different and register-hungry code path) is left at zero here and written by
// k_magprism_inside_fix afterwards.
// Tile: 16 x 7 x 6 cells = 17 x 8 x 7 = 952 nodes - FOUR passes of the 256 threads over the nodes, 93 % full (the 16 x 8 x 6 tile of rounds
// 2-4 had 1071 nodes: five passes, the fifth 16 % full) in 38 KB of node arrays: three workgroups per CU.  The x extent stays 16 cells:
// a thread row stores 128 contiguous, 128-byte-aligned bytes of an output row (a 15-cell tile, 1008 nodes in four FULL passes, was
// measured: the magnetisation-vector kernel with nine output rows per observation ran at half speed on its 120-byte segments).
// What a node keeps for the edge phase is the corner quantity the logs are taken of, r + a (sharmbox's rz2 + a2, rx1 + a1, ...: the
// sum is the node's own), so an edge is one division and one log of two LDS values.  The edge value replaces the node's in place; the
// passes walk the nodes in ascending slot order and an edge only reads slots >= its own, so ONE barrier per pass is enough (a pass
// writes its slots after the barrier that ended its reads; later passes never read them) and three doubles are live per thread
// (rounds 2-4 held all five passes' fifteen values over one barrier: 168 VGPRs and 4-13 spilled).
constexpr int MT_X = 16, MT_Y = 7, MT_Z = 6;
constexpr int MT_TX = 16, MT_TY = 8;                       // thread columns / rows of the cell phase (one row idle when the tile is full)
constexpr int MT_NODES = (MT_X + 1) * (MT_Y + 1) * (MT_Z + 1);
template <int NCM, int NCD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_magprism_tensor(int nx, int ny, int nz, const double *__restrict__ xe,
                                                         const double *__restrict__ ye, const double *__restrict__ ze, int nobs,
                                                         const double *__restrict__ xd, const double *__restrict__ yd,
                                                         const double *__restrict__ zd, const double *__restrict__ cw, MagField mf,
                                                         double *__restrict__ rows, int *__restrict__ err, double *__restrict__ sumsq)
{
    constexpr int NSUB = NCM * NCD;
    // Tz / Tx / Ty: r + a of the node (node phase), then the log of the edge that starts at the node (edge phase)
    __shared__ double Tz[MT_NODES], Tx[MT_NODES], Ty[MT_NODES], TAx[MT_NODES], TAy[MT_NODES];      // 5 x 952 doubles = 38 KB
    __shared__ double s_w[4];
    init_math_tables();                                   // (+ 4 KB; the first __syncthreads() of the observation loop publishes them)
    const double eps = 0.;
    const int tiles_x = (nx + MT_X - 1) / MT_X, tiles_y = (ny + MT_Y - 1) / MT_Y;
    const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y, bz = blockIdx.x / (tiles_x * tiles_y);
    const int i0 = bx * MT_X, j0 = by * MT_Y, k0 = bz * MT_Z;
    const int cx = min(MT_X, nx - i0), cy = min(MT_Y, ny - j0), cz = min(MT_Z, nz - k0);
    const int64_t N = (int64_t)nx * ny * nz;
    const int nnode = (cx + 1) * (cy + 1) * (cz + 1);
    int bad = 0;
#define NODE(a, b, c) (((c) * (MT_Y + 1) + (b)) * (MT_X + 1) + (a))
    // observation-independent index arithmetic, once per workgroup: the node codes (LDS slot | a << 12 | b << 18 | c << 22) in LDS,
    // ascending in the slot; the cells of a thread are the column (ta, tb) of the tile in the layers tc0, tc0 + 2, ...
    __shared__ double s_xe[MT_X + 1], s_ye[MT_Y + 1], s_ze[MT_Z + 1];
    constexpr int NPT = (MT_NODES + 255) / 256, DY = MT_X + 1, DZ = (MT_Y + 1) * (MT_X + 1);
    static_assert(MT_NODES <= 4096 && MT_X + 1 <= 64 && MT_Y + 1 <= 16, "node code fields");
    __shared__ int s_node[MT_NODES];
    for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
        const int a = n % (cx + 1), b = (n / (cx + 1)) % (cy + 1), c = n / ((cx + 1) * (cy + 1));
        s_node[n] = NODE(a, b, c) | (a << 12) | (b << 18) | (c << 22);
    }
    if (threadIdx.x <= cx) s_xe[threadIdx.x] = xe[i0 + threadIdx.x];
    if (threadIdx.x <= cy) s_ye[threadIdx.x] = ye[j0 + threadIdx.x];
    if (threadIdx.x <= cz) s_ze[threadIdx.x] = ze[k0 + threadIdx.x];
    static_assert(MT_TX * MT_TY * 2 == 256 && MT_TX >= MT_X && MT_TY >= MT_Y && MT_Z % 2 == 0, "two (x, y) planes of threads, each the lower / upper half of the tile's layers");
    constexpr int CPT = MT_Z / 2;
    const int ta = threadIdx.x % MT_TX, tb = (threadIdx.x / MT_TX) % MT_TY, tc0 = threadIdx.x / (MT_TX * MT_TY);
    const bool col_ok = ta < cx && tb < cy;
    for (int o = 0; o < nobs; ++o) {
        const double xo = xd[o], yo = yd[o], zo = zd[o];
        lds_barrier();
        // node phase
        for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
            const int code = s_node[n];
            const double rx = s_xe[(code >> 12) & 63] - xo + eps, ry = s_ye[(code >> 18) & 15] - yo + eps, rz = s_ze[code >> 22] - zo + eps;       // :336-341
            const double rxsq = rx * rx, rysq = ry * ry, rzsq = rz * rz;
            const double az = sqrt(rzsq + (rysq + rxsq));        // :361-372   a = sqrt(rz^2 + R), R = ry^2 + rx^2
            const double ax = sqrt(rxsq + (rysq + rzsq));        // :404-415   a = sqrt(rx^2 + R), R = ry^2 + rz^2
            const double ay = sqrt(rysq + (rxsq + rzsq));        // :424-435   a = sqrt(ry^2 + R), R = rx^2 + rz^2
            const int id = code & 4095;
            Tz[id] = rz + az + eps;                               // the node's operand of ty(1)'s logs, :386-389
            Tx[id] = rx + ax + eps;                               //                      ty(3)'s,      :419-422
            Ty[id] = ry + ay + eps;                               //                      tx(3)'s,      :439-442
            TAx[id] = datan2(ry * rz, (rx * az + eps));           // the terms of tx(1), :376-383
            TAy[id] = datan2(rx * rz, (ry * az + eps));           // the terms of ty(2), :392-399
        }
        lds_barrier();
        // edge phase: the 12 logs of a cell's tensor are logs of corner-pair ratios along one axis - each belongs to an EDGE of the
        // node lattice and is shared by the (up to) 4 cells around that edge: once per edge, 3.7 logs + divisions per cell instead of 12
#pragma unroll 1
        for (int k = 0; k < NPT; ++k) {
            const int n = threadIdx.x + 256 * k;
            const int code = n < nnode ? s_node[n] : -1;
            const int id = code & 4095;
            double ez = 0.0, ex = 0.0, ey = 0.0;
            if (code >= 0) {
                const int a = (code >> 12) & 63, b = (code >> 18) & 15, c = code >> 22;
                if (c < cz) ez = dlog(Tz[id + DZ] / Tz[id]);              // ty(1): (rz2 + a(k = 2)) / (rz1 + a(k = 1)), :386-389
                if (a < cx) ex = dlog(Tx[id] / Tx[id + 1]);               // ty(3): (rx1 + a(i = 1)) / (rx2 + a(i = 2)), :419-422
                if (b < cy) ey = dlog(Ty[id] / Ty[id + DY]);              // tx(3): (ry1 + a(j = 1)) / (ry2 + a(j = 2)), :439-442
            }
            lds_barrier();                    // every read of this pass is done; the passes to come read higher slots only
            if (code >= 0) { Tz[id] = ez; Tx[id] = ex; Ty[id] = ey; }
        }
        lds_barrier();
        const double *cwo = cw;
        asm volatile("" : "+s"(cwo));                 // the weights are re-read per observation, not held over the node / edge phases
        // cell phase.  Nine outputs per cell (magnetisation vector x three data components) go in THREE passes, one data component =
        // one row of the tensor each: a pass reads the 16-24 LDS operands its row needs and keeps three sums of squares, where one pass
        // over all nine kept 32 operands x 3 unrolled cells + 9 values + 9 sums live and spilled 13 registers.
        constexpr int NG = NSUB == 9 ? 3 : 1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
        double sq[NSUB];
#pragma unroll
        for (int i = 0; i < NSUB; ++i) sq[i] = 0.0;
        if (col_ok) {
            const int a = ta, b = tb;
            const double x1 = s_xe[a], x2 = s_xe[a + 1], y1 = s_ye[b], y2 = s_ye[b + 1];
            const double rx1 = x1 - xo + eps, rx2 = x2 - xo + eps, ry1 = y1 - yo + eps, ry2 = y2 - yo + eps;
            // The thread's cells are ADJACENT layers of one (x, y) column: the node and edge values of a cell's upper face are the next
            // cell's lower face and stay in registers - 12 + 16 LDS operands per cell instead of 32 (the cell phase was 27 % of the kernel).
            // corner (i, j, k), i / j / k in {1, 2}: node (a + i - 1, b + j - 1, c + k - 1); an edge sits at its lower node
#define C_(T, i, j, k) T[NODE(a + (i) - 1, b + (j) - 1, c + (k) - 1)]
            double fx[2][2], fy[2][2], gx[2], gy[2];           // lower face: TAx / TAy at (i, j), the x edges at j, the y edges at i
            {
                const int c = tc0 * CPT;
                if (c < cz) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) { fx[i][jj] = C_(TAx, i + 1, jj + 1, 1); fy[i][jj] = C_(TAy, i + 1, jj + 1, 1); }
                    gx[0] = C_(Tx, 1, 1, 1); gx[1] = C_(Tx, 1, 2, 1);
                    gy[0] = C_(Ty, 1, 1, 1); gy[1] = C_(Ty, 2, 1, 1);
                }
            }
#pragma unroll 1
            for (int j = 0; j < CPT; ++j) {
                const int c = tc0 * CPT + j;
                if (c >= cz) break;
                const double z1 = s_ze[c], z2 = s_ze[c + 1];
                double tx[3], ty[3], tz[3];
                const bool inside = x1 < xo && x2 > xo && y1 < yo && y2 > yo && z1 < zo && z2 > zo;       // :139-141: k_magprism_inside_fix writes this cell
                if (!inside && g == 0) {
                    if (rx1 == 0. || rx2 == 0.) bad |= 4;
                    if (ry1 == 0. || ry2 == 0.) bad |= 8;
                }
                double ux[2][2], uy[2][2], hx[2], hy[2];       // upper face
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) { ux[i][jj] = C_(TAx, i + 1, jj + 1, 2); uy[i][jj] = C_(TAy, i + 1, jj + 1, 2); }
                hx[0] = C_(Tx, 1, 1, 2); hx[1] = C_(Tx, 1, 2, 2);
                hy[0] = C_(Ty, 1, 1, 2); hy[1] = C_(Ty, 2, 1, 2);
                // (index [i - 1][j - 1]; f = k 1, u = k 2)
                tx[0] = ux[1][0] - ux[1][1] + fx[1][1] - fx[1][0] + ux[0][1] - ux[0][0] + fx[0][0] - fx[0][1];                  // :376-383
                ty[0] = C_(Tz, 2, 2, 1) - C_(Tz, 1, 2, 1) + C_(Tz, 1, 1, 1) - C_(Tz, 2, 1, 1);                                    // z edges, :386-389
                ty[1] = uy[0][1] - uy[1][1] + fy[1][1] - fy[0][1] + uy[1][0] - uy[0][0] + fy[0][0] - fy[1][0];                  // :392-399
                ty[2] = gx[1] - hx[1] + hx[0] - gx[0];                                                                           // x edges, :419-422
                tx[2] = gy[1] - hy[1] + hy[0] - gy[0];                                                                           // y edges, :439-442
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    gx[i] = hx[i]; gy[i] = hy[i];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) { fx[i][jj] = ux[i][jj]; fy[i][jj] = uy[i][jj]; }
                }
#undef C_
                tz[2] = -1 * (tx[0] + ty[1]);                                                                                 // :446
                tz[1] = ty[2];
                tx[1] = ty[0];
                tz[0] = tx[2];
                double out[NCD][NCM];
                mag_project<NCM, NCD>(tx, ty, tz, mf, out);
                const int64_t p = ((int64_t)(k0 + c) * ny + (j0 + b)) * nx + (i0 + a);
                const double w = cw ? cwo[p] : 1.0;
#pragma unroll
                for (int d = 0; d < NCD; ++d)
#pragma unroll
                    for (int k = 0; k < NCM; ++k) {
                        if (NG > 1 && d != g) continue;      // (this pass's tensor row; what the other rows need is dead code here)
                        double v = out[d][k];
                        if (cw) v = v * w;
                        if (inside) v = 0.0;
                        const int sub = (o * NCD + d) * NCM + k;
                        __builtin_nontemporal_store(v, &rows[(int64_t)sub * N + p]);
                        sq[d * NCM + k] = fma(v, v, sq[d * NCM + k]);
                    }
            }
        }
        if (sumsq) {                                        // cost_full (sensitivity_gravmag.F90:234), fixed order
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                if (NG > 1 && i / NCM != g) continue;
                double t = sq[i];
#pragma unroll
                for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_down(t, dd);
                lds_barrier();
                if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
                lds_barrier();
                if (threadIdx.x == 0) sumsq[(int64_t)(o * NSUB + i) * gridDim.x + blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
            }
        }
        }
    }
#undef NODE
    if (bad) atomicOr(err, bad);
}

// The cell that contains an observation (strictly: magnetic_field.f90:139-141), if any: the 6-sub-box tensor of mag_cell_tensor,
// written over the zero k_magprism_tensor left there; its squares join the tile's cost_full partial.  One workgroup per
// observation; the threads look for the cell along each axis (at most one per axis on a tensor grid), thread 0 does the rest.
template <int NCM, int NCD>
__global__ __launch_bounds__(64) void k_magprism_inside_fix(int nx, int ny, int nz, const double *__restrict__ xe,
                                                            const double *__restrict__ ye, const double *__restrict__ ze,
                                                            const double *__restrict__ xd, const double *__restrict__ yd,
                                                            const double *__restrict__ zd, const double *__restrict__ cw, MagField mf,
                                                            double *__restrict__ rows, int *__restrict__ err, double *__restrict__ sumsq,
                                                            int ntiles)
{
    constexpr int NSUB = NCM * NCD;
    __shared__ int s_cell[3];
    init_math_tables();
    const int o = blockIdx.x;
    const double xo = xd[o], yo = yd[o], zo = zd[o];
    if (threadIdx.x < 3) s_cell[threadIdx.x] = -1;
    __syncthreads();
    for (int a = threadIdx.x; a < nx; a += blockDim.x) if (xe[a] < xo && xe[a + 1] > xo) s_cell[0] = a;
    for (int b = threadIdx.x; b < ny; b += blockDim.x) if (ye[b] < yo && ye[b + 1] > yo) s_cell[1] = b;
    for (int c = threadIdx.x; c < nz; c += blockDim.x) if (ze[c] < zo && ze[c + 1] > zo) s_cell[2] = c;
    __syncthreads();
    const int a = s_cell[0], b = s_cell[1], c = s_cell[2];
    if (threadIdx.x != 0 || a < 0 || b < 0 || c < 0) return;
    int bad = 0;
    double tx[3], ty[3], tz[3];
    mag_cell_tensor(xe[a], xe[a + 1], ye[b], ye[b + 1], ze[c], ze[c + 1], xo, yo, zo, tx, ty, tz, bad);
    double out[NCD][NCM];
    mag_project<NCM, NCD>(tx, ty, tz, mf, out);
    const int64_t N = (int64_t)nx * ny * nz, p = ((int64_t)c * ny + b) * nx + a;
    const double w = cw ? cw[p] : 1.0;
    const int tiles_x = (nx + MT_X - 1) / MT_X, tiles_y = (ny + MT_Y - 1) / MT_Y;
    const int tile = ((c / MT_Z) * tiles_y + b / MT_Y) * tiles_x + a / MT_X;
#pragma unroll
    for (int d = 0; d < NCD; ++d)
#pragma unroll
        for (int k = 0; k < NCM; ++k) {
            double v = out[d][k];
            if (cw) v = v * w;
            const int sub = (o * NCD + d) * NCM + k;
            rows[(int64_t)sub * N + p] = v;
            if (sumsq) sumsq[(int64_t)(o * NSUB + d * NCM + k) * ntiles + tile] += v * v;
        }
    if (bad) atomicOr(err, bad);
}

// =============================================================================================================
// gravity gradiometry rows: gradiprism_zz / gradiprism_full (src/forward/gravmag/grav/gravity_field.f90:315-362, :207-310)
// FULL: six sub-rows per observation in the order the build stores them (sensitivity_gravmag.F90:210-212):
// XX, YY, ZZ, XY, YZ, ZX.  bad |= 16 zero denominator (:275-277), 32 bad log argument (:282-284).
// =============================================================================================================
template <bool FULL>
__global__ __launch_bounds__(256) void k_gradiprism(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                                                    const double *__restrict__ Y1, const double *__restrict__ Y2,
                                                    const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                    int nobs, const double *__restrict__ xd, const double *__restrict__ yd,
                                                    const double *__restrict__ zd, const double *__restrict__ cw,
                                                    double *__restrict__ rows, int *__restrict__ err,
                                                    double *__restrict__ sumsq)
{
    constexpr int NC = FULL ? 6 : 1;
    const double twopi = 2.0 * 3.14159265358979323846;
    __shared__ double s_sq[PRISM_MAX_BATCH];
    init_math_tables();
    if (sumsq)
        for (int o = threadIdx.x; o < nobs * NC; o += blockDim.x) s_sq[o] = 0.0;
    __syncthreads();
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x; p0 < N; p0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = p0 + threadIdx.x;
        const bool active = p < N;
        const int64_t pc = active ? p : N - 1;
        const double x1 = X1[pc], x2 = X2[pc], y1 = Y1[pc], y2 = Y2[pc], z1 = Z1[pc], z2 = Z2[pc];
        const double w = cw ? cw[pc] : 1.0;
        for (int o = 0; o < nobs; ++o) {
            double XX[2], YY[2], ZZ[2];
            XX[0] = xd[o] - x1; XX[1] = xd[o] - x2;                                                        // :232-237
            YY[0] = yd[o] - y1; YY[1] = yd[o] - y2;
            ZZ[0] = -(zd[o] - z1); ZZ[1] = -(zd[o] - z2);
            double gxx = 0.0, gyy = 0.0, gzz = 0.0, gxy = 0.0, gyz = 0.0, gzx = 0.0;
            int bad = 0;
#pragma unroll
            for (int K = 0; K < 2; ++K)
#pragma unroll
                for (int L = 0; L < 2; ++L)
#pragma unroll
                    for (int M = 0; M < 2; ++M) {
                        const double dmu = ((K + L + M) & 1) ? 1.0 : -1.0;
                        const double Rs = sqrt(XX[K] * XX[K] + YY[L] * YY[L] + ZZ[M] * ZZ[M]);            // :251
                        double vzz = -datan2(XX[K] * YY[L], Rs * ZZ[M]);                                    // :255
                        if (vzz < 0) vzz = vzz + twopi;
                        gzz = gzz + dmu * vzz;
                        if (FULL) {
                            double vxx = datan2(XX[K] * YY[L], XX[K] * XX[K] + Rs * ZZ[M] + ZZ[M] * ZZ[M]);   // :253
                            double vyy = datan2(XX[K] * YY[L], Rs * Rs + Rs * ZZ[M] - XX[K] * XX[K]);         // :254
                            if (vxx < 0) vxx = vxx + twopi;
                            if (vyy < 0) vyy = vyy + twopi;
                            const double arg1 = Rs + ZZ[M];
                            const double arg21 = Rs - YY[L], arg22 = Rs + YY[L];
                            const double arg31 = Rs - XX[K], arg32 = Rs + XX[K];
                            if (arg22 == 0. || arg32 == 0.) bad |= 16;
                            const double arg2 = arg21 / arg22, arg3 = arg31 / arg32;
                            if (arg1 <= 0. || arg2 <= 0. || arg3 <= 0.) bad |= 32;
                            const double vxy = dlog(arg1);
                            const double vzx = 0.5 * dlog(arg2);
                            const double vyz = 0.5 * dlog(arg3);
                            gxx = gxx + dmu * vxx;
                            gyy = gyy + dmu * vyy;
                            gxy = gxy + dmu * vxy;
                            gyz = gyz + dmu * vyz;
                            gzx = gzx + dmu * vzx;
                        }
                    }
            if (bad && active) atomicOr(err, bad);
            const double g6[6] = {gxx, gyy, gzz, gxy, gyz, gzx};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double v = g_grav() * (FULL ? g6[c] : gzz);                                                // :301-306, :358
                if (cw) v = v * w;
                const int sub = o * NC + c;
                if (active) rows[(int64_t)sub * N + p] = v;
                if (sumsq) {
                    double sq = active ? v * v : 0.0;
#pragma unroll
                    for (int dd = 32; dd > 0; dd >>= 1) sq += __shfl_down(sq, dd);
                    if ((threadIdx.x & 63) == 0) atomicAdd(&s_sq[sub], sq);
                }
            }
        }
    }
    if (sumsq) {
        __syncthreads();
        for (int o = threadIdx.x; o < nobs * NC; o += blockDim.x) sumsq[(int64_t)o * gridDim.x + blockIdx.x] = s_sq[o];
    }
}

// Tensor-product grid: every term of the gradient tensor depends on ONE corner, so a workgroup evaluates the terms once per
// node of its tile into LDS (1.3 - 1.5 nodes per cell instead of 8 corner evaluations) and each cell sums its 8 nodes in the
// reference's order (K outer, L, M inner) -> same bits as k_gradiprism.
template <bool FULL>
struct GradTile {
    static constexpr int X = FULL ? 16 : 32, Y = 8, Z = FULL ? 4 : 8, NV = FULL ? 6 : 1;
    static constexpr int NODES = (X + 1) * (Y + 1) * (Z + 1);
};
template <bool FULL>
__global__ __launch_bounds__(256) void k_gradiprism_tensor(int nx, int ny, int nz, const double *__restrict__ xe,
                                                           const double *__restrict__ ye, const double *__restrict__ ze, int nobs,
                                                           const double *__restrict__ xd, const double *__restrict__ yd,
                                                           const double *__restrict__ zd, const double *__restrict__ cw,
                                                           double *__restrict__ rows, int *__restrict__ err, double *__restrict__ sumsq)
{
    using GT = GradTile<FULL>;
    constexpr int NC = GT::NV;
    const double twopi = 2.0 * 3.14159265358979323846;
    __shared__ double T[NC][GT::NODES];
    __shared__ double s_w[4];
    init_math_tables();                                   // (published by the first __syncthreads() of the observation loop)
    // observation-independent index arithmetic, once per workgroup (as in k_prism_gz_tensor)
    __shared__ double s_xe[GT::X + 1], s_ye[GT::Y + 1], s_ze[GT::Z + 1];
    __shared__ int s_node[GT::NODES];                     // LDS slot | a << 12 | b << 18 | c << 22
    const int tiles_x = (nx + GT::X - 1) / GT::X, tiles_y = (ny + GT::Y - 1) / GT::Y;
    const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y, bz = blockIdx.x / (tiles_x * tiles_y);
    const int i0 = bx * GT::X, j0 = by * GT::Y, k0 = bz * GT::Z;
    const int cx = min(GT::X, nx - i0), cy = min(GT::Y, ny - j0), cz = min(GT::Z, nz - k0);
    const int64_t N = (int64_t)nx * ny * nz;
    const int nnode = (cx + 1) * (cy + 1) * (cz + 1);
    const int ncell = cx * cy * cz;
    int bad = 0;
#define NODE(a, b, c) (((c) * (GT::Y + 1) + (b)) * (GT::X + 1) + (a))
    for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
        const int a = n % (cx + 1), b = (n / (cx + 1)) % (cy + 1), c = n / ((cx + 1) * (cy + 1));
        s_node[n] = NODE(a, b, c) | (a << 12) | (b << 18) | (c << 22);
    }
    if (threadIdx.x <= cx) s_xe[threadIdx.x] = xe[i0 + threadIdx.x];
    if (threadIdx.x <= cy) s_ye[threadIdx.x] = ye[j0 + threadIdx.x];
    if (threadIdx.x <= cz) s_ze[threadIdx.x] = ze[k0 + threadIdx.x];
    constexpr int CPT = GT::X * GT::Y * GT::Z / 256;      // cells per thread
    int c_slot[CPT];
    int64_t c_col[CPT];
    double c_w[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int q = threadIdx.x + 256 * j;
        c_slot[j] = -1;
        c_col[j] = 0;
        c_w[j] = 1.0;
        if (q < ncell) {
            const int a = q % cx, b = (q / cx) % cy, c = q / (cx * cy);
            c_slot[j] = NODE(a, b, c);
            c_col[j] = ((int64_t)(k0 + c) * ny + (j0 + b)) * nx + (i0 + a);
            if (cw) c_w[j] = cw[c_col[j]];
        }
    }
    for (int o = 0; o < nobs; ++o) {
        const double xo = xd[o], yo = yd[o], zo = zd[o];
        lds_barrier();
        for (int n = threadIdx.x; n < nnode; n += blockDim.x) {
            const int code = s_node[n];
            const double XX = xo - s_xe[(code >> 12) & 63], YY = yo - s_ye[(code >> 18) & 15], ZZ = -(zo - s_ze[code >> 22]);   // gravity_field.f90:232-237
            const double Rs = sqrt(XX * XX + YY * YY + ZZ * ZZ);                                         // :251
            double vzz = -datan2(XX * YY, Rs * ZZ);                                                       // :255
            if (vzz < 0) vzz = vzz + twopi;
            const int id = code & 4095;
            if (FULL) {
                double vxx = datan2(XX * YY, XX * XX + Rs * ZZ + ZZ * ZZ);                                // :253
                double vyy = datan2(XX * YY, Rs * Rs + Rs * ZZ - XX * XX);                                // :254
                if (vxx < 0) vxx = vxx + twopi;
                if (vyy < 0) vyy = vyy + twopi;
                const double arg1 = Rs + ZZ;
                const double arg21 = Rs - YY, arg22 = Rs + YY;
                const double arg31 = Rs - XX, arg32 = Rs + XX;
                if (arg22 == 0. || arg32 == 0.) bad |= 16;
                const double arg2 = arg21 / arg22, arg3 = arg31 / arg32;
                if (arg1 <= 0. || arg2 <= 0. || arg3 <= 0.) bad |= 32;
                T[0][id] = vxx;
                T[NC > 1 ? 1 : 0][id] = vyy;
                T[NC > 2 ? 2 : 0][id] = vzz;
                T[NC > 3 ? 3 : 0][id] = dlog(arg1);                                                       // vxy, :286
                T[NC > 4 ? 4 : 0][id] = 0.5 * dlog(arg3);                                                 // vyz, :288
                T[NC > 5 ? 5 : 0][id] = 0.5 * dlog(arg2);                                                 // vzx, :287
            } else {
                T[0][id] = vzz;
            }
        }
        lds_barrier();
        double sq[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) sq[i] = 0.0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            if (c_slot[j] < 0) continue;
            const int64_t p = c_col[j];
#pragma unroll
            for (int comp = 0; comp < NC; ++comp) {
                const double *t0 = &T[comp][c_slot[j]];
                double gsum = 0.0;
#pragma unroll
                for (int K = 0; K < 2; ++K)
#pragma unroll
                    for (int L = 0; L < 2; ++L)
#pragma unroll
                        for (int M = 0; M < 2; ++M) {
                            const double dmu = ((K + L + M) & 1) ? 1.0 : -1.0;
                            gsum = gsum + dmu * t0[NODE(K, L, M)];
                        }
                double v = g_grav() * gsum;                                                               // :301-306, :358
                if (cw) v = v * c_w[j];
                __builtin_nontemporal_store(v, &rows[(int64_t)(o * NC + comp) * N + p]);
                sq[comp] = fma(v, v, sq[comp]);
            }
        }
        if (sumsq) {                                        // cost_full (sensitivity_gravmag.F90:234), fixed order
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                double t = sq[i];
#pragma unroll
                for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_down(t, dd);
                lds_barrier();
                if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
                lds_barrier();
                if (threadIdx.x == 0) sumsq[(int64_t)(o * NC + i) * gridDim.x + blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
            }
        }
    }
#undef NODE
    if (bad) atomicOr(err, bad);
}

// What produces the lines of one observation (sensitivity_gravmag.F90:193-220): nsub = ncd*ncm lines in (d, k) order.
enum { GEN_GZ = 0, GEN_GZZ = 1, GEN_FTG = 2, GEN_MAG = 3, GEN_G3 = 4 };
struct RowGen {
    int kind = GEN_GZ;
    int ncd = 1, ncm = 1;
    MagField mf{};
    int nsub() const { return ncd * ncm; }
};

// dircos, magnetic_field.f90:91-110 (host)
static MagField make_mag_field(double incl, double decl, double azim, double intensity)
{
    const double PI = 3.14159265358979323846;
    const double d2rad = PI / 180.0;
    const double decl2 = std::fmod(450.0 - decl, 360.0);
    const double xincl = incl * d2rad, xdecl = decl2 * d2rad, xazim = azim * d2rad;
    MagField mf;
    mf.magv[0] = std::cos(xincl) * std::cos(xdecl - xazim);
    mf.magv[1] = std::cos(xincl) * std::sin(xdecl - xazim);
    mf.magv[2] = std::sin(xincl);
    mf.intensity = intensity;
    return mf;
}

// Verifies that the six cell arrays describe a tensor-product grid with shared faces; ok[0] is cleared otherwise.
__global__ void k_check_tensor(int nx, int ny, int nz, const double *__restrict__ X1, const double *__restrict__ X2,
                               const double *__restrict__ Y1, const double *__restrict__ Y2, const double *__restrict__ Z1,
                               const double *__restrict__ Z2, const double *__restrict__ xe, const double *__restrict__ ye,
                               const double *__restrict__ ze, int *__restrict__ ok)
{
    const int64_t N = (int64_t)nx * ny * nz;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(p % nx), j = (int)((p / nx) % ny), k = (int)(p / ((int64_t)nx * ny));
        const bool good = X1[p] == xe[i] && X2[p] == xe[i + 1] && Y1[p] == ye[j] && Y2[p] == ye[j + 1] && Z1[p] == ze[k] &&
                          Z2[p] == ze[k + 1];
        if (!good) *ok = 0;
    }
}

__global__ void k_gather_edges(int nx, int ny, int nz, const double *__restrict__ X1, const double *__restrict__ X2,
                               const double *__restrict__ Y1, const double *__restrict__ Y2, const double *__restrict__ Z1,
                               const double *__restrict__ Z2, double *__restrict__ xe, double *__restrict__ ye,
                               double *__restrict__ ze)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t sxy = (int64_t)nx * ny;
    if (t < nx) xe[t] = X1[t];
    if (t == nx) xe[nx] = X2[nx - 1];
    if (t < ny) ye[t] = Y1[(int64_t)t * nx];
    if (t == ny) ye[ny] = Y2[(int64_t)(ny - 1) * nx];
    if (t < nz) ze[t] = Z1[(int64_t)t * sxy];
    if (t == nz) ze[nz] = Z2[(int64_t)(nz - 1) * sxy];
}

int detect_tensor_grid(tfx_ctx *ctx)
{
    hipStream_t s = ctx->stream;
    const int nx = ctx->nx, ny = ctx->ny, nz = ctx->nz;
    ctx->tensor_grid = false;
    TFX_TRY(ctx->edges[0].alloc(nx + 1));
    TFX_TRY(ctx->edges[1].alloc(ny + 1));
    TFX_TRY(ctx->edges[2].alloc(nz + 1));
    DBuf<int> ok;
    TFX_TRY(ok.alloc(1));
    int one = 1;
    TFX_HIP(hipMemcpyAsync(ok.p, &one, sizeof(int), hipMemcpyHostToDevice, s));
    const int m = std::max(nx, std::max(ny, nz)) + 1;
    hipLaunchKernelGGL(k_gather_edges, dim3((m + 255) / 256), dim3(256), 0, s, nx, ny, nz, ctx->grid[0].p, ctx->grid[1].p,
                       ctx->grid[2].p, ctx->grid[3].p, ctx->grid[4].p, ctx->grid[5].p, ctx->edges[0].p, ctx->edges[1].p,
                       ctx->edges[2].p);
    const int grid = (int)std::min<int64_t>((ctx->N + 255) / 256, (int64_t)ctx->num_cu * 8);
    hipLaunchKernelGGL(k_check_tensor, dim3(grid), dim3(256), 0, s, nx, ny, nz, ctx->grid[0].p, ctx->grid[1].p, ctx->grid[2].p,
                       ctx->grid[3].p, ctx->grid[4].p, ctx->grid[5].p, ctx->edges[0].p, ctx->edges[1].p, ctx->edges[2].p, ok.p);
    TFX_HIP(hipGetLastError());
    int h = 0;
    TFX_HIP(hipMemcpyAsync(&h, ok.p, sizeof(int), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    ctx->tensor_grid = (h == 1) && !ctx->force_general_prism;
    return 0;
}

static int grad_tiles(tfx_ctx *ctx, bool full)
{
    const int X = full ? GradTile<true>::X : GradTile<false>::X, Y = 8, Z = full ? GradTile<true>::Z : GradTile<false>::Z;
    return ((ctx->nx + X - 1) / X) * ((ctx->ny + Y - 1) / Y) * ((ctx->nz + Z - 1) / Z);
}

// Lines of a batch of observations already on the device: d_rows[(o*nsub + sub)*N + p]; picks the tensor-grid kernel
// when it applies.  d_sumsq (optional): [nobs*nsub][*nblk] partial sums of squares per line.
int prism_rows_dev(tfx_ctx *ctx, const RowGen &gen, int nobs, const double *d_x, const double *d_y, const double *d_z,
                   const double *d_cw, double *d_rows, int *d_err, double *d_sumsq = nullptr, int *nblk = nullptr)
{
    if (nobs * gen.nsub() > PRISM_MAX_BATCH) return fail(TFX_E_ARG, "prism batch %d > %d", nobs * gen.nsub(), PRISM_MAX_BATCH);
    hipStream_t s = ctx->stream;
    const int64_t N = ctx->N;
    const int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)ctx->num_cu * 16);
#define GRID_ARGS N, ctx->grid[0].p, ctx->grid[1].p, ctx->grid[2].p, ctx->grid[3].p, ctx->grid[4].p, ctx->grid[5].p, nobs, d_x, d_y, d_z, d_cw
    if (gen.kind == GEN_MAG && ctx->tensor_grid) {
        const int tiles = ((ctx->nx + MT_X - 1) / MT_X) * ((ctx->ny + MT_Y - 1) / MT_Y) * ((ctx->nz + MT_Z - 1) / MT_Z);
#define TENSOR_ARGS ctx->nx, ctx->ny, ctx->nz, ctx->edges[0].p, ctx->edges[1].p, ctx->edges[2].p, nobs, d_x, d_y, d_z, d_cw, gen.mf, d_rows, d_err, d_sumsq
#define FIX_ARGS ctx->nx, ctx->ny, ctx->nz, ctx->edges[0].p, ctx->edges[1].p, ctx->edges[2].p, d_x, d_y, d_z, d_cw, gen.mf, d_rows, d_err, d_sumsq, tiles
#define MAG_TENSOR(M, D)                                                                                      \
    do {                                                                                                      \
        hipLaunchKernelGGL((k_magprism_tensor<M, D>), dim3(tiles), dim3(256), 0, s, TENSOR_ARGS);              \
        hipLaunchKernelGGL((k_magprism_inside_fix<M, D>), dim3(nobs), dim3(64), 0, s, FIX_ARGS);               \
    } while (0)
        if (gen.ncm == 1 && gen.ncd == 1) MAG_TENSOR(1, 1);
        else if (gen.ncm == 1 && gen.ncd == 3) MAG_TENSOR(1, 3);
        else if (gen.ncm == 3 && gen.ncd == 1) MAG_TENSOR(3, 1);
        else if (gen.ncm == 3 && gen.ncd == 3) MAG_TENSOR(3, 3);
        else return fail(TFX_E_ARG, "Wrong number of components in magnetic_field_magprism!");
#undef MAG_TENSOR
#undef FIX_ARGS
#undef TENSOR_ARGS
        if (nblk) *nblk = tiles;
    } else if (gen.kind == GEN_MAG) {
        if (gen.ncm == 1 && gen.ncd == 1) hipLaunchKernelGGL((k_magprism<1, 1>), dim3(grid), dim3(256), 0, s, GRID_ARGS, gen.mf, d_rows, d_err, d_sumsq);
        else if (gen.ncm == 1 && gen.ncd == 3) hipLaunchKernelGGL((k_magprism<1, 3>), dim3(grid), dim3(256), 0, s, GRID_ARGS, gen.mf, d_rows, d_err, d_sumsq);
        else if (gen.ncm == 3 && gen.ncd == 1) hipLaunchKernelGGL((k_magprism<3, 1>), dim3(grid), dim3(256), 0, s, GRID_ARGS, gen.mf, d_rows, d_err, d_sumsq);
        else if (gen.ncm == 3 && gen.ncd == 3) hipLaunchKernelGGL((k_magprism<3, 3>), dim3(grid), dim3(256), 0, s, GRID_ARGS, gen.mf, d_rows, d_err, d_sumsq);
        else return fail(TFX_E_ARG, "Wrong number of components in magnetic_field_magprism!");            // magnetic_field.f90:258-282
        if (nblk) *nblk = grid;
    } else if ((gen.kind == GEN_GZZ || gen.kind == GEN_FTG) && ctx->tensor_grid) {
        const int tiles = grad_tiles(ctx, gen.kind == GEN_FTG);
        if (gen.kind == GEN_FTG)
            hipLaunchKernelGGL((k_gradiprism_tensor<true>), dim3(tiles), dim3(256), 0, s, ctx->nx, ctx->ny, ctx->nz, ctx->edges[0].p,
                               ctx->edges[1].p, ctx->edges[2].p, nobs, d_x, d_y, d_z, d_cw, d_rows, d_err, d_sumsq);
        else
            hipLaunchKernelGGL((k_gradiprism_tensor<false>), dim3(tiles), dim3(256), 0, s, ctx->nx, ctx->ny, ctx->nz, ctx->edges[0].p,
                               ctx->edges[1].p, ctx->edges[2].p, nobs, d_x, d_y, d_z, d_cw, d_rows, d_err, d_sumsq);
        if (nblk) *nblk = tiles;
    } else if (gen.kind == GEN_GZZ) {
        hipLaunchKernelGGL((k_gradiprism<false>), dim3(grid), dim3(256), 0, s, GRID_ARGS, d_rows, d_err, d_sumsq);
        if (nblk) *nblk = grid;
    } else if (gen.kind == GEN_FTG) {
        hipLaunchKernelGGL((k_gradiprism<true>), dim3(grid), dim3(256), 0, s, GRID_ARGS, d_rows, d_err, d_sumsq);
        if (nblk) *nblk = grid;
    } else if (gen.kind == GEN_G3 && ctx->tensor_grid) {
        const int tiles = ((ctx->nx + G3_X - 1) / G3_X) * ((ctx->ny + G3_Y - 1) / G3_Y) * ((ctx->nz + G3_Z - 1) / G3_Z);
        hipLaunchKernelGGL(k_prism_g3_tensor, dim3(tiles), dim3(256), 0, s, ctx->nx, ctx->ny, ctx->nz, ctx->edges[0].p, ctx->edges[1].p,
                           ctx->edges[2].p, nobs, d_x, d_y, d_z, d_cw, d_rows, d_err, d_sumsq);
        if (nblk) *nblk = tiles;
    } else if (gen.kind == GEN_G3) {
        hipLaunchKernelGGL(k_prism_g3, dim3(grid), dim3(256), 0, s, GRID_ARGS, d_rows, d_err, d_sumsq);
        if (nblk) *nblk = grid;
    } else if (ctx->tensor_grid) {
        const int tiles = ((ctx->nx + PT_X - 1) / PT_X) * ((ctx->ny + PT_Y - 1) / PT_Y) * ((ctx->nz + PT_Z - 1) / PT_Z);
        // (gen_grid_limit: the build's overlap mode caps the resident workgroups - see build_kernel_any)
        const int wgs = ctx->gen_grid_limit > 0 ? std::min(tiles, ctx->gen_grid_limit) : tiles;
        hipLaunchKernelGGL(k_prism_gz_tensor, dim3(wgs), dim3(256), 0, s, ctx->nx, ctx->ny, ctx->nz, ctx->edges[0].p,
                           ctx->edges[1].p, ctx->edges[2].p, nobs, d_x, d_y, d_z, d_cw, d_rows, d_err, d_sumsq, tiles);
        if (nblk) *nblk = tiles;
    } else {
        hipLaunchKernelGGL(k_prism_gz, dim3(grid), dim3(256), 0, s, GRID_ARGS, d_rows, d_err, d_sumsq);
        if (nblk) *nblk = grid;
    }
#undef GRID_ARGS
    TFX_HIP(hipGetLastError());
    return 0;
}

// number of sum-of-squares partials per line that prism_rows_dev will write
int prism_partials(tfx_ctx *ctx, const RowGen &gen)
{
    if (ctx->tensor_grid && gen.kind == GEN_GZ) return ((ctx->nx + PT_X - 1) / PT_X) * ((ctx->ny + PT_Y - 1) / PT_Y) * ((ctx->nz + PT_Z - 1) / PT_Z);
    if (ctx->tensor_grid && gen.kind == GEN_G3) return ((ctx->nx + G3_X - 1) / G3_X) * ((ctx->ny + G3_Y - 1) / G3_Y) * ((ctx->nz + G3_Z - 1) / G3_Z);
    if (ctx->tensor_grid && (gen.kind == GEN_GZZ || gen.kind == GEN_FTG)) return grad_tiles(ctx, gen.kind == GEN_FTG);
    if (ctx->tensor_grid && gen.kind == GEN_MAG) return ((ctx->nx + MT_X - 1) / MT_X) * ((ctx->ny + MT_Y - 1) / MT_Y) * ((ctx->nz + MT_Z - 1) / MT_Z);
    return (int)std::min<int64_t>((ctx->N + 255) / 256, (int64_t)ctx->num_cu * 16);
}

// out[row] = sum of red[row][0..n) in index order
__global__ void k_rows_final_sum(const double *__restrict__ red, int n, double *__restrict__ out)
{
    const int row = blockIdx.x;
    const double *r = red + (int64_t)row * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += r[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// =============================================================================================================
// column weight, depth weighting type 1
// =============================================================================================================
__global__ void k_depth_weight(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                               const double *__restrict__ Y1, const double *__restrict__ Y2, const double *__restrict__ Z1,
                               const double *__restrict__ Z2, double power, double Z0, double *__restrict__ w,
                               unsigned long long *__restrict__ maxbits, int *__restrict__ err)
{
    double mx = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
        const double depth = 0.5 * (Z1[p] + Z2[p]);                                          // grid.F90:277
        double v = 0.0;
        if (depth + Z0 > 0.0) v = pow(depth + Z0, -power / 2.0);                             // weights_gravmag.f90:214-215
        else atomicOr(err, 1);
        const double vol = fabs((X2[p] - X1[p]) * (Y2[p] - Y1[p]) * (Z2[p] - Z1[p]));        // grid.F90:289-291
        v = v * sqrt(vol);                                                                   // :174
        w[p] = v;
        mx = fmax(mx, v);
    }
    // positive doubles order like their bit patterns
    atomicMax(maxbits, (unsigned long long)__double_as_longlong(mx));
}

__global__ void k_depth_weight_finish(int64_t N, double *__restrict__ w, const unsigned long long *__restrict__ maxbits,
                                      double multiplier, int *__restrict__ err)
{
    const double norm = __longlong_as_double((long long)*maxbits);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
        double v = w[p] / norm;                                                              // :243
        if (v == 0.0) { atomicOr(err, 2); continue; }
        v = 1.0 / v;                                                                         // :190
        w[p] = v * multiplier;                                                               // problem_joint_gravmag.F90:178
    }
}

// distance weighting, type 2 (weights_gravmag.f90:81-138): one thread per cell, sequential sum over the data in the
// reference's order; the data coordinates are wave-uniform scalar loads.  (R+R0)^power: x*x / x*x*x for the usual
// integer powers (what a correctly rounded pow returns up to the last bit), pow() otherwise.
__device__ __forceinline__ double pow_int_or_general(double x, double p)
{
    if (p == 2.0) return x * x;
    if (p == 3.0) return x * x * x;
    return pow(x, p);
}

__global__ __launch_bounds__(256) void k_distance_weight(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                                                         const double *__restrict__ Y1, const double *__restrict__ Y2,
                                                         const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                         int64_t ndata, const double *__restrict__ xd, const double *__restrict__ yd,
                                                         const double *__restrict__ zd, double power, double beta,
                                                         double *__restrict__ w, unsigned long long *__restrict__ maxbits)
{
    const double R0 = 0.1, dfactor = 0.25;                                                  // :85-88
    double mx = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
        const double x1 = X1[p], x2 = X2[p], y1 = Y1[p], y2 = Y2[p], z1 = Z1[p], z2 = Z2[p];
        const double dVj = fabs((x2 - x1) * (y2 - y1) * (z2 - z1));
        const double dhx = dfactor * fabs(x2 - x1), dhy = dfactor * fabs(y2 - y1), dhz = dfactor * fabs(z2 - z1);
        double wr = 0.0;
        for (int64_t j = 0; j < ndata; ++j) {
            double dx[2], dy[2], dz[2];
            double t;
            t = x1 + dhx - xd[j]; dx[0] = t * t;                                            // :100-106
            t = y1 + dhy - yd[j]; dy[0] = t * t;
            t = z1 + dhz - zd[j]; dz[0] = t * t;
            t = x2 - dhx - xd[j]; dx[1] = t * t;
            t = y2 - dhy - yd[j]; dy[1] = t * t;
            t = z2 - dhz - zd[j]; dz[1] = t * t;
            double integral = 0.0;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const double R = sqrt(dx[ii] + dy[jj] + dz[kk]);                    // :115
                        integral = integral + 1.0 / pow_int_or_general(R + R0, power);      // :121-123
                    }
            integral = integral * dVj / 8.0;                                                // :124
            wr = wr + integral * integral;                                                  // :126
        }
        double v = (1.0 / sqrt(dVj)) * pow(wr, beta / 4.0);                                 // :130
        v = v * sqrt(dVj);                                                                  // :174
        w[p] = v;
        mx = fmax(mx, v);
    }
    atomicMax(maxbits, (unsigned long long)__double_as_longlong(mx));
}

// minimum-distance weighting, type 3 (weights_gravmag.f90:140-162): w = sqrt(1 / (min_j |cell centre - datum j| + R0)^power), then
// sqrt(volume) (:174).  One thread per cell; the data coordinates are wave-uniform scalar loads; the minimum is taken over the
// distances themselves (the reference compares sqrt values), so ties and rounding behave the same.
__global__ __launch_bounds__(256) void k_mindist_weight(int64_t N, const double *__restrict__ X1, const double *__restrict__ X2,
                                                        const double *__restrict__ Y1, const double *__restrict__ Y2,
                                                        const double *__restrict__ Z1, const double *__restrict__ Z2,
                                                        int64_t ndata, const double *__restrict__ xd, const double *__restrict__ yd,
                                                        const double *__restrict__ zd, double power, double *__restrict__ w,
                                                        unsigned long long *__restrict__ maxbits)
{
    const double R0 = 0.01;                                                                  // :142
    double mx = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) {
        const double x1 = X1[p], x2 = X2[p], y1 = Y1[p], y2 = Y2[p], z1 = Z1[p], z2 = Z2[p];
        const double xc = 0.5 * (x1 + x2), yc = 0.5 * (y1 + y2), zc = 0.5 * (z1 + z2);       // grid.F90:248-277
        double mind2 = 1.e60;
        // sqrt is monotone and correctly rounded: min over sqrt(d2) = sqrt(min d2); the 1.d30 start value of :149 caps it below
        for (int64_t j = 0; j < ndata; ++j) {
            const double dx = xc - xd[j], dy = yc - yd[j], dz = zc - zd[j];
            const double d2 = dx * dx + dy * dy + dz * dz;                                   // :151-153
            mind2 = fmin(mind2, d2);
        }
        const double mindist = fmin(sqrt(mind2), 1.e30);
        double v = sqrt(1.0 / pow_int_or_general(mindist + R0, power));                      // :160
        v = v * sqrt(fabs((x2 - x1) * (y2 - y1) * (z2 - z1)));                               // :174
        w[p] = v;
        mx = fmax(mx, v);
    }
    atomicMax(maxbits, (unsigned long long)__double_as_longlong(mx));
}

// =============================================================================================================
// lifting wavelets: one LDS tile = XT lines x L positions, all levels of the axis done in LDS
// =============================================================================================================
struct WaveAxis {
    int L;            // line length
    int XT;           // lines per tile
    int P;            // LDS pitch (doubles) between consecutive positions
    int64_t astride;  // global stride between consecutive positions of a line
    int mode;         // 0: x axis (lines contiguous: line l at l*L), 1: y/z axis (inner-contiguous)
    int64_t inner;    // mode 1: number of contiguous inner elements (nx for y, nx*ny for z)
    int64_t outer_stride;   // mode 1: stride between outer slabs (nx*ny for y; unused for z)
    int64_t nlines;   // mode 0: number of lines per vector
    int64_t ntiles_inner;   // mode 1: tiles per outer slab
};

struct WaveConst { double sq2, c0, c1, c2, c3, c4; };
constexpr int WAVE_FUSE_ITEMS = 8;       // pairs per thread the fused D4 level keeps in registers (L <= 256 at 16 lines per tile)

// All levels of one axis on the LDS tile T (L positions x nq lines, pitch P): the lifting steps of wavelet_transform.F90 with the
// reference's operations in the reference's order.  Ends with a barrier.  The barriers order LDS only (lds_barrier): global loads a
// caller has in flight (k_wavelet_axis_pipe prefetches the next tile) stay in flight across the levels.
template <int TYPE, int DIR>
__device__ __forceinline__ void wave_levels(double *__restrict__ T, const int L, const int P, const int nq, const bool fast, const int q,
                                            const int mr, const int MR, const bool nq_pow2, const int nq_shift, const int tid, const int nt,
                                            const WaveConst &wc)
{
#define DIVQ(e) (nq_pow2 ? ((e) >> nq_shift) : ((e) / nq))
    int nscale = 0;
    while ((2 << nscale) <= L) ++nscale;      // = int(log(L)/log(2)) of wavelet_transform.F90:85 for every L < 5000
    for (int lv = 0; lv < nscale; ++lv) {
        const int istep = (DIR == 1) ? (lv + 1) : (nscale - lv);
        const int step = 1 << istep;
        const int ngmin = step / 2;               // 0-based index of the first detail slot
        const int ng = (L - 1 - ngmin) / step + 1;
        const int ilmax = (ng - 1) * step;
        const int work = ng * nq;
        if (fast) {
            // Full tile (nq == XT, a power of two): thread (q, mr) owns line q and the pairs m = mr, mr + MR, ...; the LDS index of
            // LO(m) advances by a constant per item, so a pass costs two integer adds per item instead of a divide / multiply chain.
            const int a0 = (mr * step) * P + q, dA = MR * step * P, dHI = ngmin * P, dS = step * P;
#define FOR_ITEMS for (int m = mr, a = a0; m < ng; m += MR, a += dA)
            if (TYPE == 1 && DIR == 1) {
                FOR_ITEMS {
                    double lo = T[a], hi = T[a + dHI];
                    hi = hi - lo;
                    lo = lo + hi / 2.0;
                    lo = lo * wc.sq2;
                    hi = hi / wc.sq2;
                    T[a] = lo; T[a + dHI] = hi;
                }
                lds_barrier();
            } else if (TYPE == 1 && DIR == 2) {
                FOR_ITEMS {
                    double lo = T[a], hi = T[a + dHI];
                    lo = lo / wc.sq2;
                    hi = hi * wc.sq2;
                    lo = lo - hi / 2.0;
                    hi = hi + lo;
                    T[a] = lo; T[a + dHI] = hi;
                }
                lds_barrier();
            } else if (TYPE == 2 && DIR == 1 && ng <= WAVE_FUSE_ITEMS * MR) {
                // Fused D4 level: a thread owns a run of CONSECUTIVE pairs m0 .. m0+cnt-1 of its line and produces their final
                // (lo, hi) from the RAW values of pairs m0-1 .. m0+cnt, with exactly the reference's operations (same bits):
                //   lo1(m) = lo + hi*c0 (:296-300); hi2(m) = hi - lo1(m)*c1 - lo1(m-1)*c2 (:302-319, m-1 wraps to the last pair);
                //   lo3(m) = lo1(m) - hi2(m+1) (:321-345, m+1 wraps to pair 0); lo3*c3, hi2*c4 (:347-363).
                // lo1 / hi2 of a pair are computed once per thread and handed down the run (cnt + 2 pairs read for cnt written:
                // 20 LDS reads and 9 flops per pair at 8 pairs per thread, against 48 and 17 with strided ownership); two barriers
                // per level instead of four.  Lanes 0-31 of a wave (2 values of mr x 16 lines) read 32 distinct bank pairs.
                const int ipt = (ng + MR - 1) / MR;
                const int m0 = mr * ipt;
                const int cnt = min(ipt, ng - m0);
                double olo[WAVE_FUSE_ITEMS], ohi[WAVE_FUSE_ITEMS];
                if (cnt > 0) {
                    const int apv = ((m0 == 0 ? ng - 1 : m0 - 1) * step) * P + q;
                    const int av = (m0 * step) * P + q;
                    const double lo1_prev = T[apv] + T[apv + dHI] * wc.c0;
                    const double hi0 = T[av + dHI];
                    double lo1_cur = T[av] + hi0 * wc.c0;
                    double hi2_cur = hi0 - lo1_cur * wc.c1 - lo1_prev * wc.c2;
#pragma unroll
                    for (int i = 0; i < WAVE_FUSE_ITEMS; ++i) {
                        if (i < cnt) {
                            const int an = ((m0 + i == ng - 1 ? 0 : m0 + i + 1) * step) * P + q;
                            const double hin = T[an + dHI];
                            const double lo1n = T[an] + hin * wc.c0;
                            const double hi2n = hin - lo1n * wc.c1 - lo1_cur * wc.c2;
                            olo[i] = (lo1_cur - hi2n) * wc.c3;
                            ohi[i] = hi2_cur * wc.c4;
                            lo1_cur = lo1n;
                            hi2_cur = hi2n;
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int i = 0; i < WAVE_FUSE_ITEMS; ++i) {
                    if (i < cnt) {
                        const int a = ((m0 + i) * step) * P + q;
                        T[a] = olo[i]; T[a + dHI] = ohi[i];
                    }
                }
                lds_barrier();
            } else if (TYPE == 2 && DIR == 1) {
                FOR_ITEMS { T[a] = T[a] + T[a + dHI] * wc.c0; }
                lds_barrier();
                FOR_ITEMS {
                    const double prev = (m == 0) ? T[ilmax * P + q] : T[a - dS];
                    T[a + dHI] = T[a + dHI] - T[a] * wc.c1 - prev * wc.c2;
                }
                lds_barrier();
                FOR_ITEMS {
                    const double nxt = (m == ng - 1) ? T[dHI + q] : T[a + dHI + dS];
                    T[a] = T[a] - nxt;
                }
                lds_barrier();
                FOR_ITEMS { T[a] = T[a] * wc.c3; T[a + dHI] = T[a + dHI] * wc.c4; }
                lds_barrier();
            } else {
                FOR_ITEMS { T[a] = T[a] * wc.c4; T[a + dHI] = T[a + dHI] * wc.c3; }
                lds_barrier();
                FOR_ITEMS {
                    const double nxt = (m == ng - 1) ? T[dHI + q] : T[a + dHI + dS];
                    T[a] = T[a] + nxt;
                }
                lds_barrier();
                FOR_ITEMS {
                    const double prev = (m == 0) ? T[ilmax * P + q] : T[a - dS];
                    T[a + dHI] = T[a + dHI] + T[a] * wc.c1 + prev * wc.c2;
                }
                lds_barrier();
                FOR_ITEMS { T[a] = T[a] - T[a + dHI] * wc.c0; }
                lds_barrier();
            }
#undef FOR_ITEMS
            continue;
        }
#define LO(m) T[((m) * step) * P + q]
#define HI(m) T[(ngmin + (m) * step) * P + q]
        if (TYPE == 1 && DIR == 1) {          // Haar forward, wavelet_transform.F90:103-149 (per pair, fused)
            for (int e = tid; e < work; e += nt) {
                const int m = DIVQ(e), q = e - m * nq;
                double lo = LO(m), hi = HI(m);
                hi = hi - lo;
                lo = lo + hi / 2.0;
                lo = lo * wc.sq2;
                hi = hi / wc.sq2;
                LO(m) = lo; HI(m) = hi;
            }
            lds_barrier();
        } else if (TYPE == 1 && DIR == 2) {   // Haar inverse, :186-232
            for (int e = tid; e < work; e += nt) {
                const int m = DIVQ(e), q = e - m * nq;
                double lo = LO(m), hi = HI(m);
                lo = lo / wc.sq2;
                hi = hi * wc.sq2;
                lo = lo - hi / 2.0;
                hi = hi + lo;
                LO(m) = lo; HI(m) = hi;
            }
            lds_barrier();
        } else if (TYPE == 2 && DIR == 1) {   // D4 forward, :284-365
            for (int e = tid; e < work; e += nt) { const int m = DIVQ(e), q = e - m * nq; LO(m) = LO(m) + HI(m) * wc.c0; }
            lds_barrier();
            for (int e = tid; e < work; e += nt) {
                const int m = DIVQ(e), q = e - m * nq;
                const double prev = (m == 0) ? T[ilmax * P + q] : LO(m - 1);
                HI(m) = HI(m) - LO(m) * wc.c1 - prev * wc.c2;
            }
            lds_barrier();
            for (int e = tid; e < work; e += nt) {
                const int m = DIVQ(e), q = e - m * nq;
                const double nxt = (m == ng - 1) ? HI(0) : HI(m + 1);
                LO(m) = LO(m) - nxt;
            }
            lds_barrier();
            for (int e = tid; e < work; e += nt) { const int m = DIVQ(e), q = e - m * nq; LO(m) = LO(m) * wc.c3; HI(m) = HI(m) * wc.c4; }
            lds_barrier();
        } else {                              // D4 inverse, :413-495
            for (int e = tid; e < work; e += nt) { const int m = DIVQ(e), q = e - m * nq; LO(m) = LO(m) * wc.c4; HI(m) = HI(m) * wc.c3; }
            lds_barrier();
            for (int e = tid; e < work; e += nt) {
                const int m = DIVQ(e), q = e - m * nq;
                const double nxt = (m == ng - 1) ? HI(0) : HI(m + 1);
                LO(m) = LO(m) + nxt;
            }
            lds_barrier();
            for (int e = tid; e < work; e += nt) {
                const int m = DIVQ(e), q = e - m * nq;
                const double prev = (m == 0) ? T[ilmax * P + q] : LO(m - 1);
                HI(m) = HI(m) + LO(m) * wc.c1 + prev * wc.c2;
            }
            lds_barrier();
            for (int e = tid; e < work; e += nt) { const int m = DIVQ(e), q = e - m * nq; LO(m) = LO(m) - HI(m) * wc.c0; }
            lds_barrier();
        }
#undef LO
#undef HI
    }
#undef DIVQ
}

template <int TYPE, int DIR>
__global__ __launch_bounds__(256) void k_wavelet_axis(double *__restrict__ s, int64_t vec_stride, WaveAxis ax, WaveConst wc)
{
    extern __shared__ __attribute__((aligned(16))) double T[];
    const int L = ax.L, XT = ax.XT, P = ax.P;
    double *base = s + (int64_t)blockIdx.y * vec_stride;
    // which lines does this tile hold
    int64_t g0;       // global offset of (line q = 0, position 0)
    int nq;           // valid lines in the tile
    if (ax.mode == 0) {
        const int64_t l0 = (int64_t)blockIdx.x * XT;
        nq = (int)min((int64_t)XT, ax.nlines - l0);
        g0 = l0 * L;
    } else {
        const int64_t o = blockIdx.x / ax.ntiles_inner, ti = blockIdx.x % ax.ntiles_inner;
        const int64_t m0 = ti * XT;
        nq = (int)min((int64_t)XT, ax.inner - m0);
        g0 = o * ax.outer_stride + m0;
    }
    const int tid = threadIdx.x, nt = blockDim.x;
    typedef double d2 __attribute__((ext_vector_type(2)));
    // e -> (e / nq, e % nq) without an integer division when nq is a power of two (full tiles)
    const bool nq_pow2 = (nq & (nq - 1)) == 0;
    const int nq_shift = 31 - __clz(nq);
#define DIVQ(e) (nq_pow2 ? ((e) >> nq_shift) : ((e) / nq))
    const bool fast = nq == XT && nq_pow2 && (nt & (nq - 1)) == 0;
    const int q = tid & (nq - 1), mr = tid >> nq_shift, MR = nt >> nq_shift;      // fast path: line and first pair of this thread
    // ---- load: 8 independent global loads in flight per thread, then the LDS writes
    // (x axis: element e = q*L + a; e advances by nt per step, so (q, a) advance by (nt / L, nt % L) - one integer division
    // per thread instead of one per element)
    const int total = nq * L;
    const int dq = nt / L, da = nt - dq * L;
    const bool wide_x = ax.mode == 0 && (L & 1) == 0 && (vec_stride & 1) == 0;       // (the pairs are 16-byte aligned)
    const int npair = total / 2;
    const int dq2 = (2 * nt) / L, da2 = 2 * nt - dq2 * L;
    int rq = tid / L, ra = tid - rq * L;
    // y / z axis, full tile, even strides: 16-byte loads / stores - thread (q2, mr2) moves lines 2 q2, 2 q2 + 1 of positions mr2, mr2 + MR2, ...
    const bool wide_yz = ax.mode == 1 && fast && nq >= 2 && (ax.astride & 1) == 0 && (g0 & 1) == 0 && (vec_stride & 1) == 0;
    const int sh2 = nq_shift > 0 ? nq_shift - 1 : 0;
    const int hq = nq >> 1, q2 = tid & (hq - 1), mr2 = tid >> sh2, MR2 = nt >> sh2;
    if (wide_yz) {
        const double *gp = base + g0 + (int64_t)mr2 * ax.astride + 2 * q2;
        const int64_t gstep = (int64_t)MR2 * ax.astride;
        int la0 = mr2 * P + 2 * q2;
        const int lstep = MR2 * P;
        for (int ab = mr2; ab < L; ab += MR2 * 8) {
            d2 tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ab + k * MR2 < L) tmp[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(&gp[k * gstep]));
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ab + k * MR2 < L) { T[la0 + k * lstep] = tmp[k].x; T[la0 + k * lstep + 1] = tmp[k].y; }
            gp += 8 * gstep;
            la0 += 8 * lstep;
        }
    } else if (ax.mode == 1 && fast) {
        // y / z axis, full tile: thread (q, mr) walks positions a = mr, mr + MR, ... of line q with constant strides
        const double *gp = base + g0 + (int64_t)mr * ax.astride + q;
        const int64_t gstep = (int64_t)MR * ax.astride;
        int la0 = mr * P + q;
        const int lstep = MR * P;
        for (int ab = mr; ab < L; ab += MR * 8) {
            double tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ab + k * MR < L) tmp[k] = __builtin_nontemporal_load(&gp[k * gstep]);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ab + k * MR < L) T[la0 + k * lstep] = tmp[k];
            gp += 8 * gstep;
            la0 += 8 * lstep;
        }
    } else if (wide_x) {
        // x axis, even line length: the tile is one contiguous run of nq*L doubles - 16-byte loads, two positions of a line each
        int rq2 = (2 * tid) / L, ra2 = 2 * tid - rq2 * L;
        for (int p0 = tid; p0 < npair; p0 += nt * 8) {
            d2 tmp[8];
            int la[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int p = p0 + k * nt;
                la[k] = -1;
                if (p < npair) { la[k] = ra2 * P + rq2; tmp[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(&base[g0 + 2 * (int64_t)p])); }
                rq2 += dq2; ra2 += da2;
                if (ra2 >= L) { ra2 -= L; rq2 += 1; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (la[k] >= 0) { T[la[k]] = tmp[k].x; T[la[k] + P] = tmp[k].y; }
        }
    } else
    for (int e0 = tid; e0 < total; e0 += nt * 8) {
        double tmp[8];
        int la[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * nt;
            la[k] = -1;
            if (e < total) {
                if (ax.mode == 0) { la[k] = ra * P + rq; tmp[k] = __builtin_nontemporal_load(&base[g0 + e]); }      // the XT lines of an x tile are contiguous: q*L + a = e
                else { const int a = DIVQ(e), q = e - a * nq; la[k] = a * P + q; tmp[k] = base[g0 + (int64_t)a * ax.astride + q]; }
            }
            rq += dq; ra += da;
            if (ra >= L) { ra -= L; rq += 1; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (la[k] >= 0) T[la[k]] = tmp[k];
    }
    __syncthreads();
    wave_levels<TYPE, DIR>(T, L, P, nq, fast, q, mr, MR, nq_pow2, nq_shift, tid, nt, wc);
    // ---- store
    rq = tid / L; ra = tid - rq * L;
    if (wide_yz) {
        double *gp = base + g0 + (int64_t)mr2 * ax.astride + 2 * q2;
        const int64_t gstep = (int64_t)MR2 * ax.astride;
        int la0 = mr2 * P + 2 * q2;
        const int lstep = MR2 * P;
        for (int ab = mr2; ab < L; ab += MR2 * 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ab + k * MR2 < L) {
                    d2 v;
                    v.x = T[la0 + k * lstep];
                    v.y = T[la0 + k * lstep + 1];
                    __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(&gp[k * gstep]));
                }
            gp += 8 * gstep;
            la0 += 8 * lstep;
        }
    } else if (ax.mode == 1 && fast) {
        double *gp = base + g0 + (int64_t)mr * ax.astride + q;
        const int64_t gstep = (int64_t)MR * ax.astride;
        int la0 = mr * P + q;
        const int lstep = MR * P;
        for (int ab = mr; ab < L; ab += MR * 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ab + k * MR < L) __builtin_nontemporal_store(T[la0 + k * lstep], &gp[k * gstep]);
            gp += 8 * gstep;
            la0 += 8 * lstep;
        }
    } else if (wide_x) {
        int rq2 = (2 * tid) / L, ra2 = 2 * tid - rq2 * L;
        for (int p0 = tid; p0 < npair; p0 += nt * 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int p = p0 + k * nt;
                if (p < npair) {
                    d2 v;
                    v.x = T[ra2 * P + rq2];
                    v.y = T[ra2 * P + rq2 + P];
                    __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(&base[g0 + 2 * (int64_t)p]));
                }
                rq2 += dq2; ra2 += da2;
                if (ra2 >= L) { ra2 -= L; rq2 += 1; }
            }
        }
    } else
    for (int e0 = tid; e0 < total; e0 += nt * 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * nt;
            if (e < total) {
                if (ax.mode == 0) { __builtin_nontemporal_store(T[ra * P + rq], &base[g0 + e]); }
                else { const int a = DIVQ(e), q = e - a * nq; base[g0 + (int64_t)a * ax.astride + q] = T[a * P + q]; }
            }
            rq += dq; ra += da;
            if (ra >= L) { ra -= L; rq += 1; }
        }
    }
#undef DIVQ
}

// Software-pipelined form: a PERSISTENT workgroup walks tiles t = blockIdx.x, + gridDim.x, ... of the
// (vector, tile) space and issues the global loads of tile t + 1 into registers (8 x 16 bytes per thread = one 16 x 256 tile) BEFORE
// it lifts tile t in LDS, so a workgroup always has a tile of reads in flight: the HBM latency is covered by the workgroup itself and
// not by four co-resident ones (k_wavelet_axis needs 16 waves per CU; three workgroups cost it 30 %).  Restricted to what the build's
// headline shapes are: every tile full (XT lines, a power of two), L * XT <= 4096 doubles, even line length / strides (16-byte
// transfers); launch_axis falls back to k_wavelet_axis otherwise.  Same lifting code (wave_levels) -> same bits.
template <int TYPE, int DIR, int MODE>
__global__ __launch_bounds__(256) void k_wavelet_axis_pipe(double *__restrict__ s, int64_t vec_stride, WaveAxis ax, WaveConst wc, int ntiles, int total)
{
    extern __shared__ __attribute__((aligned(16))) double T[];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int L = ax.L, XT = ax.XT, P = ax.P;
    const int tid = threadIdx.x;
    constexpr int nt = 256;
    const int nq = XT, nq_shift = 31 - __clz(nq);
    const int q = tid & (nq - 1), mr = tid >> nq_shift, MR = nt >> nq_shift;
    // MODE 0 (x axis): the tile is one contiguous run of XT * L doubles; transfer k of this thread is pair p = tid + 256 k
    const int npair = (XT * L) >> 1;
    const int dq2 = (2 * nt) / L, da2 = 2 * nt - dq2 * L;
    const int rq2_0 = (2 * tid) / L, ra2_0 = 2 * tid - rq2_0 * L;
    // MODE 1 (y / z axis): thread (q2, mr2) moves lines 2 q2, 2 q2 + 1 at positions mr2, mr2 + MR2, ...
    const int sh2 = nq_shift > 0 ? nq_shift - 1 : 0;
    const int hq = nq >> 1, q2 = tid & (hq - 1), mr2 = tid >> sh2, MR2 = nt >> sh2;
    const int64_t gstep = (int64_t)MR2 * ax.astride;
    const int lstep = MR2 * P;

    auto origin = [&](int t) -> double * {
        const int vec = t / ntiles, ti = t - vec * ntiles;
        double *base = s + (int64_t)vec * vec_stride;
        if (MODE == 0) return base + (int64_t)ti * XT * L;
        const int64_t o = ti / ax.ntiles_inner, tii = ti - o * ax.ntiles_inner;
        return base + o * ax.outer_stride + tii * XT;
    };
    d2 pre[8];
    auto issue = [&](const double *src) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (tid + k * nt < npair) pre[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src + 2 * (tid + k * nt)));
        } else {
            const double *gp = src + (int64_t)mr2 * ax.astride + 2 * q2;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (mr2 + k * MR2 < L) pre[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(gp + k * gstep));
        }
    };
    auto commit = [&]() {
        if (MODE == 0) {
            int rq2 = rq2_0, ra2 = ra2_0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (tid + k * nt < npair) { T[ra2 * P + rq2] = pre[k].x; T[ra2 * P + rq2 + P] = pre[k].y; }
                rq2 += dq2; ra2 += da2;
                if (ra2 >= L) { ra2 -= L; rq2 += 1; }
            }
        } else {
            const int la0 = mr2 * P + 2 * q2;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (mr2 + k * MR2 < L) { T[la0 + k * lstep] = pre[k].x; T[la0 + k * lstep + 1] = pre[k].y; }
        }
    };
    auto store = [&](double *dst) {
        if (MODE == 0) {
            int rq2 = rq2_0, ra2 = ra2_0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (tid + k * nt < npair) {
                    d2 v;
                    v.x = T[ra2 * P + rq2];
                    v.y = T[ra2 * P + rq2 + P];
                    __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(dst + 2 * (tid + k * nt)));
                }
                rq2 += dq2; ra2 += da2;
                if (ra2 >= L) { ra2 -= L; rq2 += 1; }
            }
        } else {
            double *gp = dst + (int64_t)mr2 * ax.astride + 2 * q2;
            const int la0 = mr2 * P + 2 * q2;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (mr2 + k * MR2 < L) {
                    d2 v;
                    v.x = T[la0 + k * lstep];
                    v.y = T[la0 + k * lstep + 1];
                    __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(gp + k * gstep));
                }
        }
    };

    // LDS slots of this thread's 8 transfers: the same for every tile, and owned by this thread alone in commit, exchange and store
    // (so the exchange below needs no barrier between its LDS read and its LDS write)
    d2 out[8];
    auto exchange = [&]() {        // out = lifted tile t (LDS -> registers), LDS = tile t + 1 (registers -> LDS)
        if (MODE == 0) {
            int rq2 = rq2_0, ra2 = ra2_0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (tid + k * nt < npair) {
                    const int a = ra2 * P + rq2;
                    out[k].x = T[a]; out[k].y = T[a + P];
                    T[a] = pre[k].x; T[a + P] = pre[k].y;
                }
                rq2 += dq2; ra2 += da2;
                if (ra2 >= L) { ra2 -= L; rq2 += 1; }
            }
        } else {
            const int la0 = mr2 * P + 2 * q2;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (mr2 + k * MR2 < L) {
                    const int a = la0 + k * lstep;
                    out[k].x = T[a]; out[k].y = T[a + 1];
                    T[a] = pre[k].x; T[a + 1] = pre[k].y;
                }
        }
    };
    auto store_out = [&](double *dst) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (tid + k * nt < npair) __builtin_nontemporal_store(out[k], reinterpret_cast<d2 *>(dst + 2 * (tid + k * nt)));
        } else {
            double *gp = dst + (int64_t)mr2 * ax.astride + 2 * q2;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (mr2 + k * MR2 < L) __builtin_nontemporal_store(out[k], reinterpret_cast<d2 *>(gp + k * gstep));
        }
    };

    int t = blockIdx.x;
    if (t >= total) return;
    double *cur = origin(t);
    issue(cur);
    commit();
    lds_barrier();
    for (;;) {
        const int tn = t + (int)gridDim.x;
        const bool more = tn < total;
        double *nxt = more ? origin(tn) : cur;
        if (more) issue(nxt);                       // tile t + 1: in flight over the whole lifting of tile t
        wave_levels<TYPE, DIR>(T, L, P, nq, true, q, mr, MR, true, nq_shift, tid, nt, wc);
        if (!more) { store(cur); break; }
        // The loads of tile t + 1 are the YOUNGEST vector-memory operations here (the stores of tile t - 1 were issued before them), so
        // the wait for them drains nothing else; the stores of tile t go out after the exchange and fly over the next lifting.
        exchange();
        store_out(cur);
        lds_barrier();
        t = tn;
        cur = nxt;
    }
}

static WaveConst wave_consts()
{
    WaveConst w;
    w.sq2 = std::sqrt(2.0);
    w.c0 = std::sqrt(3.0);                                  // wavelet_transform.F90:252-256
    w.c1 = std::sqrt(3.0) / 4.0;
    w.c2 = (std::sqrt(3.0) - 2.0) / 4.0;
    w.c3 = (std::sqrt(3.0) - 1.0) / std::sqrt(2.0);
    w.c4 = (std::sqrt(3.0) + 1.0) / std::sqrt(2.0);
    return w;
}

constexpr size_t WAVE_LDS_BUDGET = 96 * 1024;

template <int TYPE, int DIR>
static int launch_axis(tfx_ctx *ctx, double *d, int64_t vec_stride, int64_t nvec, WaveAxis ax, unsigned ntiles)
{
    const size_t lds = (size_t)ax.L * ax.P * sizeof(double);
    // the attribute belongs to the (kernel, device) pair: remembered per ctx, i.e. per device (a process may drive several)
    size_t &lds_set = ctx->wave_lds_attr[(TYPE - 1) * 2 + (DIR - 1)];
    if (lds > lds_set) {
        TFX_HIP(hipFuncSetAttribute((const void *)k_wavelet_axis<TYPE, DIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WAVE_LDS_BUDGET + 4096));
        lds_set = WAVE_LDS_BUDGET + 4096;
    }
    // software-pipelined persistent form (debug key "wave_pipe" / TFX_WAVE_PIPE = workgroups per CU, 0: off): full power-of-two tiles of
    // <= 4096 doubles moved by 16-byte transfers, and enough tiles for every workgroup to pipeline over a few
    const int64_t total = (int64_t)ntiles * nvec;
    const bool full_tiles = ax.mode == 0 ? (ax.nlines % ax.XT == 0) : (ax.inner % ax.XT == 0);
    const bool pow2 = (ax.XT & (ax.XT - 1)) == 0 && ax.XT >= 2 && ax.XT <= 256;
    const bool even = ax.mode == 0 ? ((ax.L & 1) == 0 && (vec_stride & 1) == 0)
                                   : ((ax.astride & 1) == 0 && (ax.outer_stride & 1) == 0 && (vec_stride & 1) == 0);
    const int64_t pipe_grid = (int64_t)ctx->wave_pipe * ctx->num_cu;
    if (ctx->wave_pipe > 0 && full_tiles && pow2 && even && (int64_t)ax.L * ax.XT <= 4096 && total >= 4 * pipe_grid && total < (int64_t)1 << 30 &&
        lds <= 64 * 1024) {
        if (ax.mode == 0)
            hipLaunchKernelGGL((k_wavelet_axis_pipe<TYPE, DIR, 0>), dim3((unsigned)pipe_grid), dim3(256), lds, ctx->stream, d, vec_stride, ax, wave_consts(),
                               (int)ntiles, (int)total);
        else
            hipLaunchKernelGGL((k_wavelet_axis_pipe<TYPE, DIR, 1>), dim3((unsigned)pipe_grid), dim3(256), lds, ctx->stream, d, vec_stride, ax, wave_consts(),
                               (int)ntiles, (int)total);
        TFX_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL((k_wavelet_axis<TYPE, DIR>), dim3(ntiles, (unsigned)nvec), dim3(256), lds, ctx->stream, d, vec_stride, ax, wave_consts());
    TFX_HIP(hipGetLastError());
    return 0;
}

static int pick_xt(int L, int64_t avail, int want, int *XT, int *P)
{
    int xt = (int)std::min<int64_t>(want, std::max<int64_t>(1, avail));
    while (xt > 1 && (size_t)L * (size_t)(xt | 1) * sizeof(double) > WAVE_LDS_BUDGET) xt /= 2;
    int p = xt | 1;     // odd pitch
    if ((size_t)L * p * sizeof(double) > WAVE_LDS_BUDGET)
        return fail(TFX_E_ARG, "wavelet axis length %d exceeds the LDS line budget", L);
    *XT = xt;
    *P = p;
    return 0;
}

// in place on nvec device arrays of n1*n2*n3 doubles, back to back (vec_stride = n1*n2*n3)
int wavelet_dev(tfx_ctx *ctx, double *d, int n1, int n2, int n3, int64_t nvec, int type, int dir, int axis_from, int axis_to)
{
    if (type != 1 && type != 2) return fail(TFX_E_ARG, "Unknown wavelet type!");      // wavelet_transform.F90:46-48
    if (dir != 1 && dir != 2) return fail(TFX_E_ARG, "bad wavelet direction");
    if (nvec <= 0) return 0;
    const int64_t N = (int64_t)n1 * n2 * n3;
    // lines per LDS tile and axis (tuning knob TFX_WAVE_XT="x,y,z"; 16 lines of 256 doubles = 34 KB per workgroup)
    static int want[3] = {0, 0, 0};
    if (!want[0]) {
        int w[3] = {16, 16, 16};
        if (const char *e = getenv("TFX_WAVE_XT")) sscanf(e, "%d,%d,%d", &w[0], &w[1], &w[2]);
        for (int i = 0; i < 3; ++i) want[i] = std::max(1, std::min(64, w[i]));
    }
    for (int axis = axis_from; axis < axis_to; ++axis) {    // axis order x -> y -> z for forward AND inverse (:82-93, :165-176)
        WaveAxis ax{};
        unsigned ntiles = 0;
        if (axis == 0) {
            if (n1 < 2) continue;
            ax.L = n1; ax.mode = 0; ax.astride = 1; ax.nlines = (int64_t)n2 * n3;
            TFX_TRY(pick_xt(n1, ax.nlines, want[0], &ax.XT, &ax.P));
            ntiles = (unsigned)((ax.nlines + ax.XT - 1) / ax.XT);
        } else if (axis == 1) {
            if (n2 < 2) continue;
            ax.L = n2; ax.mode = 1; ax.astride = n1; ax.inner = n1; ax.outer_stride = (int64_t)n1 * n2;
            TFX_TRY(pick_xt(n2, n1, want[1], &ax.XT, &ax.P));
            ax.ntiles_inner = (n1 + ax.XT - 1) / ax.XT;
            ntiles = (unsigned)(ax.ntiles_inner * n3);
        } else {
            if (n3 < 2) continue;
            ax.L = n3; ax.mode = 1; ax.astride = (int64_t)n1 * n2; ax.inner = (int64_t)n1 * n2; ax.outer_stride = 0;
            TFX_TRY(pick_xt(n3, ax.inner, want[2], &ax.XT, &ax.P));
            ax.ntiles_inner = (ax.inner + ax.XT - 1) / ax.XT;
            ntiles = (unsigned)ax.ntiles_inner;
        }
        if (type == 1 && dir == 1) TFX_TRY((launch_axis<1, 1>(ctx, d, N, nvec, ax, ntiles)));
        else if (type == 1 && dir == 2) TFX_TRY((launch_axis<1, 2>(ctx, d, N, nvec, ax, ntiles)));
        else if (type == 2 && dir == 1) TFX_TRY((launch_axis<2, 1>(ctx, d, N, nvec, ax, ntiles)));
        else TFX_TRY((launch_axis<2, 2>(ctx, d, N, nvec, ax, ntiles)));
    }
    return 0;
}

// =============================================================================================================
// exact order statistic: thr = sort_ascending(|row|)[N-K]  (sensitivity_gravmag.F90:240-250)
// radix select on the 63-bit pattern of |x| (non-negative doubles order like unsigned integers)
// =============================================================================================================
constexpr int SEL_BINS = 4096;
constexpr int SEL_DIGIT_BITS = 12;
constexpr int SEL_NDIG = 6;

struct SelState {
    unsigned long long prefix;    // selected high bits so far (already shifted into place)
    unsigned long long rank;      // 1-based rank (ascending) inside the candidate set
    unsigned long long ncand;     // candidates in the current buffer
    unsigned long long nnext;     // append counter for the next buffer
    unsigned long long base;      // the keys of the row are stored minus this (0: full select; band.lo: band select)
    unsigned int bin;
    int top;                      // the keys of the row are < 2^top (64: full select; bit length of hi - lo: band select)
};

// digit d of a row whose keys are below 2^top: bits [max(top - 12 (d + 1), 0), top - 12 d) - for top = 64 the windows are
// 52 / 40 / 28 / 16 / 4 / 0 (the last one 4 bits wide); a window below bit 0 is empty (mask 0: every key in bin 0)
__device__ __forceinline__ void sel_window(int top, int digit, int &shift, unsigned int &mask)
{
    const int hi = max(top - SEL_DIGIT_BITS * digit, 0);
    shift = max(hi - SEL_DIGIT_BITS, 0);
    mask = (1u << (hi - shift)) - 1u;
}

// from_rows: the keys are |row| itself (digit 0 of the full select); else the candidate key buffer
__global__ __launch_bounds__(256) void k_sel_hist(const double *__restrict__ rows, int64_t N,
                                                  const unsigned long long *__restrict__ cand, int64_t cand_stride,
                                                  const SelState *__restrict__ st, int digit, unsigned int *__restrict__ hist,
                                                  int from_rows)
{
    __shared__ unsigned int h[SEL_BINS];
    const int row = blockIdx.y;
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    int shift;
    unsigned int mask;
    sel_window(st[row].top, digit, shift, mask);
    if (from_rows) {
        const double *r = rows + (int64_t)row * N;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(r[i]));
            atomicAdd(&h[(unsigned int)(key >> shift) & mask], 1u);
        }
    } else {
        const unsigned long long *c = cand + (int64_t)row * cand_stride;
        const int64_t n = (int64_t)st[row].ncand;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            atomicAdd(&h[(unsigned int)(c[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    unsigned int *g = hist + (int64_t)row * SEL_BINS;
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&g[i], h[i]);
}

// one block per row: find the bin holding the rank-th smallest, update prefix / rank
__global__ void k_sel_pick(SelState *__restrict__ st, unsigned int *__restrict__ hist, int digit)
{
    const int row = blockIdx.x;
    __shared__ unsigned long long part[256];
    unsigned int *g = hist + (int64_t)row * SEL_BINS;
    const int per = SEL_BINS / 256;
    unsigned long long s = 0;
    for (int i = 0; i < per; ++i) s += g[threadIdx.x * per + i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long rank = st[row].rank;
        unsigned long long run = 0;
        int t = 0;
        while (t < 255 && run + part[t] < rank) { run += part[t]; ++t; }
        int b = t * per;
        while (b < SEL_BINS - 1 && run + g[b] < rank) { run += g[b]; ++b; }
        st[row].bin = (unsigned int)b;
        st[row].rank = rank - run;
        int shift;
        unsigned int mask;
        sel_window(st[row].top, digit, shift, mask);
        st[row].prefix |= ((unsigned long long)b) << shift;
        st[row].nnext = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) g[i] = 0;     // ready for the next digit
}

// keep the keys that fall into the picked bin.  Each wave stages its candidates in LDS and reserves output space with
// one global atomic per flush (same-address device atomics serialise at the memory side).
constexpr int SEL_STAGE = 512;
__global__ __launch_bounds__(256) void k_sel_filter(const double *__restrict__ rows, int64_t N,
                                                    const unsigned long long *__restrict__ cand_in,
                                                    unsigned long long *__restrict__ cand_out, int64_t cand_stride,
                                                    SelState *__restrict__ st, int digit, int from_rows)
{
    __shared__ unsigned long long stage[4][SEL_STAGE];
    const int row = blockIdx.y;
    int shift;
    unsigned int mask;
    sel_window(st[row].top, digit, shift, mask);
    const unsigned int bin = st[row].bin;
    unsigned long long *out = cand_out + (int64_t)row * cand_stride;
    const int64_t n = from_rows ? N : (int64_t)st[row].ncand;
    const double *r = rows + (int64_t)row * N;
    const unsigned long long *c = cand_in + (int64_t)row * cand_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *mine = stage[wave];
    int fill = 0;                                                   // wave-uniform
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + wave * 64; i0 < n; i0 += stride) {
        const int64_t i = i0 + lane;
        unsigned long long key = 0;
        bool hit = false;
        if (i < n) {
            key = from_rows ? (unsigned long long)__double_as_longlong(fabs(r[i])) : c[i];
            hit = ((unsigned int)(key >> shift) & mask) == bin;
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (hit) mine[fill + pre] = key;
            fill += __popcll(m);
            if (fill > SEL_STAGE - 64) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&st[row].nnext, (unsigned long long)fill);
                base = __shfl(base, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // staged keys of all lanes visible to the wave
                __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < fill; k += 64) out[base + k] = mine[k];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                fill = 0;
            }
        }
    }
    if (fill > 0) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&st[row].nnext, (unsigned long long)fill);
        base = __shfl(base, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < fill; k += 64) out[base + k] = mine[k];
    }
}

__global__ void k_sel_advance(SelState *__restrict__ st, int nrows)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < nrows) st[row].ncand = st[row].nnext;
}

__global__ void k_sel_init(SelState *__restrict__ st, int nrows, unsigned long long rank)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < nrows) { st[row].prefix = 0; st[row].rank = rank; st[row].ncand = 0; st[row].nnext = 0; st[row].base = 0; st[row].bin = 0; st[row].top = 64; }
}

// one block per row, after the filter of digit `first_digit - 1` (its output: st.nnext keys in cand): the remaining digits in one
// launch.  Few candidates (the normal case of the band select: the first digit spreads the band over >= 2048 bins): ranks by
// counting in LDS.  Many (keys piled up on few values): the digit loop over the candidate list with an LDS histogram.
constexpr int FIN_CAP = 1024;
__global__ __launch_bounds__(256) void k_sel_finish(SelState *__restrict__ st, const unsigned long long *__restrict__ cand, int64_t cand_stride,
                                                    int first_digit)
{
    __shared__ unsigned long long keys[FIN_CAP];
    __shared__ unsigned int h[SEL_BINS];
    __shared__ unsigned long long part[256];
    __shared__ unsigned long long s_prefix, s_rank;
    const int row = blockIdx.x;
    const unsigned long long n = st[row].nnext;
    if (n == 0) return;                                        // a band that missed: the batch is redone
    const unsigned long long *c = cand + (int64_t)row * cand_stride;
    const unsigned long long rank = st[row].rank;
    if (n <= (unsigned long long)FIN_CAP) {
        const int m = (int)n;
        for (int i = threadIdx.x; i < m; i += blockDim.x) keys[i] = c[i];
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const unsigned long long k = keys[i];
            unsigned int less = 0, eq = 0;
            for (int j = 0; j < m; ++j) {
                const unsigned long long o = keys[j];
                less += o < k;
                eq += o == k;
            }
            if ((unsigned long long)less < rank && rank <= (unsigned long long)less + eq) st[row].prefix = k;   // equal keys write the same value
        }
        return;
    }
    const int top = st[row].top;
    if (threadIdx.x == 0) { s_prefix = st[row].prefix; s_rank = rank; }
    for (int d = first_digit; d < SEL_NDIG; ++d) {
        int shift;
        unsigned int mask;
        sel_window(top, d, shift, mask);
        const int above = max(top - SEL_DIGIT_BITS * d, 0);     // the bits from here up are decided (d >= 1: above <= 52)
        for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) h[i] = 0;
        __syncthreads();
        const unsigned long long pfx = s_prefix >> above;
        for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned long long k = c[i];
            if ((k >> above) == pfx) atomicAdd(&h[(unsigned int)(k >> shift) & mask], 1u);
        }
        __syncthreads();
        constexpr int per = SEL_BINS / 256;
        unsigned long long sum = 0;
        for (int i = 0; i < per; ++i) sum += h[threadIdx.x * per + i];
        part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long r = s_rank;
            unsigned long long run = 0;
            int t = 0;
            while (t < 255 && run + part[t] < r) { run += part[t]; ++t; }
            int b = t * per;
            while (b < SEL_BINS - 1 && run + h[b] < r) { run += h[b]; ++b; }
            s_rank = r - run;
            s_prefix |= ((unsigned long long)b) << shift;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) st[row].prefix = s_prefix;
}

// fail != null (band select): a threshold below the 1e-30 floor means the band's "everything above hi is kept" does not hold
__global__ void k_sel_result(const SelState *__restrict__ st, int nrows, double *__restrict__ thr, int *__restrict__ fail)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < nrows) {
        double t = __longlong_as_double((long long)(st[row].prefix + st[row].base));
        if (t < 1.e-30) {                                  // sensitivity_gravmag.F90:252-256
            t = 1.e-30;
            if (fail) atomicOr(fail, 2);
        }
        thr[row] = t;
    }
}

// ---- band select: bracket the order statistic from a sample, then select exactly inside the bracket -------------------------
// The full select reads every row twice (histogram + filter of digit 0) before the candidate set is small.  Here a
// hashed-position sample of SMP_N coefficients per row gives two sample order statistics lo <= hi that bracket the wanted
// one with overwhelming probability; the compaction's count pass - which reads the row anyway - counts the keys above hi
// and collects the few keys in [lo, hi]; the exact select then runs on that band only.  Rows whose band misses (or
// overflows its buffer) make the whole batch fall back to the full select, so the result is always the exact order
// statistic.
constexpr int SMP_THREADS = 1024, SMP_PER = 16, SMP_N = SMP_THREADS * SMP_PER;
constexpr int BAND_SLOTS = 2048;  // = CMP_SEG: a segment's slots hold all of it (band candidates cluster: whole planes of coarse-level
                                  // coefficients sit near the threshold); only the used slots are ever touched

struct BandRow {
    unsigned long long lo, hi;    // key bounds of the band (inclusive)
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)      // splitmix64 finaliser
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// one block per row: the rank_lo-th and rank_hi-th smallest of the sampled |c| (1-based; rank_lo < 1 -> lo = 0,
// rank_hi > SMP_N -> hi = all ones), by an 8-bit radix select on register-resident keys with LDS histograms.  Bounds, not
// values, are wanted: the select stops after SMP_DIGITS digits (sign, exponent, 20 mantissa bits) and rounds lo down / hi up.
__global__ __launch_bounds__(SMP_THREADS) void k_sel_sample(const double *__restrict__ rows, int64_t N,
                                                            int rank_lo, int rank_hi, BandRow *__restrict__ band)
{
    __shared__ unsigned int h[2][256];
    __shared__ unsigned int wsum[2][4];
    __shared__ unsigned long long s_prefix[2];
    __shared__ unsigned int s_rank[2];
    const int row = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const double *r = rows + (int64_t)row * N;
    unsigned long long key[SMP_PER];
#pragma unroll
    for (int j = 0; j < SMP_PER; ++j) {
        // hashed positions: wavelet coefficients are large on index lattices (multiples of 2^l per axis) that any regular
        // stride would over- or under-sample
        const unsigned long long t = (unsigned long long)(threadIdx.x + j * SMP_THREADS);
        key[j] = (unsigned long long)__double_as_longlong(fabs(r[mix64(t) % (unsigned long long)N]));
    }
    if (threadIdx.x < 2) {
        s_prefix[threadIdx.x] = 0;
        s_rank[threadIdx.x] = (unsigned int)min(max(threadIdx.x == 0 ? rank_lo : rank_hi, 1), SMP_N);
    }
    constexpr int SMP_DIGITS = 4;
    for (int d = 0; d < SMP_DIGITS; ++d) {
        const int shift = 56 - 8 * d;
        if (threadIdx.x < 512) h[threadIdx.x >> 8][threadIdx.x & 255] = 0;
        __syncthreads();
        const unsigned long long p0 = s_prefix[0], p1 = s_prefix[1];
#pragma unroll
        for (int j = 0; j < SMP_PER; ++j) {
            const unsigned int dig = (unsigned int)(key[j] >> shift) & 255u;
            const unsigned long long top = (d == 0) ? 0ull : (key[j] >> (shift + 8));
            const bool in0 = d == 0 || top == (p0 >> (shift + 8)), in1 = d == 0 || top == (p1 >> (shift + 8));
            // the leading digits (exponent bits) put most keys of a wave into one or two bins: count those with a ballot
            // instead of 64 serialised same-address LDS atomics
            bool todo0 = in0, todo1 = in1;
            for (int it = 0; it < 2; ++it) {
                const unsigned long long any = __ballot(todo0 || todo1);
                if (!any) break;
                const unsigned int d0 = (unsigned int)__builtin_amdgcn_readlane((int)dig, (int)__builtin_ctzll(any));
                const unsigned long long m0 = __ballot(todo0 && dig == d0), m1 = __ballot(todo1 && dig == d0);
                if (lane == 0) {
                    if (m0) atomicAdd(&h[0][d0], (unsigned int)__popcll(m0));
                    if (m1) atomicAdd(&h[1][d0], (unsigned int)__popcll(m1));
                }
                if (dig == d0) { todo0 = false; todo1 = false; }
            }
            if (todo0) atomicAdd(&h[0][dig], 1u);
            if (todo1) atomicAdd(&h[1][dig], 1u);
        }
        __syncthreads();
        // pick the bin of each chain: inclusive scan of its 256 counts by 256 threads (4 waves)
        unsigned int cntv = 0, incl = 0, rank = 0;
        const int c = threadIdx.x >> 8, bin = threadIdx.x & 255;
        if (threadIdx.x < 512) {
            rank = s_rank[c];
            cntv = h[c][bin];
            incl = cntv;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (lane == 63) wsum[c][(threadIdx.x >> 6) & 3] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 512) {
            const int w = (threadIdx.x >> 6) & 3;
            for (int i = 0; i < w; ++i) incl += wsum[c][i];
            if (incl >= rank && incl - cntv < rank) {          // exactly one bin per chain (1 <= rank <= SMP_N)
                s_rank[c] = rank - (incl - cntv);
                s_prefix[c] |= ((unsigned long long)bin) << shift;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        BandRow br;
        br.lo = rank_lo < 1 ? 0ull : s_prefix[0];
        br.hi = rank_hi > SMP_N ? ~0ull : (s_prefix[1] | ((1ull << (64 - 8 * SMP_DIGITS)) - 1ull));
        band[row] = br;
    }
}

struct SelectWork {
    DBuf<SelState> st;
    DBuf<unsigned int> hist;
    DBuf<unsigned long long> candA, candB;
    int cap_rows = 0;
    int64_t cap_N = 0;
};

static int select_prepare(SelectWork &wk, int nrows, int64_t N)
{
    if (wk.cap_rows >= nrows && wk.cap_N >= N) return 0;
    TFX_TRY(wk.st.alloc(nrows));
    TFX_TRY(wk.hist.alloc((size_t)nrows * SEL_BINS));
    TFX_TRY(wk.candA.alloc((size_t)nrows * N));
    TFX_TRY(wk.candB.alloc((size_t)nrows * N));
    wk.cap_rows = nrows;
    wk.cap_N = N;
    return 0;
}

// thr[row] for nrows rows of N coefficients; K = entries to keep
int select_threshold_dev(tfx_ctx *ctx, SelectWork &wk, const double *d_rows, int nrows, int64_t N, int64_t K, double *d_thr)
{
    hipStream_t s = ctx->stream;
    if (K >= N) {                                            // "Taking all elements": thr = -1 -> floor 1e-30 (:244-256)
        std::vector<double> h((size_t)nrows, 1.e-30);
        TFX_HIP(hipMemcpyAsync(d_thr, h.data(), nrows * sizeof(double), hipMemcpyHostToDevice, s));
        TFX_HIP(hipStreamSynchronize(s));
        return 0;
    }
    TFX_TRY(select_prepare(wk, nrows, N));
    TFX_HIP(hipMemsetAsync(wk.hist.p, 0, (size_t)nrows * SEL_BINS * sizeof(unsigned int), s));
    hipLaunchKernelGGL(k_sel_init, dim3((nrows + 63) / 64), dim3(64), 0, s, wk.st.p, nrows, (unsigned long long)(N - K));
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->num_cu * 4 / std::max(1, nrows) + 1, (N + 255) / 256));
    unsigned long long *in = wk.candA.p, *out = wk.candB.p;
    for (int d = 0; d < SEL_NDIG; ++d) {
        hipLaunchKernelGGL(k_sel_hist, dim3(gx, nrows), dim3(256), 0, s, d_rows, N, in, wk.cap_N, wk.st.p, d, wk.hist.p, d == 0);
        hipLaunchKernelGGL(k_sel_pick, dim3(nrows), dim3(256), 0, s, wk.st.p, wk.hist.p, d);
        if (d + 1 < SEL_NDIG) {
            hipLaunchKernelGGL(k_sel_filter, dim3(gx, nrows), dim3(256), 0, s, d_rows, N, in, out, wk.cap_N, wk.st.p, d, d == 0);
            hipLaunchKernelGGL(k_sel_advance, dim3((nrows + 63) / 64), dim3(64), 0, s, wk.st.p, nrows);
            std::swap(in, out);
        }
    }
    hipLaunchKernelGGL(k_sel_result, dim3((nrows + 63) / 64), dim3(64), 0, s, wk.st.p, nrows, d_thr, (int *)nullptr);
    TFX_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================================
// ordered compaction: keep |c| > thr in ascending column order (sensitivity_gravmag.F90:258-272)
// =============================================================================================================
constexpr int CMP_THREADS = 256;
constexpr int CMP_PER_THREAD = 8;
constexpr int CMP_SEG = CMP_THREADS * CMP_PER_THREAD;     // 2048 elements per block
static_assert(CMP_PER_THREAD * (CMP_THREADS / 64) <= 64 && CMP_PER_THREAD % 2 == 0, "one wave scans the (sub-block, wave) counts");

struct CompactArgs {
    const double *rows;       // [nrows][N]
    int64_t N;
    const double *thr;        // [nrows]
    int keep_all;             // compression off: every column is stored (:289-295)
    int64_t col_begin, col_end;   // columns kept by this rank, output column = p - col_begin
    int nseg;                 // segments per row
    int cnt_segs;             // consecutive segments a block of the count pass walks
    int32_t *seg_cnt;         // [nrows][nseg] kept-in-range counts
    int32_t *seg_all;         // [nrows][nseg] kept over all columns
    int32_t *seg_off;         // [nrows][nseg] exclusive scan
    double *seg_cost;         // [nrows][nseg] sum of discarded^2
    int32_t *out_cols;        // [.. ][stride] (null: statistics only)
    float *out_vals;
    int64_t out_stride;
    int32_t *nel;             // [nrows] kept in range, per line
    int ncm;                  // lines per matrix row (model components): line r is component r % ncm of output row r / ncm,
    int64_t comp_stride;      //   appended after the earlier components with columns shifted by k*comp_stride (:829-846)
    int32_t *nel_all;         // [nrows] kept over all columns (the reference's nel)
    double *cost_disc;        // [nrows]
    const float *scale;       // [nrows] (float)(problem_weight*data_weight)   (:841)
    int32_t *hist;            // [N] per-column nnz (sensit_nnz, :267) or null
    // band select (null band: thr is final when the count pass runs)
    const BandRow *band;      // [nrows]
    double *slot_vals;        // [nrows][nseg][BAND_SLOTS] the coefficients of each segment that may be kept (|c| >= band.lo), in order
    uint16_t *slot_pos;       // ... and their position inside the segment
    int32_t *seg_slot;        // [nrows][nseg] how many
    int32_t *seg_band;        // [nrows][nseg] how many of them lie inside the band (<= band.hi)
    int32_t *seg_boff;        // [nrows][nseg] exclusive scan of seg_band
    unsigned long long *band_keys;    // [nrows][key_stride] dense keys for the select
    int64_t key_stride;
    SelState *st;
    unsigned long long K;
    int *fail;                // set when a band missed: the write pass leaves everything untouched, the batch is redone
};

__device__ __forceinline__ bool keep_elem(double v, double thr, int keep_all) { return keep_all || fabs(v) > thr; }

// exclusive scan of NCNT <= 64 (sub-block, wave) counts in LDS by the first wave; returns the total to every lane of that wave
// (call from wave 0 only, after a barrier)
template <int NCNT>
__device__ __forceinline__ int scan_wave0(int *wcnt, int lane)
{
    const int c = lane < NCNT ? wcnt[lane] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < NCNT; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane < NCNT) wcnt[lane] = incl - c;
    return __shfl(incl, NCNT - 1);
}

// Count pass.  A block walks CNT_SEGS consecutive segments of CMP_SEG = 2048 elements; thread t holds the elements
// k*512 + 2t, k*512 + 2t + 1 (k = 0..3) of a segment (16-byte loads where the row start allows) and has the loads of the next
// segment in flight while it counts this one.  Counts the kept elements of each segment.  With a band (threshold not known yet):
// counts the keys above band.hi and those inside the band, copies every coefficient that may be kept (key >= band.lo, ~2.5 % of
// the row) with its position into the segment's own slots in ascending position (no atomics) - the write pass then never reads
// the row again - and sums the energy of the rest.
constexpr int CNT_SEGS = 4;
constexpr int CNT_LOADS = CMP_PER_THREAD / 2;

__device__ __forceinline__ void cnt_load(const double *__restrict__ r, int64_t N, int64_t b, bool vec, double *dst)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < CNT_LOADS; ++k) {
        const int64_t p = b + (int64_t)k * (2 * CMP_THREADS);
        if (vec && p + 1 < N) {
            const d2 t = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(&r[p]));
            dst[2 * k] = t.x;
            dst[2 * k + 1] = t.y;
        } else {
            dst[2 * k] = p < N ? __builtin_nontemporal_load(&r[p]) : 0.0;
            dst[2 * k + 1] = p + 1 < N ? __builtin_nontemporal_load(&r[p + 1]) : 0.0;
        }
    }
}

__global__ __launch_bounds__(CMP_THREADS) void k_cmp_count(CompactArgs a)
{
    const int row = blockIdx.y;
    const int seg_begin = blockIdx.x * a.cnt_segs, seg_end = min(seg_begin + a.cnt_segs, a.nseg);
    const double *r = a.rows + (int64_t)row * a.N;
    const bool vec = (reinterpret_cast<uintptr_t>(r) & 15) == 0;
    const bool banded = a.band != nullptr;
    const double thr = banded ? 0.0 : a.thr[row];
    const unsigned long long lo = banded ? a.band[row].lo : 0ull, hi = banded ? a.band[row].hi : 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = CMP_THREADS / 64;
    // LDS by segment parity: a fast wave may start the next segment while a slow one still reads this one's scan
    __shared__ int wcnt[2][CNT_LOADS * NW];                 // slot count of (load k, wave w): slot order = (k, w, lane, half)
    __shared__ int s_cnt[2][NW], s_all[2][NW], s_band[2][NW];
    __shared__ double s_cost[2][NW];
    double v[CMP_PER_THREAD], nx[CMP_PER_THREAD];
    cnt_load(r, a.N, (int64_t)seg_begin * CMP_SEG + 2 * threadIdx.x, vec, v);
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int par = (seg - seg_begin) & 1;
        if (seg + 1 < seg_end) cnt_load(r, a.N, (int64_t)(seg + 1) * CMP_SEG + 2 * threadIdx.x, vec, nx);
        const int64_t base = (int64_t)seg * CMP_SEG + 2 * threadIdx.x;
        int cnt = 0, cnt_all = 0, nband = 0;
        int lpre[CMP_PER_THREAD];
        unsigned slotmask = 0;
        double cost = 0.0;
#pragma unroll
        for (int k = 0; k < CNT_LOADS; ++k) {
            bool sl[2] = {false, false};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t p = base + (int64_t)k * (2 * CMP_THREADS) + h;
                const double x = v[2 * k + h];
                if (p < a.N) {
                    bool keep;
                    if (banded) {
                        const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(x));
                        keep = key > hi;
                        sl[h] = key >= lo;
                        if (sl[h] && !keep) nband += 1;
                        if (!sl[h]) cost = fma(x, x, cost);
                    } else keep = keep_elem(x, thr, a.keep_all);
                    if (keep) {
                        cnt_all += 1;
                        if (p >= a.col_begin && p < a.col_end) cnt += 1;
                    }
                }
            }
            if (banded) {
                const unsigned long long m0 = __ballot(sl[0]), m1 = __ballot(sl[1]);
                const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0)) +
                                __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0));
                lpre[2 * k] = pre;
                lpre[2 * k + 1] = pre + (sl[0] ? 1 : 0);
                if (sl[0]) slotmask |= 1u << (2 * k);
                if (sl[1]) slotmask |= 1u << (2 * k + 1);
                if (lane == 0) wcnt[par][k * NW + wave] = __popcll(m0) + __popcll(m1);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            cnt += __shfl_down(cnt, d);
            cnt_all += __shfl_down(cnt_all, d);
            if (banded) { nband += __shfl_down(nband, d); cost += __shfl_down(cost, d); }
        }
        if (lane == 0) { s_cnt[par][wave] = cnt; s_all[par][wave] = cnt_all; s_band[par][wave] = nband; s_cost[par][wave] = cost; }
        __syncthreads();
        const int64_t sg = (int64_t)row * a.nseg + seg;
        if (wave == 0) {
            int run = 0;
            if (banded) run = scan_wave0<CNT_LOADS * NW>(wcnt[par], lane);
            if (lane == 0) {
                int c = 0, ca = 0, nb = 0;
                double cs = 0.0;
                for (int i = 0; i < NW; ++i) { c += s_cnt[par][i]; ca += s_all[par][i]; nb += s_band[par][i]; cs += s_cost[par][i]; }
                a.seg_cnt[sg] = c;
                a.seg_all[sg] = ca;
                if (banded) { a.seg_slot[sg] = run; a.seg_band[sg] = nb; a.seg_cost[sg] = cs; }
            }
        }
        if (banded) {
            __syncthreads();
            if (slotmask) {
                double *bv = a.slot_vals + sg * BAND_SLOTS;
                uint16_t *bp = a.slot_pos + sg * BAND_SLOTS;
#pragma unroll
                for (int j = 0; j < CMP_PER_THREAD; ++j)
                    if (slotmask & (1u << j)) {
                        const int pos = wcnt[par][(j >> 1) * NW + wave] + lpre[j];
                        bv[pos] = v[j];
                        bp[pos] = (uint16_t)((j >> 1) * (2 * CMP_THREADS) + 2 * threadIdx.x + (j & 1));
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < CMP_PER_THREAD; ++j) v[j] = nx[j];
    }
}

// one block per row, after the count pass: band size B and keys above (G); does the band hold the wanted order statistic
// (and did every segment fit its slots)?  Offsets of the segments' candidates in the dense key list; select state.
__global__ void k_band_scan(CompactArgs a)
{
    const int row = blockIdx.x;
    __shared__ long long part[256], gpart[256];
    __shared__ int s_over;
    const int per = (a.nseg + 255) / 256;
    const int b = threadIdx.x * per, e = min(b + per, a.nseg);
    if (threadIdx.x == 0) s_over = 0;
    __syncthreads();
    long long sb = 0, sgv = 0;
    for (int i = b; i < e; ++i) {
        sb += a.seg_band[(int64_t)row * a.nseg + i];
        sgv += a.seg_all[(int64_t)row * a.nseg + i];
    }
    part[threadIdx.x] = sb;
    gpart[threadIdx.x] = sgv;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0, G = 0;
        for (int i = 0; i < 256; ++i) { const long long v = part[i]; part[i] = run; run += v; G += gpart[i]; }
        const unsigned long long B = (unsigned long long)run, N = (unsigned long long)a.N;
        const unsigned long long below = N - (unsigned long long)G - B, r = N - a.K;    // r: 1-based ascending rank of the threshold (:240-250)
        const bool ok = !s_over && B <= (unsigned long long)a.key_stride && below < r && r <= below + B;
        // the exact select runs on key - lo: its first 12-bit digit splits the band's key range [0, hi - lo] into >= 2048 bins
        const unsigned long long span = a.band[row].hi - a.band[row].lo;
        a.st[row].prefix = 0;
        a.st[row].rank = ok ? r - below : 1;
        a.st[row].ncand = ok ? B : 0;
        a.st[row].nnext = 0;
        a.st[row].base = a.band[row].lo;
        a.st[row].bin = 0;
        a.st[row].top = span ? 64 - __clzll((long long)span) : 0;
        if (!ok) atomicOr(a.fail, 1);
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = b; i < e; ++i) {
        a.seg_boff[(int64_t)row * a.nseg + i] = (int32_t)min(run, (long long)INT32_MAX);
        run += a.seg_band[(int64_t)row * a.nseg + i];
    }
}

// band members of the slots -> dense key list of the row, stored minus band.lo (one wave per segment).  In the three slot kernels
// every load whose address does not depend on another load is issued before the first test (the first 64 slots speculatively: a
// segment owns BAND_SLOTS >= 64 of them whether it uses them or not): one memory latency per wave instead of a chain of four.
__global__ __launch_bounds__(256) void k_band_gather(CompactArgs a)
{
    const int row = blockIdx.y, seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= a.nseg) return;
    const int64_t sg = (int64_t)row * a.nseg + seg;
    const double *bv = a.slot_vals + sg * BAND_SLOTS;
    const double v0 = bv[lane];
    const int nb = a.seg_band[sg], n = a.seg_slot[sg], boff = a.seg_boff[sg];
    const unsigned long long lo = a.band[row].lo, hi = a.band[row].hi, ncand = a.st[row].ncand;
    if (ncand == 0 || nb == 0) return;
    unsigned long long *out = a.band_keys + (int64_t)row * a.key_stride + boff;
    int run = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        unsigned long long key = 0;
        bool in = false;
        if (i < n) {
            key = (unsigned long long)__double_as_longlong(fabs(i0 == 0 ? v0 : bv[i]));
            in = key <= hi;
        }
        const unsigned long long m = __ballot(in);
        if (in) out[run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = key - lo;
        run += __popcll(m);
    }
}

// band members above the final threshold join their segment's counts (one wave per segment, owner adds: no atomics)
__global__ __launch_bounds__(256) void k_band_fix(CompactArgs a)
{
    const int row = blockIdx.y, seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= a.nseg) return;
    const int64_t sg = (int64_t)row * a.nseg + seg;
    const double *bv = a.slot_vals + sg * BAND_SLOTS;
    const uint16_t *bp = a.slot_pos + sg * BAND_SLOTS;
    const double v0 = bv[lane];
    const uint16_t p0 = bp[lane];
    const int nb = a.seg_band[sg], n = a.seg_slot[sg];
    const double thr = a.thr[row];
    const unsigned long long hikey = a.band[row].hi;
    if (nb == 0) return;
    const double hi = __longlong_as_double((long long)hikey);                // NaN pattern when hi = all ones: no key is above it
    const bool hi_all = hikey == ~0ull;
    int cnt = 0, cnt_all = 0;
    for (int i = lane; i < n; i += 64) {
        const double av = fabs(i < 64 ? v0 : bv[i]);
        if ((hi_all || av <= hi) && av > thr) {                              // inside the band (not yet counted) and kept
            const int64_t p = (int64_t)seg * CMP_SEG + (i < 64 ? p0 : bp[i]);
            cnt_all += 1;
            if (p >= a.col_begin && p < a.col_end) cnt += 1;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { cnt += __shfl_down(cnt, d); cnt_all += __shfl_down(cnt_all, d); }
    if (lane == 0 && cnt_all) {
        a.seg_all[sg] += cnt_all;
        a.seg_cnt[sg] += cnt;
    }
}

// write pass of the band path: the kept entries come out of the segment's slots (ascending positions), the row itself is not
// read again; the band members below the threshold complete the discarded energy (one wave per segment)
__global__ __launch_bounds__(256) void k_slot_write(CompactArgs a)
{
    if (*a.fail) return;
    const int row = blockIdx.y, seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= a.nseg) return;
    const int64_t sg = (int64_t)row * a.nseg + seg;
    const double *bv = a.slot_vals + sg * BAND_SLOTS;
    const uint16_t *bp = a.slot_pos + sg * BAND_SLOTS;
    const double v0 = bv[lane];
    const uint16_t p0 = bp[lane];
    const int n = a.seg_slot[sg];
    const double thr = a.thr[row];
    const float sc = a.scale ? a.scale[row] : 1.0f;
    int segoff = a.seg_off[sg];
    if (n == 0) return;
    const int comp = row % a.ncm, mrow = row / a.ncm;
    for (int kk = 0; kk < comp; ++kk) segoff += a.nel[row - comp + kk];
    const int64_t cshift = (int64_t)comp * a.comp_stride - a.col_begin;
    int32_t *oc = a.out_cols ? a.out_cols + (int64_t)mrow * a.out_stride : nullptr;
    float *ov = a.out_cols ? a.out_vals + (int64_t)mrow * a.out_stride : nullptr;
    double cost = 0.0;
    int run = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        double v = 0.0;
        int64_t p = 0;
        bool keep = false;
        if (i < n) {
            v = i0 == 0 ? v0 : bv[i];
            p = (int64_t)seg * CMP_SEG + (i0 == 0 ? p0 : bp[i]);
            if (keep_elem(v, thr, 0)) {
                if (a.hist) atomicAdd(&a.hist[p], 1);
                keep = p >= a.col_begin && p < a.col_end;
            } else cost = fma(v, v, cost);
        }
        const unsigned long long m = __ballot(keep);
        if (keep && oc) {
            const int pos = segoff + run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            oc[pos] = (int32_t)(p + cshift);
            float f = (float)v;                              // real(x, MATRIX_PRECISION), :265
            if (a.scale) f = f * sc;                         // :841
            ov[pos] = f;
        }
        run += __popcll(m);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cost += __shfl_down(cost, d);
    if (lane == 0 && cost != 0.0) a.seg_cost[sg] += cost;
}

// one block per row: exclusive scan of the segment counts, totals
__global__ void k_cmp_scan(CompactArgs a)
{
    const int row = blockIdx.x;
    __shared__ int part[256], apart[256];
    const int per = (a.nseg + 255) / 256;
    const int b = threadIdx.x * per, e = min(b + per, a.nseg);
    int s = 0, sa = 0;
    for (int i = b; i < e; ++i) {
        s += a.seg_cnt[(int64_t)row * a.nseg + i];
        sa += a.seg_all[(int64_t)row * a.nseg + i];
    }
    part[threadIdx.x] = s;
    apart[threadIdx.x] = sa;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0, arun = 0;
        for (int i = 0; i < 256; ++i) { int v = part[i]; part[i] = run; run += v; arun += apart[i]; }
        a.nel[row] = run;
        a.nel_all[row] = arun;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = b; i < e; ++i) { a.seg_off[(int64_t)row * a.nseg + i] = run; run += a.seg_cnt[(int64_t)row * a.nseg + i]; }
}

// one block per row: discarded energy of the row = sum of the segment sums (fixed order)
__global__ void k_cmp_cost(CompactArgs a)
{
    if (a.fail && *a.fail) return;
    const int row = blockIdx.x;
    __shared__ double cpart[256];
    const int per = (a.nseg + 255) / 256;
    const int b = threadIdx.x * per, e = min(b + per, a.nseg);
    double cs = 0.0;
    for (int i = b; i < e; ++i) cs += a.seg_cost[(int64_t)row * a.nseg + i];
    cpart[threadIdx.x] = cs;
    __syncthreads();
    if (threadIdx.x == 0) {
        double crun = 0.0;
        for (int i = 0; i < 256; ++i) crun += cpart[i];
        a.cost_disc[row] = crun;
    }
}

// second pass over the row: writes the kept in-range entries at their final positions, adds the kept columns to the nnz
// histogram and sums the discarded energy of the segment (:258-283)
__global__ __launch_bounds__(CMP_THREADS) void k_cmp_write(CompactArgs a)
{
    if (a.fail && *a.fail) return;
    const int row = blockIdx.y, seg = blockIdx.x;
    const double *r = a.rows + (int64_t)row * a.N;
    const double thr = a.thr[row];
    const float sc = a.scale ? a.scale[row] : 1.0f;
    const int64_t base = (int64_t)seg * CMP_SEG + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = CMP_THREADS / 64;
    __shared__ int wcnt[CMP_PER_THREAD * NW];           // kept count of (sub-block k, wave w), column order = (k, w, lane)
    __shared__ double s_cost[NW];
    double v[CMP_PER_THREAD];
    int lpre[CMP_PER_THREAD];
    unsigned keepmask = 0;
    double cost = 0.0;
#pragma unroll
    for (int k = 0; k < CMP_PER_THREAD; ++k) {
        const int64_t p = base + (int64_t)k * CMP_THREADS;
        v[k] = 0.0;
        bool keep = false;
        if (p < a.N) {
            v[k] = r[p];
            if (keep_elem(v[k], thr, a.keep_all)) {
                if (a.hist) atomicAdd(&a.hist[p], 1);
                keep = p >= a.col_begin && p < a.col_end;
            } else cost = fma(v[k], v[k], cost);
        }
        const unsigned long long m = __ballot(keep);
        lpre[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (keep) keepmask |= 1u << k;
        if (lane == 0) wcnt[k * NW + wave] = __popcll(m);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cost += __shfl_down(cost, d);
    if (lane == 0) s_cost[wave] = cost;
    __syncthreads();
    if (wave == 0) {                                     // exclusive scan of the 32 (k, w) counts
        (void)scan_wave0<CMP_PER_THREAD * NW>(wcnt, lane);
        if (lane == 0) {
            double cs = 0.0;
            for (int i = 0; i < NW; ++i) cs += s_cost[i];
            a.seg_cost[(int64_t)row * a.nseg + seg] = cs;
        }
    }
    if (!a.out_cols) return;
    __syncthreads();
    int segoff = a.seg_off[(int64_t)row * a.nseg + seg];
    const int comp = row % a.ncm, mrow = row / a.ncm;
    for (int kk = 0; kk < comp; ++kk) segoff += a.nel[row - comp + kk];
    const int64_t cshift = (int64_t)comp * a.comp_stride - a.col_begin;
    int32_t *oc = a.out_cols + (int64_t)mrow * a.out_stride;
    float *ov = a.out_vals + (int64_t)mrow * a.out_stride;
#pragma unroll
    for (int k = 0; k < CMP_PER_THREAD; ++k)
        if (keepmask & (1u << k)) {
            const int pos = segoff + wcnt[k * NW + wave] + lpre[k];
            oc[pos] = (int32_t)(base + (int64_t)k * CMP_THREADS + cshift);
            float f = (float)v[k];                           // real(x, MATRIX_PRECISION), :265
            if (a.scale) f = f * sc;                         // :841
            ov[pos] = f;
        }
}

// entries of the matrix rows = sum over the model-component lines
__global__ void k_merge_nel(const int32_t *__restrict__ nel_sub, int ncm, int nrows, int32_t *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int n = 0;
    for (int k = 0; k < ncm; ++k) n += nel_sub[r * ncm + k];
    out[r] = n;
}

// dense store of a batch of rows: out[row][c] = (float)rows[row][col_begin + c] * scale[row]   (:289-295, :841)
// (line = blockIdx.y: component line % ncm of matrix row line / ncm, stored at column offset (line % ncm)*ncols)
__global__ __launch_bounds__(256) void k_dense_store(const double *__restrict__ rows, int64_t N, int64_t col_begin, int64_t ncols,
                                                     const float *__restrict__ scale, float *__restrict__ out, int64_t ld, int ncm)
{
    const int row = blockIdx.y;
    const double *r = rows + (int64_t)row * N + col_begin;
    float *o = out + (int64_t)(row / ncm) * ld + (int64_t)(row % ncm) * ncols;
    const float sc = scale[row];
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x) {
        float f = (float)r[c];
        f = f * sc;
        o[c] = f;
    }
}

__global__ void k_fill_i32(int32_t *__restrict__ p, int64_t n, int32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] += v;
}

// per-line statistics of a batch packed for one device -> host copy: [cost_full, cost_disc] doubles, then [nel_all, nel] ints, fail
struct BatchStat { double cost_full, cost_disc; int32_t nel_all, nel; };
__global__ void k_pack_stats(int n, const double *__restrict__ cost_full, const double *__restrict__ cost_disc,
                             const int32_t *__restrict__ nel_all, const int32_t *__restrict__ nel, const int *__restrict__ fail,
                             BatchStat *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        BatchStat b;
        b.cost_full = cost_full ? cost_full[i] : 0.0;
        b.cost_disc = cost_disc[i];
        b.nel_all = nel_all[i];
        b.nel = nel[i];
        out[i] = b;
    }
    if (i == 0) {                                  // the fail flag rides in an extra record
        BatchStat b;
        b.cost_full = b.cost_disc = 0.0;
        b.nel_all = *fail;
        b.nel = 0;
        out[n] = b;
    }
}

struct CompactWork {
    DBuf<int32_t> seg_cnt, seg_all, seg_off, nel, nel_all;
    DBuf<double> seg_cost, cost_disc, thr, red;
    DBuf<float> scale;
    int cap_rows = 0;
    int nseg = 0;
    // band select
    DBuf<BandRow> band;
    DBuf<double> slot_vals;
    DBuf<uint16_t> slot_pos;
    DBuf<int32_t> seg_slot, seg_band, seg_boff;
    DBuf<int> fail;
};

static int compact_prepare(CompactWork &cw, int nrows, int64_t N)
{
    const int nseg = (int)((N + CMP_SEG - 1) / CMP_SEG);
    if (cw.cap_rows >= nrows && cw.nseg == nseg) return 0;
    TFX_TRY(cw.seg_cnt.alloc((size_t)nrows * nseg));
    TFX_TRY(cw.seg_off.alloc((size_t)nrows * nseg));
    TFX_TRY(cw.seg_all.alloc((size_t)nrows * nseg));
    TFX_TRY(cw.seg_cost.alloc((size_t)nrows * nseg));
    TFX_TRY(cw.nel.alloc(nrows));
    TFX_TRY(cw.nel_all.alloc(nrows));
    TFX_TRY(cw.cost_disc.alloc(nrows));
    TFX_TRY(cw.thr.alloc(nrows));
    TFX_TRY(cw.scale.alloc(nrows));
    TFX_TRY(cw.red.alloc((size_t)nrows * 256));
    TFX_TRY(cw.fail.alloc(1));
    cw.cap_rows = nrows;
    cw.nseg = nseg;
    return 0;
}

static int band_prepare(CompactWork &cw, int nrows)
{
    const size_t nsl = (size_t)nrows * cw.nseg;
    if (cw.band.n >= (size_t)nrows && cw.seg_band.n >= nsl) return 0;
    TFX_TRY(cw.band.alloc(nrows));
    TFX_TRY(cw.slot_vals.alloc(nsl * BAND_SLOTS));
    TFX_TRY(cw.slot_pos.alloc(nsl * BAND_SLOTS));
    TFX_TRY(cw.seg_slot.alloc(nsl));
    TFX_TRY(cw.seg_band.alloc(nsl));
    TFX_TRY(cw.seg_boff.alloc(nsl));
    return 0;
}

// Sample ranks that bracket the (N-K)-th smallest of N with a miss probability of ~1e-5 per row (4.5 sigma of the
// binomial count of sample points below the threshold, + 2 for the rounding of the ranks).
static void band_sample_ranks(int64_t N, int64_t K, int *rank_lo, int *rank_hi)
{
    const double p = (double)(N - K) / (double)N;
    const double mid = p * SMP_N, m = 4.5 * std::sqrt(SMP_N * p * (1.0 - p)) + 2.0;
    *rank_lo = (int)std::floor(mid - m);
    *rank_hi = (int)std::ceil(mid + m) + 1;
}

// lines [nrows][N] (device) -> out_cols/out_vals matrix rows (stride; ncm consecutive lines form one row),
// d_nel_out[nrows/ncm] entries per matrix row; per-line nel / nel_all / cost_disc stay in cw.
// sel == null: cw.thr holds the thresholds.  sel != null (band select, K < N): the thresholds are found on the way - cw.thr
// is written, and *cw.fail != 0 afterwards means a band missed: nothing was written, redo the batch with the full select.
static int compact_dev(tfx_ctx *ctx, CompactWork &cw, const double *d_rows, int nrows, int64_t N, int keep_all,
                       int64_t col_begin, int64_t col_end, int32_t *out_cols, float *out_vals, int64_t out_stride,
                       int32_t *d_nel_out, const float *d_scale, int32_t *d_hist, int ncm = 1, SelectWork *sel = nullptr,
                       int64_t K = 0)
{
    hipStream_t s = ctx->stream;
    CompactArgs a{};
    a.rows = d_rows; a.N = N; a.thr = cw.thr.p; a.keep_all = keep_all; a.col_begin = col_begin; a.col_end = col_end;
    a.nseg = cw.nseg; a.seg_cnt = cw.seg_cnt.p; a.seg_all = cw.seg_all.p; a.seg_off = cw.seg_off.p; a.seg_cost = cw.seg_cost.p;
    a.out_cols = out_cols; a.out_vals = out_vals; a.out_stride = out_stride; a.nel = cw.nel.p; a.nel_all = cw.nel_all.p;
    a.cost_disc = cw.cost_disc.p; a.scale = d_scale; a.hist = d_hist; a.ncm = ncm; a.comp_stride = col_end - col_begin;
    TFX_HIP(hipMemsetAsync(cw.fail.p, 0, sizeof(int), s));
    a.cnt_segs = CNT_SEGS;
    const unsigned cnt_gx = (unsigned)((cw.nseg + a.cnt_segs - 1) / a.cnt_segs);
    if (sel) {
        TFX_TRY(select_prepare(*sel, nrows, N));
        TFX_TRY(band_prepare(cw, nrows));
        int rank_lo, rank_hi;
        band_sample_ranks(N, K, &rank_lo, &rank_hi);
        hipLaunchKernelGGL(k_sel_sample, dim3(nrows), dim3(SMP_THREADS), 0, s, d_rows, N, rank_lo, rank_hi, cw.band.p);
        a.band = cw.band.p; a.slot_vals = cw.slot_vals.p; a.slot_pos = cw.slot_pos.p; a.seg_slot = cw.seg_slot.p; a.seg_band = cw.seg_band.p;
        a.seg_boff = cw.seg_boff.p; a.band_keys = sel->candA.p; a.key_stride = sel->cap_N; a.st = sel->st.p;
        a.K = (unsigned long long)K; a.fail = cw.fail.p;
        hipLaunchKernelGGL(k_cmp_count, dim3(cnt_gx, nrows), dim3(CMP_THREADS), 0, s, a);
        hipLaunchKernelGGL(k_band_scan, dim3(nrows), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_band_gather, dim3((cw.nseg + 3) / 4, nrows), dim3(256), 0, s, a);
        // exact select inside the band (dense keys in candA)
        TFX_HIP(hipMemsetAsync(sel->hist.p, 0, (size_t)nrows * SEL_BINS * sizeof(unsigned int), s));
        const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->num_cu * 4 / std::max(1, nrows) + 1, (N / 64 + 255) / 256));
        unsigned long long *in = sel->candA.p, *out = sel->candB.p;
        // digit 0 over all blocks (it spreads the band's key range over >= 2048 bins), the rest by one block per row
        hipLaunchKernelGGL(k_sel_hist, dim3(gx, nrows), dim3(256), 0, s, d_rows, N, in, sel->cap_N, sel->st.p, 0, sel->hist.p, 0);
        hipLaunchKernelGGL(k_sel_pick, dim3(nrows), dim3(256), 0, s, sel->st.p, sel->hist.p, 0);
        hipLaunchKernelGGL(k_sel_filter, dim3(gx, nrows), dim3(256), 0, s, d_rows, N, in, out, sel->cap_N, sel->st.p, 0, 0);
        hipLaunchKernelGGL(k_sel_finish, dim3(nrows), dim3(256), 0, s, sel->st.p, out, sel->cap_N, 1);
        hipLaunchKernelGGL(k_sel_result, dim3((nrows + 63) / 64), dim3(64), 0, s, sel->st.p, nrows, cw.thr.p, cw.fail.p);
        hipLaunchKernelGGL(k_band_fix, dim3((cw.nseg + 3) / 4, nrows), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_cmp_count, dim3(cnt_gx, nrows), dim3(CMP_THREADS), 0, s, a);
    }
    hipLaunchKernelGGL(k_cmp_scan, dim3(nrows), dim3(256), 0, s, a);
    if (sel) hipLaunchKernelGGL(k_slot_write, dim3((cw.nseg + 3) / 4, nrows), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_cmp_write, dim3(cw.nseg, nrows), dim3(CMP_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_cmp_cost, dim3(nrows), dim3(256), 0, s, a);
    if (d_nel_out) hipLaunchKernelGGL(k_merge_nel, dim3((nrows / ncm + 63) / 64), dim3(64), 0, s, cw.nel.p, ncm, nrows / ncm, d_nel_out);
    TFX_HIP(hipGetLastError());
    return 0;
}

}  // namespace tfx

using namespace tfx;

extern "C" {

static int geometry_error(int herr)
{
    if (herr & 1) return fail(TFX_E_GEOMETRY, "Data coordinate coincides with model grid boundary (YZ). Adjust the model grid!");
    if (herr & 2) return fail(TFX_E_GEOMETRY, "Data coordinate coincides with model grid boundary (XZ). Adjust the model grid!");
    if (herr & 4) return fail(TFX_E_GEOMETRY, "The model grid X-boundary coincides with the data position");
    if (herr & 8) return fail(TFX_E_GEOMETRY, "The model grid Y-boundary coincides with the data position");
    if (herr & 16) return fail(TFX_E_GEOMETRY, "Zero denominator in gradiprism_full! Adjust the model grid.");      // gravity_field.f90:275-277
    if (herr & 32) return fail(TFX_E_GEOMETRY, "Bad log argument in gradiprism_full! Adjust the model grid.");      // :282-284
    if (herr & 64) return fail(TFX_E_GEOMETRY, "Data coordinate coincides with model grid boundary (XY). Adjust the model grid!");   // :102-104
    return 0;
}

static int make_rowgen(RowGen &gen, int problem_type, int data_type, int ncd, int ncm, const double *mag_field)
{
    gen = RowGen{};
    gen.ncd = ncd;
    gen.ncm = ncm;
    if (problem_type == 1) {
        if (ncm != 1) return fail(TFX_E_ARG, "gravity kernels have one model component");
        if (data_type == 1) {                                                                   // sensitivity_gravmag.F90:195-198
            // one component: graviprism_z (gravity_field.f90:131-195); three: graviprism_full (:41-126), rows X, Y, Z
            if (ncd == 1) gen.kind = GEN_GZ;
            else if (ncd == 3) gen.kind = GEN_G3;
            else return fail(TFX_E_ARG, "gravity data (type 1) has one (gz) or three (gx, gy, gz) data components");
        } else if (data_type == 2) {                                                            // :199-214
            if (ncd == 1) gen.kind = GEN_GZZ;
            else if (ncd == 6) gen.kind = GEN_FTG;
            else return fail(TFX_E_ARG, "Wrong number of gravity gradiometry data components!");
        } else return fail(TFX_E_ARG, "unknown gravity data type %d", data_type);
    } else if (problem_type == 2) {
        if (!mag_field) return fail(TFX_E_ARG, "magnetic field (incl, decl, azim, intensity) missing");
        if (!((ncm == 1 || ncm == 3) && (ncd == 1 || ncd == 3)))
            return fail(TFX_E_ARG, "Wrong number of components in magnetic_field_magprism!");   // magnetic_field.f90:258-282
        gen.kind = GEN_MAG;
        gen.mf = make_mag_field(mag_field[0], mag_field[1], mag_field[2], mag_field[3]);
    } else return fail(TFX_E_ARG, "problem_type must be 1 (grav) or 2 (magn)");
    return 0;
}

static int prism_rows_any(tfx_ctx *ctx, const RowGen &gen, int64_t ndata, const double *xd, const double *yd, const double *zd,
                          double *rows_out)
{
    if (!ctx || !xd || !yd || !zd || !rows_out) return fail(TFX_E_ARG, "tfx_prism_rows: null argument");
    if (ctx->N == 0) return fail(TFX_E_STATE, "tfx_prism_rows: set the grid first");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t N = ctx->N;
    const int nsub = gen.nsub();
    // observations per batch: <= 64 lines per launch, <= 2 GB of lines
    const int B = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ndata, PRISM_MAX_BATCH / nsub),
                                                              (int64_t)(1u << 28) / (N * nsub)));
    DBuf<double> dobs, drows;
    DBuf<int> derr;
    TFX_TRY(dobs.alloc((size_t)3 * B));
    TFX_TRY(drows.alloc((size_t)B * nsub * N));
    TFX_TRY(derr.alloc(1));
    TFX_HIP(hipMemsetAsync(derr.p, 0, sizeof(int), s));
    for (int64_t o0 = 0; o0 < ndata; o0 += B) {
        const int nb = (int)std::min<int64_t>(B, ndata - o0);
        TFX_HIP(hipMemcpyAsync(dobs.p, xd + o0, nb * sizeof(double), hipMemcpyDefault, s));
        TFX_HIP(hipMemcpyAsync(dobs.p + B, yd + o0, nb * sizeof(double), hipMemcpyDefault, s));
        TFX_HIP(hipMemcpyAsync(dobs.p + 2 * B, zd + o0, nb * sizeof(double), hipMemcpyDefault, s));
        TFX_TRY(prism_rows_dev(ctx, gen, nb, dobs.p, dobs.p + B, dobs.p + 2 * B, nullptr, drows.p, derr.p));
        TFX_HIP(hipMemcpyAsync(rows_out + o0 * nsub * N, drows.p, (size_t)nb * nsub * N * sizeof(double), hipMemcpyDefault, s));
        TFX_HIP(hipStreamSynchronize(s));
    }
    int herr = 0;
    TFX_HIP(hipMemcpy(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost));
    return geometry_error(herr);
}

int tfx_prism_rows_gz(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double *rows_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    return prism_rows_any(ctx, RowGen{}, ndata, xd, yd, zd, rows_out);
}

int tfx_prism_rows_mag(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double incl,
                       double decl, double azim, double intensity, double *rows_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    RowGen gen;
    gen.kind = GEN_MAG;
    gen.mf = make_mag_field(incl, decl, azim, intensity);
    return prism_rows_any(ctx, gen, ndata, xd, yd, zd, rows_out);
}

int tfx_prism_rows(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int nmodel_components, int64_t ndata,
                   const double *xd, const double *yd, const double *zd, const double *mag_field, double *rows_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    RowGen gen;
    TFX_TRY(make_rowgen(gen, problem_type, data_type, ndata_components, nmodel_components, mag_field));
    return prism_rows_any(ctx, gen, ndata, xd, yd, zd, rows_out);
}

int tfx_column_weight_type1(tfx_ctx *ctx, double power, double Z0, double multiplier, double *cw_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !cw_out) return fail(TFX_E_ARG, "tfx_column_weight_type1: null argument");
    if (ctx->N == 0) return fail(TFX_E_STATE, "tfx_column_weight_type1: set the grid first");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t N = ctx->N;
    DBuf<double> w;
    DBuf<unsigned long long> mx;
    DBuf<int> derr;
    TFX_TRY(w.alloc((size_t)N));
    TFX_TRY(mx.alloc(1));
    TFX_TRY(derr.alloc(1));
    TFX_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long), s));
    TFX_HIP(hipMemsetAsync(derr.p, 0, sizeof(int), s));
    const int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)ctx->num_cu * 8);
    hipLaunchKernelGGL(k_depth_weight, dim3(grid), dim3(256), 0, s, N, ctx->grid[0].p, ctx->grid[1].p, ctx->grid[2].p,
                       ctx->grid[3].p, ctx->grid[4].p, ctx->grid[5].p, power, Z0, w.p, mx.p, derr.p);
    hipLaunchKernelGGL(k_depth_weight_finish, dim3(grid), dim3(256), 0, s, N, w.p, mx.p, multiplier, derr.p);
    TFX_HIP(hipGetLastError());
    int herr = 0;
    TFX_HIP(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    if (herr & 1) return fail(TFX_E_NUMERIC, "Error: non-positive depth in calc_depth_weight_pixel!");
    if (herr & 2) return fail(TFX_E_NUMERIC, "Zero damping weight! Exiting.");
    TFX_TRY(copy_any(cw_out, w.p, (size_t)N * sizeof(double), s));
    return 0;
}

int tfx_column_weight_type2(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double power,
                            double beta, double multiplier, double *cw_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !cw_out || !xd || !yd || !zd) return fail(TFX_E_ARG, "tfx_column_weight_type2: null argument");
    if (ctx->N == 0) return fail(TFX_E_STATE, "tfx_column_weight_type2: set the grid first");
    if (ndata <= 0) return fail(TFX_E_ARG, "no data");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t N = ctx->N;
    DBuf<double> w, dobs;
    DBuf<unsigned long long> mx;
    DBuf<int> derr;
    TFX_TRY(w.alloc((size_t)N));
    TFX_TRY(dobs.alloc((size_t)3 * ndata));
    TFX_TRY(mx.alloc(1));
    TFX_TRY(derr.alloc(1));
    TFX_HIP(hipMemcpyAsync(dobs.p, xd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dobs.p + ndata, yd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dobs.p + 2 * ndata, zd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long), s));
    TFX_HIP(hipMemsetAsync(derr.p, 0, sizeof(int), s));
    const int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)ctx->num_cu * 16);
    hipLaunchKernelGGL(k_distance_weight, dim3(grid), dim3(256), 0, s, N, ctx->grid[0].p, ctx->grid[1].p, ctx->grid[2].p,
                       ctx->grid[3].p, ctx->grid[4].p, ctx->grid[5].p, ndata, dobs.p, dobs.p + ndata, dobs.p + 2 * ndata, power,
                       beta, w.p, mx.p);
    hipLaunchKernelGGL(k_depth_weight_finish, dim3(grid), dim3(256), 0, s, N, w.p, mx.p, multiplier, derr.p);
    TFX_HIP(hipGetLastError());
    int herr = 0;
    TFX_HIP(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    if (herr & 2) return fail(TFX_E_NUMERIC, "Zero damping weight! Exiting.");
    TFX_TRY(copy_any(cw_out, w.p, (size_t)N * sizeof(double), s));
    return 0;
}

int tfx_column_weight_type3(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd, double power,
                            double multiplier, double *cw_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !cw_out || !xd || !yd || !zd) return fail(TFX_E_ARG, "tfx_column_weight_type3: null argument");
    if (ctx->N == 0) return fail(TFX_E_STATE, "tfx_column_weight_type3: set the grid first");
    if (ndata <= 0) return fail(TFX_E_ARG, "no data");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t N = ctx->N;
    DBuf<double> w, dobs;
    DBuf<unsigned long long> mx;
    DBuf<int> derr;
    TFX_TRY(w.alloc((size_t)N));
    TFX_TRY(dobs.alloc((size_t)3 * ndata));
    TFX_TRY(mx.alloc(1));
    TFX_TRY(derr.alloc(1));
    TFX_HIP(hipMemcpyAsync(dobs.p, xd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dobs.p + ndata, yd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dobs.p + 2 * ndata, zd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long), s));
    TFX_HIP(hipMemsetAsync(derr.p, 0, sizeof(int), s));
    const int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)ctx->num_cu * 16);
    hipLaunchKernelGGL(k_mindist_weight, dim3(grid), dim3(256), 0, s, N, ctx->grid[0].p, ctx->grid[1].p, ctx->grid[2].p, ctx->grid[3].p,
                       ctx->grid[4].p, ctx->grid[5].p, ndata, dobs.p, dobs.p + ndata, dobs.p + 2 * ndata, power, w.p, mx.p);
    hipLaunchKernelGGL(k_depth_weight_finish, dim3(grid), dim3(256), 0, s, N, w.p, mx.p, multiplier, derr.p);
    TFX_HIP(hipGetLastError());
    int herr = 0;
    TFX_HIP(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    if (herr & 2) return fail(TFX_E_NUMERIC, "Zero damping weight! Exiting.");
    TFX_TRY(copy_any(cw_out, w.p, (size_t)N * sizeof(double), s));
    return 0;
}

// diagnostics: the device build of fastmath.h on host arrays (tests compare it with the host libm)
__global__ void k_fastmath_eval(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out_log,
                                double *__restrict__ out_atan2)
{
    tfx::init_math_tables();
    __syncthreads();
    const tfx::FastMathTables tb = tfx::math_tables();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        out_log[i] = tfx::fast_log(a[i], tb);
        out_atan2[i] = tfx::fast_atan2(a[i], b[i], tb);
    }
}

int tfx_fastmath_eval(tfx_ctx *ctx, int64_t n, const double *a, const double *b, double *out_log, double *out_atan2)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    using namespace tfx;
    if (!ctx || !a || !b || !out_log || !out_atan2 || n < 0) return fail(TFX_E_ARG, "tfx_fastmath_eval: bad argument");
    if (n == 0) return 0;
    TFX_HIP(hipSetDevice(ctx->device));
    DBuf<double> da, db, dl, dt;
    TFX_TRY(da.alloc((size_t)n));
    TFX_TRY(db.alloc((size_t)n));
    TFX_TRY(dl.alloc((size_t)n));
    TFX_TRY(dt.alloc((size_t)n));
    TFX_TRY(copy_any(da.p, a, (size_t)n * sizeof(double), ctx->stream));
    TFX_TRY(copy_any(db.p, b, (size_t)n * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_fastmath_eval, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, ctx->stream, n, da.p, db.p, dl.p, dt.p);
    TFX_HIP(hipGetLastError());
    TFX_TRY(copy_any(out_log, dl.p, (size_t)n * sizeof(double), ctx->stream));
    TFX_TRY(copy_any(out_atan2, dt.p, (size_t)n * sizeof(double), ctx->stream));
    return 0;
}

int tfx_wavelet(tfx_ctx *ctx, double *sarr, int n1, int n2, int n3, int64_t nvec, int type, int direction)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !sarr) return fail(TFX_E_ARG, "tfx_wavelet: null argument");
    if (n1 <= 0 || n2 <= 0 || n3 <= 0 || nvec < 0) return fail(TFX_E_ARG, "tfx_wavelet: bad size");
    TFX_HIP(hipSetDevice(ctx->device));
    const int64_t N = (int64_t)n1 * n2 * n3;
    hipPointerAttribute_t attr;
    bool dev = hipPointerGetAttributes(&attr, sarr) == hipSuccess && attr.type == hipMemoryTypeDevice;
    if (!dev) (void)hipGetLastError();
    if (dev) {
        TFX_TRY(wavelet_dev(ctx, sarr, n1, n2, n3, nvec, type, direction));
        TFX_HIP(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    DBuf<double> d;
    TFX_TRY(d.alloc((size_t)(N * nvec)));
    TFX_TRY(copy_any(d.p, sarr, (size_t)(N * nvec) * sizeof(double), ctx->stream));
    TFX_TRY(wavelet_dev(ctx, d.p, n1, n2, n3, nvec, type, direction));
    TFX_TRY(copy_any(sarr, d.p, (size_t)(N * nvec) * sizeof(double), ctx->stream));
    return 0;
}

int tfx_compress_row(tfx_ctx *ctx, const double *row, int64_t N, int64_t K, int32_t *cols_out, float *vals_out,
                     int64_t *nel_out, double *thr_out, double *cost_discarded_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !row || !cols_out || !vals_out || !nel_out) return fail(TFX_E_ARG, "tfx_compress_row: null argument");
    if (N <= 0 || K < 0) return fail(TFX_E_ARG, "tfx_compress_row: bad size");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DBuf<double> d;
    DBuf<int32_t> oc;
    DBuf<float> ov;
    TFX_TRY(d.alloc((size_t)N));
    const int64_t stride = std::max<int64_t>(1, std::min<int64_t>(N, K));
    TFX_TRY(oc.alloc((size_t)stride));
    TFX_TRY(ov.alloc((size_t)stride));
    TFX_TRY(copy_any(d.p, row, (size_t)N * sizeof(double), s));
    SelectWork sw;
    CompactWork cw;
    TFX_TRY(compact_prepare(cw, 1, N));
    // same threshold path as the build: band select for large rows, the full select otherwise or when the band missed
    const bool banded = K > 0 && K < N && N >= ctx->band_min_n;
    int h_fail = 0;
    if (banded) {
        TFX_TRY(compact_dev(ctx, cw, d.p, 1, N, 0, 0, N, oc.p, ov.p, stride, nullptr, nullptr, nullptr, 1, &sw, K));
        TFX_HIP(hipMemcpyAsync(&h_fail, cw.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
        TFX_HIP(hipStreamSynchronize(s));
        ctx->band_batches += 1;
        if (h_fail) ctx->band_fallbacks += 1;
    }
    if (!banded || h_fail) {
        TFX_TRY(select_threshold_dev(ctx, sw, d.p, 1, N, K, cw.thr.p));
        TFX_TRY(compact_dev(ctx, cw, d.p, 1, N, 0, 0, N, oc.p, ov.p, stride, nullptr, nullptr, nullptr));
    }
    int32_t nel = 0;
    double thr = 0, cd = 0;
    TFX_HIP(hipMemcpyAsync(&nel, cw.nel.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipMemcpyAsync(&thr, cw.thr.p, sizeof(double), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipMemcpyAsync(&cd, cw.cost_disc.p, sizeof(double), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    if (nel > stride) return fail(TFX_E_NUMERIC, "Wrong number of elements in calculate_and_write_sensit!");   // :275-277
    std::vector<int32_t> hc((size_t)nel);
    TFX_TRY(copy_any(hc.data(), oc.p, (size_t)nel * sizeof(int32_t), s));
    for (int32_t i = 0; i < nel; ++i) cols_out[i] = hc[(size_t)i] + 1;     // 1-based across the ABI
    TFX_TRY(copy_any(vals_out, ov.p, (size_t)nel * sizeof(float), s));
    *nel_out = nel;
    if (thr_out) *thr_out = thr;
    if (cost_discarded_out) *cost_discarded_out = cd;
    return 0;
}

// General build.  One observation yields nsub = ncd*ncm LINES (d outer, k inner - the record order of the reference's
// SENSIT files, sensitivity_gravmag.F90:222-311); every line is weighted, transformed, thresholded and compacted on its
// own.  Matrix row (i*ncd + d) is the concatenation of its ncm lines with columns k*ncols + (cell - col_begin)
// (read_sensitivity_kernel, :829-852).  data_weight: [ndata*ncd], d fastest (data_weight(d, i)).
static int build_kernel_any(tfx_ctx *ctx, const RowGen &gen, int64_t ndata, const double *xd, const double *yd, const double *zd,
                            const double *column_weight, int compression_type, double rate, double problem_weight,
                            const double *data_weight, int64_t col_begin, int64_t col_end, int64_t *nnz_out,
                            double *error_sum_out, int32_t *nnz_hist_out, RowStore *rs = nullptr)
{
    // rs != null: keep the compressed rows (all columns, global 0-based column indices) row-major on the device instead of
    // laying them out as this rank's tiled matrix - the row-parallel half of the multi-GPU build
    const bool to_rs = rs != nullptr;
    // tfx_matrix_reserve arms ONE build: consumed here, before any exit, so that a build that fails on its arguments, a row-store build or
    // a counting pass (no columns kept) can not leave it armed for a later, larger build; a build into the other slot ignores it
    int64_t reserved = 0;
    if (ctx) {
        if (ctx->reserve_nnz > 0 && ctx->reserve_slot == ctx->slot) reserved = ctx->reserve_nnz;
        ctx->reserve_nnz = 0;
    }
    if (to_rs) { col_begin = 0; col_end = ctx ? ctx->N : 0; }
    if (to_rs && compression_type == 0) return fail(TFX_E_ARG, "the row store is for compressed kernels (dense kernels are built per column range)");
    if (!ctx || !xd || !yd || !zd || !column_weight) return fail(TFX_E_ARG, "tfx_build_kernel: null argument");
    if (ctx->N == 0) return fail(TFX_E_STATE, "tfx_build_kernel: set the grid first");
    if (compression_type < 0 || compression_type > 2) return fail(TFX_E_ARG, "Unknown wavelet type!");
    if (rate < 0 || rate > 1) return fail(TFX_E_ARG, "Wrong compression rate! It must be between 0 and 1.");   // :112-114
    const int64_t N = ctx->N;
    if (col_begin < 0 || col_end > N || col_begin > col_end) return fail(TFX_E_ARG, "bad column range");
    if (ndata <= 0) return fail(TFX_E_ARG, "no data");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // TFX_BUILD_TIMING=1: wall-clock phases of the build on stderr (host clock; the loop time includes its per-batch waits)
    static const bool timing = getenv("TFX_BUILD_TIMING") != nullptr;
    const auto wall = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = wall();
    double t_wait = 0.0, t_append = 0.0, t_lap = t_enter;
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = wall();
        fprintf(stderr, "[tfx] build setup: %-28s %.3f s\n", what, t - t_lap);
        t_lap = t;
    };
    const int ncd = gen.ncd, ncm = gen.ncm, nsub = gen.nsub();
    const int64_t ncols = col_end - col_begin;            // cells kept; the matrix has ncm*ncols columns
    const int64_t nrows_m = ndata * ncd;                  // matrix rows
    const int64_t nlines = ndata * nsub;
    const bool keep_matrix = ncols > 0 && !to_rs;
    const int64_t K = compression_type > 0 ? (int64_t)(rate * (double)N) : N;      // get_nel_compressed, :64-77
    const int64_t lstride = std::max<int64_t>(1, std::min<int64_t>(K, std::max<int64_t>(ncols, 1)));   // entries per line, at most
    const int64_t stride = lstride * ncm;                 // ... per matrix row
    // observation coordinates and scale factors on the device
    DBuf<double> dobs, dcw, drows, dred;
    DBuf<int> derr;
    DBuf<int32_t> dhist;
    TFX_TRY(dobs.alloc((size_t)3 * ndata));
    TFX_TRY(dcw.alloc((size_t)N));
    TFX_TRY(derr.alloc(1));
    TFX_HIP(hipMemcpyAsync(dobs.p, xd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dobs.p + ndata, yd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dobs.p + 2 * ndata, zd, ndata * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemcpyAsync(dcw.p, column_weight, (size_t)N * sizeof(double), hipMemcpyDefault, s));
    TFX_HIP(hipMemsetAsync(derr.p, 0, sizeof(int), s));
    if (nnz_hist_out) {
        TFX_TRY(dhist.alloc((size_t)N));
        TFX_HIP(hipMemsetAsync(dhist.p, 0, (size_t)N * sizeof(int32_t), s));
    }
    std::vector<float> hscale((size_t)nlines);            // per line: (float)(problem_weight * data_weight(d, i)), :838
    {
        std::vector<double> hdw;
        if (data_weight) {
            hdw.resize((size_t)nrows_m);
            TFX_TRY(copy_any(hdw.data(), data_weight, (size_t)nrows_m * sizeof(double), s));
        }
        for (int64_t r = 0; r < nrows_m; ++r) {
            const float sc = (float)(problem_weight * (data_weight ? hdw[(size_t)r] : 1.0));
            for (int k = 0; k < ncm; ++k) hscale[(size_t)(r * ncm + k)] = sc;
        }
    }
    DBuf<float> dscale;
    TFX_TRY(dscale.alloc((size_t)nlines));
    TFX_HIP(hipMemcpyAsync(dscale.p, hscale.data(), (size_t)nlines * sizeof(float), hipMemcpyHostToDevice, s));
    lap("uploads queued");
    // observations per batch: the line buffer + select candidates (2x) stay around 6 GB, at most 32 lines (one observation
    // at least)
    // (TFX_LINES_CAP / TFX_LINES_BYTES_LOG2: tuning knobs for the batch size)
    static const int64_t cap_lines = getenv("TFX_LINES_CAP") ? std::max(1, std::min(PRISM_MAX_BATCH, atoi(getenv("TFX_LINES_CAP")))) : 32;
    static const int cap_log2 = getenv("TFX_LINES_BYTES_LOG2") ? std::max(20, std::min(34, atoi(getenv("TFX_LINES_BYTES_LOG2")))) - 3 : 28;
    const int64_t lines_cap = std::max<int64_t>(1, std::min<int64_t>(cap_lines, ((int64_t)1 << cap_log2) / N));
    const int ob_max = (int)std::max<int64_t>(1, lines_cap / nsub);
    TiledMatrix &m = ctx->selmat();
    ctx->target = &m;
    if (keep_matrix && compression_type == 0) {
        // No compression (sensitivity_gravmag.F90:287-295): every column is stored -> dense fp32 block, no index stream.
        TFX_TRY(matrix_begin_dense(ctx, nrows_m, ncm * ncols));
        TFX_TRY(drows.alloc((size_t)ob_max * nsub * N));
        const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (ncols + 255) / 256));
        for (int64_t g = 0; g < ndata; g += ob_max) {
            const int nb = (int)std::min<int64_t>(ob_max, ndata - g);
            TFX_TRY(prism_rows_dev(ctx, gen, nb, dobs.p + g, dobs.p + ndata + g, dobs.p + 2 * ndata + g, dcw.p, drows.p, derr.p));
            hipLaunchKernelGGL(k_dense_store, dim3(gx, nb * nsub), dim3(256), 0, s, drows.p, N, col_begin, ncols, dscale.p + g * nsub,
                               m.dense.p + g * ncd * m.ld, m.ld, ncm);
            TFX_HIP(hipGetLastError());
        }
        int herr = 0;
        TFX_HIP(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, s));
        TFX_HIP(hipStreamSynchronize(s));
        TFX_TRY(geometry_error(herr));
        TFX_TRY(matrix_finish(ctx));
        if (nnz_out) *nnz_out = nrows_m * ncm * ncols;
        if (error_sum_out) *error_sum_out = 0.0;
        if (nnz_hist_out) {
            hipLaunchKernelGGL(k_fill_i32, dim3(1024), dim3(256), 0, s, dhist.p, N, (int32_t)nlines);       // :291-294: every column, every line
            TFX_HIP(hipGetLastError());
            TFX_TRY(copy_any(nnz_hist_out, dhist.p, (size_t)N * sizeof(int32_t), s));
        }
        return 0;
    }
    if (keep_matrix) {
        // (a caller that knows how many entries the column range will hold - it has the per-column histogram of a counting pass - says so
        // with tfx_matrix_reserve: a range of a rank-partitioned kernel holds 1 / P of the rows x K bound used otherwise)
        int64_t upper = nrows_m * stride;
        if (reserved > 0) upper = std::min(upper, reserved);
        TFX_TRY(matrix_begin(ctx, nrows_m, ncm * ncols, upper));
        lap("matrix_begin");
    }
    const int RB = keep_matrix ? m.RB : (int)std::min<int64_t>(RB_MAX, (nrows_m + 63) / 64 * 64);
    TFX_TRY(drows.alloc((size_t)ob_max * nsub * N));
    const int npart = prism_partials(ctx, gen);
    const int lines_max = ob_max * nsub;
    DBuf<double> dcf;
    TFX_TRY(dred.alloc((size_t)lines_max * npart));
    TFX_TRY(dcf.alloc((size_t)2 * lines_max));        // cost_full of the lines of a batch: one set per statistics slot (a redone batch needs its own)
    // staging area of finished matrix rows: a row block (RB rows) plus the rows of one observation that may straddle it
    const int stage_rows = RB + ncd;
    DBuf<int32_t> ell_cols, ell_nel;
    DBuf<float> ell_vals;
    DBuf<int64_t> ell_off;
    if (keep_matrix) {
        TFX_TRY(ell_cols.alloc((size_t)stage_rows * stride));
        TFX_TRY(ell_vals.alloc((size_t)stage_rows * stride));
        TFX_TRY(ell_off.alloc(stage_rows));
        std::vector<int64_t> ho(stage_rows);
        for (int r = 0; r < stage_rows; ++r) ho[r] = (int64_t)r * stride;
        TFX_HIP(hipMemcpyAsync(ell_off.p, ho.data(), stage_rows * sizeof(int64_t), hipMemcpyHostToDevice, s));
    }
    TFX_TRY(ell_nel.alloc(stage_rows));
    if (to_rs) {
        rs->nrows = nrows_m;
        rs->stride = std::max<int64_t>(1, K) * ncm;
        rs->ncm = ncm;
        rs->N = N;
        TFX_TRY(rs->cols.alloc((size_t)(nrows_m * rs->stride)));
        TFX_TRY(rs->vals.alloc((size_t)(nrows_m * rs->stride)));
        TFX_TRY(rs->nel.alloc((size_t)nrows_m));
    }
    lap("row / staging buffers");
    SelectWork sw;
    CompactWork cw;
    TFX_TRY(compact_prepare(cw, lines_max, N));
    lap("compaction work areas");
    double err_sum = 0.0;
    int64_t nnz_total = 0;
    DBuf<BatchStat> dstat;
    TFX_TRY(dstat.alloc(lines_max + 1));
    // pinned, two of them: the statistics of batch b are read one batch later, when batch b + 1 is already queued (the GPU never waits
    // for the host's round trip: 3852 batches x ~0.2 ms at the headline size), so its copy must not land on top of batch b's
    BatchStat *h_stat_all = nullptr;
    TFX_HIP(hipHostMalloc((void **)&h_stat_all, (size_t)2 * (lines_max + 1) * sizeof(BatchStat)));
    struct PinnedFree { void *p; ~PinnedFree() { (void)hipHostFree(p); } } h_stat_guard{h_stat_all};
    BatchStat *h_stat2[2] = {h_stat_all, h_stat_all + (lines_max + 1)};
    struct StatEvents {
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~StatEvents() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } se;
    for (int i = 0; i < 2; ++i) TFX_HIP(hipEventCreateWithFlags(&se.ev[i], hipEventDisableTiming));
    int fill = 0;                 // finished rows waiting in the staging area
    int64_t r0 = 0;               // first matrix row of the staging area
    bool band_off = false;
    const int64_t batches0 = ctx->band_batches, fallbacks0 = ctx->band_fallbacks;
    // Overlapped build: the row generator is VALU-bound (fp64 log / atan2 / sqrt per grid node), everything after it - wavelet
    // passes, threshold, compaction, tile scatter - is HBM-bound.  The generator of batch b + 1 runs on its own (low-priority) stream
    // into the next row buffer, queued behind the wavelet passes of batch b (the two exclude each other through the LDS); the
    // threshold / compaction chain of batch b runs on a third stream beside the wavelet passes of batch b + 1 (below).
    // (ctx->build_overlap = 0: one stream, one buffer - the sequential order of round 1.)
    // (short builds: the extra buffers and the stream cost more than they hide; build_overlap = 2 forces the overlapped form: tests)
    const bool overlap = ctx->build_overlap == 2 || (ctx->build_overlap && ndata >= 8 * (int64_t)ob_max);
    // Three row buffers in overlap mode: batch b + 1 is generated while batch b is transformed, and batch b - 1's transformed rows stay
    // until its statistics are confirmed (a band that missed redoes the batch from them with the full select)
    const int nbuf = overlap ? 3 : 1;
    DBuf<double> drows2, dred2, drows3, dred3;
    double *rows_buf[3] = {drows.p, drows.p, drows.p}, *red_buf[3] = {dred.p, dred.p, dred.p};
    struct GenStream {
        hipStream_t st = nullptr;
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr}, ev0 = nullptr;
        ~GenStream()
        {
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
            for (hipEvent_t e : {ev[0], ev[1], ev[2], ev0}) if (e) (void)hipEventDestroy(e);
        }
    } gs;
    if (overlap) {
        TFX_TRY(drows2.alloc((size_t)ob_max * nsub * N));
        TFX_TRY(dred2.alloc((size_t)lines_max * npart));
        TFX_TRY(drows3.alloc((size_t)ob_max * nsub * N));
        TFX_TRY(dred3.alloc((size_t)lines_max * npart));
        rows_buf[1] = drows2.p;
        red_buf[1] = dred2.p;
        rows_buf[2] = drows3.p;
        red_buf[2] = dred3.p;
        // lowest priority where the runtime offers priorities (the main stream's kernels get the freed slots first); a plain
        // non-blocking stream otherwise
        int prio_lo = 0, prio_hi = 0;
        // (the generator on a CU-masked stream of its own - 64 / 96 / 128 / 160 of the 256 CUs, started with the batch - was measured in
        // round 5: 27.9 / 24.9 / 21.4 / 21.0 s of loop against 18.9 s, profiles/r05_gen_cumask_probe.txt; not kept)
        if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
            hipStreamCreateWithPriority(&gs.st, hipStreamNonBlocking, prio_lo) != hipSuccess) {
            (void)hipGetLastError();
            gs.st = nullptr;
            TFX_HIP(hipStreamCreateWithFlags(&gs.st, hipStreamNonBlocking));
        }
        for (int i = 0; i < 3; ++i) TFX_HIP(hipEventCreateWithFlags(&gs.ev[i], hipEventDisableTiming));
        TFX_HIP(hipEventCreateWithFlags(&gs.ev0, hipEventDisableTiming));
        TFX_HIP(hipEventRecord(gs.ev0, s));                 // the uploads of the observations / weights queued above
        TFX_HIP(hipStreamWaitEvent(gs.st, gs.ev0, 0));
    }
    // Third stream: the generator holds every VGPR of the SIMDs it runs on, so nothing runs beside it; the wavelet passes leave
    // registers, wave slots and 21 KB of LDS per CU free.  The threshold / compaction chain of batch b (one read of the rows, then
    // latency-bound work on the few candidates) therefore waits for the generator of batch b + 1 and runs beside the wavelet passes
    // of batch b + 1: per batch the device does generator, then (wavelet || chain).  Ordering between the streams: the chain waits
    // for its batch's wavelet event (recorded on the main stream, i.e. behind every earlier append of staged rows as well) and for
    // the next generator's event; the main stream meets the chain only through the host (confirm waits for the statistics event).
    struct ChainStream {
        hipStream_t st = nullptr;
        hipEvent_t evw[3] = {nullptr, nullptr, nullptr};
        ~ChainStream()
        {
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
            for (hipEvent_t e : evw) if (e) (void)hipEventDestroy(e);
        }
    } chs;
    const bool chain_late = overlap && compression_type > 0 && ctx->chain_under_wavelet;
    if (chain_late) {
        TFX_HIP(hipStreamCreateWithFlags(&chs.st, hipStreamNonBlocking));
        for (int i = 0; i < 3; ++i) TFX_HIP(hipEventCreateWithFlags(&chs.evw[i], hipEventDisableTiming));
    }
    hipStream_t const chain_s = chain_late ? chs.st : s;
    struct StreamSwap {           // the compaction helpers launch on ctx->stream
        tfx_ctx *c;
        hipStream_t keep;
        StreamSwap(tfx_ctx *c_, hipStream_t t) : c(c_), keep(c_->stream) { c->stream = t; }
        ~StreamSwap() { c->stream = keep; }
    };
    // observations of the batch that starts at g_ with fill_ rows staged: just enough to complete the current row block (so that
    // with one data component blocks never straddle)
    auto batch_obs = [&](int64_t g_, int fill_) -> int {
        const int64_t want = ((int64_t)RB - fill_ + ncd - 1) / ncd;
        return (int)std::min<int64_t>(std::min<int64_t>(ob_max, std::max<int64_t>(1, want)), ndata - g_);
    };
    auto generate = [&](int64_t g_, int nb_, int slot_) -> int {
        hipStream_t keep = ctx->stream;
        if (overlap) ctx->stream = gs.st;
        ctx->gen_grid_limit = ctx->gen_wgs_per_cu > 0 ? ctx->gen_wgs_per_cu * ctx->num_cu : 0;      // (debug key: resident generator workgroups per CU)
        const int rc = prism_rows_dev(ctx, gen, nb_, dobs.p + g_, dobs.p + ndata + g_, dobs.p + 2 * ndata + g_, dcw.p, rows_buf[slot_], derr.p,
                                      compression_type > 0 ? red_buf[slot_] : nullptr);
        ctx->stream = keep;
        ctx->gen_grid_limit = 0;
        TFX_TRY(rc);
        if (overlap) TFX_HIP(hipEventRecord(gs.ev[slot_], gs.st));
        return 0;
    };
    lap("streams, pinned stats");
    const double t_loop = wall();
    // A batch in flight: queued on the GPU, its statistics not yet read
    struct Pending {
        bool active = false, banded = false;
        int64_t g = 0;
        int nb = 0, nl = 0, slot = 0, fill_at = 0, hs = 0;
    };
    // threshold (full select when asked, else already known / bracketed by the band) + compaction + the statistics copy of a batch
    auto compact_batch = [&](const Pending &b, bool full_select) -> int {
        double *const rows = rows_buf[b.slot];
        StreamSwap on_chain(ctx, chain_s);
        if (full_select) TFX_TRY(select_threshold_dev(ctx, sw, rows, b.nl, N, K, cw.thr.p));              // :240-256
        SelectWork *sel = (b.banded && !full_select) ? &sw : nullptr;
        if (to_rs)
            TFX_TRY(compact_dev(ctx, cw, rows, b.nl, N, 0, 0, N, rs->cols.p + (size_t)(b.g * ncd) * rs->stride,
                                rs->vals.p + (size_t)(b.g * ncd) * rs->stride, rs->stride, rs->nel.p + b.g * ncd, dscale.p + b.g * nsub,
                                nnz_hist_out ? dhist.p : nullptr, ncm, sel, K));
        else
            TFX_TRY(compact_dev(ctx, cw, rows, b.nl, N, compression_type == 0, col_begin, col_end,
                                keep_matrix ? ell_cols.p + (size_t)b.fill_at * stride : nullptr,
                                keep_matrix ? ell_vals.p + (size_t)b.fill_at * stride : nullptr, stride, ell_nel.p + b.fill_at,
                                dscale.p + b.g * nsub, nnz_hist_out ? dhist.p : nullptr, ncm, sel, K));
        // per-line statistics
        hipLaunchKernelGGL(k_pack_stats, dim3((b.nl + 63) / 64), dim3(64), 0, chain_s, b.nl, compression_type > 0 ? dcf.p + (size_t)b.hs * lines_max : nullptr,
                           cw.cost_disc.p, cw.nel_all.p, cw.nel.p, cw.fail.p, dstat.p);
        TFX_HIP(hipMemcpyAsync(h_stat2[b.hs], dstat.p, (size_t)(b.nl + 1) * sizeof(BatchStat), hipMemcpyDeviceToHost, chain_s));
        TFX_HIP(hipEventRecord(se.ev[b.hs], chain_s));
        return 0;
    };
    // reads a batch's statistics (waits for them); a band that missed redoes the batch from its transformed rows with the full
    // radix select - out of order behind whatever has been queued since: the rows of a batch have their own place in the staging area
    auto confirm = [&](Pending &b) -> int {
        if (!b.active) return 0;
        const BatchStat *hs = h_stat2[b.hs];
        const double tw = timing ? wall() : 0.0;
        TFX_HIP(hipEventSynchronize(se.ev[b.hs]));
        if (b.banded) {
            ctx->band_batches += 1;
            if (hs[b.nl].nel_all) {
                ctx->band_fallbacks += 1;
                TFX_TRY(compact_batch(b, true));
                TFX_HIP(hipEventSynchronize(se.ev[b.hs]));
            }
            // a sample that keeps missing (rows the pseudo-random positions do not represent): stop trying
            if (ctx->band_batches - batches0 >= 8 && 4 * (ctx->band_fallbacks - fallbacks0) > ctx->band_batches - batches0) band_off = true;
        }
        if (timing) t_wait += wall() - tw;
        for (int i = 0; i < b.nl; ++i) {
            if (hs[i].nel_all > K) return fail(TFX_E_NUMERIC, "Wrong number of elements in calculate_and_write_sensit!");   // :275-277
            if (compression_type > 0) err_sum += std::sqrt(hs[i].cost_disc / hs[i].cost_full);                        // :283
            nnz_total += hs[i].nel;
        }
        b.active = false;
        return 0;
    };
    int slot = 0;
    int nb_cur = batch_obs(0, 0);
    int64_t nbatch = 0;
    Pending pend;
    if (overlap) TFX_TRY(generate(0, nb_cur, 0));
    for (int64_t g = 0; g < ndata;) {
        Pending cur;
        cur.active = true;
        cur.g = g;
        cur.nb = nb_cur;
        cur.nl = nb_cur * nsub;                                     // lines of this batch
        cur.slot = slot;
        cur.fill_at = fill;
        cur.hs = (int)(nbatch & 1);
        const int nb = cur.nb, nl = cur.nl;
        const int slot_next = (slot + 1) % nbuf;
        double *const rows_cur = rows_buf[slot], *const red_cur = red_buf[slot];
        int64_t g_after = g + nb;
        if (overlap) {
            int fill_after = fill + nb * ncd;
            if (fill_after >= RB || g_after >= ndata) fill_after %= RB;
            nb_cur = g_after < ndata ? batch_obs(g_after, fill_after) : 0;
            TFX_HIP(hipStreamWaitEvent(s, gs.ev[slot], 0));
        } else {
            TFX_TRY(generate(g, nb, 0));
        }
        // The next batch's generator is queued behind this batch's wavelet passes (ctx->gen_after_wavelet, the default): it then
        // shares the GPU with the count / select / compaction kernels - pure HBM streams, the complement of its fp64 arithmetic -
        // and runs alone for the rest; beside the wavelet kernels (65 % VALU-busy themselves) both only slow each other down.
        // Its buffer was last read by the compaction of batch b - 2, whose statistics the host has confirmed.
        auto queue_next = [&]() -> int {
            if (!overlap || nb_cur <= 0) return 0;
            if (ctx->gen_after_wavelet > 0) {
                TFX_HIP(hipEventRecord(gs.ev0, s));
                TFX_HIP(hipStreamWaitEvent(gs.st, gs.ev0, 0));
            }
            return generate(g_after, nb_cur, slot_next);
        };
        const int gen_at = compression_type == 0 ? 0 : std::max(0, std::min(3, ctx->gen_after_wavelet));
        if (gen_at == 0) TFX_TRY(queue_next());
        // threshold: bracketed from a sample and finished inside the compaction's count pass (band select) for large rows,
        // else the full radix select up front
        cur.banded = compression_type > 0 && K < N && K > 0 && N >= ctx->band_min_n && !band_off;
        if (compression_type > 0) {
            hipLaunchKernelGGL(k_rows_final_sum, dim3(nl), dim3(256), 0, s, red_cur, npart, dcf.p + (size_t)cur.hs * lines_max);   // cost_full :234
            TFX_HIP(hipGetLastError());
            TFX_TRY(wavelet_dev(ctx, rows_cur, ctx->nx, ctx->ny, ctx->nz, nl, compression_type, 1, 0, gen_at));       // :237
            if (gen_at > 0) TFX_TRY(queue_next());
            TFX_TRY(wavelet_dev(ctx, rows_cur, ctx->nx, ctx->ny, ctx->nz, nl, compression_type, 1, gen_at, 3));
        }
        if (chain_late) {
            TFX_HIP(hipEventRecord(chs.evw[slot], s));
            TFX_HIP(hipStreamWaitEvent(chs.st, chs.evw[slot], 0));
            if (nb_cur > 0) TFX_HIP(hipStreamWaitEvent(chs.st, gs.ev[slot_next], 0));
        }
        TFX_TRY(compact_batch(cur, compression_type > 0 && !cur.banded));
        // the previous batch ran while this one was being queued: its statistics are there (or nearly)
        TFX_TRY(confirm(pend));
        g += nb;
        fill += nb * ncd;
        nbatch += 1;
        if (fill >= RB || g >= ndata || !overlap) {
            // a row block is complete (or the build is): everything staged must be final before it is laid out as tiles
            TFX_TRY(confirm(cur));
            int herr = 0;
            TFX_HIP(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, s));
            TFX_HIP(hipStreamSynchronize(s));
            TFX_TRY(geometry_error(herr));
            while (fill >= RB || (g >= ndata && fill > 0)) {
                const int nr = std::min(fill, RB);
                const double ta = timing ? wall() : 0.0;
                if (keep_matrix) TFX_TRY(matrix_append_rows(ctx, r0, nr, ell_cols.p, ell_vals.p, ell_nel.p, ell_off.p, stride));
                if (timing) t_append += wall() - ta;
                r0 += nr;
                const int left = fill - nr;
                if (left > 0) {        // rows of the straddling observation move to the front (left < ncd <= nr: no overlap)
                    if (keep_matrix) {
                        TFX_HIP(hipMemcpyAsync(ell_cols.p, ell_cols.p + (size_t)nr * stride, (size_t)left * stride * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
                        TFX_HIP(hipMemcpyAsync(ell_vals.p, ell_vals.p + (size_t)nr * stride, (size_t)left * stride * sizeof(float), hipMemcpyDeviceToDevice, s));
                    }
                    TFX_HIP(hipMemcpyAsync(ell_nel.p, ell_nel.p + nr, (size_t)left * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
                }
                fill = left;
            }
        } else {
            pend = cur;
        }
        if (overlap) slot = slot_next;
        else if (g < ndata) nb_cur = batch_obs(g, fill);
    }
    const double t_fin = wall();
    if (keep_matrix) {
        TFX_TRY(matrix_finish(ctx));
        m.nnz = nnz_total;
    }
    if (timing)
        fprintf(stderr, "[tfx] build timing: setup %.2f s (allocations, uploads, matrix_begin), loop %.2f s (waiting on the GPU %.2f s, appends %.2f s), "
                        "finish %.2f s\n", t_loop - t_enter, t_fin - t_loop, t_wait, t_append, wall() - t_fin);
    if (nnz_out) *nnz_out = nnz_total;
    if (error_sum_out) *error_sum_out = err_sum;
    if (nnz_hist_out) TFX_TRY(copy_any(nnz_hist_out, dhist.p, (size_t)N * sizeof(int32_t), s));
    return 0;
}

// ---- row store: the row-parallel half of the multi-GPU build -------------------------------------------------
int tfx_rowstore_build(tfx_ctx *ctx, int problem_type, int64_t ndata, const double *xd, const double *yd, const double *zd,
                       const double *column_weight, const double *mag_field, int compression_type, double rate,
                       double problem_weight, const double *data_weight, int64_t *nnz_out, double *error_sum_out,
                       int32_t *nnz_hist_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    RowGen gen;
    TFX_TRY(make_rowgen(gen, problem_type, 1, 1, 1, mag_field));
    return build_kernel_any(ctx, gen, ndata, xd, yd, zd, column_weight, compression_type, rate, problem_weight, data_weight, 0, 0,
                            nnz_out, error_sum_out, nnz_hist_out, &ctx->rowstore());
}

// the same with any data type / data components the reference supports (one model component)
int tfx_rowstore_build_ex(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int64_t ndata, const double *xd,
                          const double *yd, const double *zd, const double *column_weight, const double *mag_field,
                          int compression_type, double rate, double problem_weight, const double *data_weight, int64_t *nnz_out,
                          double *error_sum_out, int32_t *nnz_hist_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    RowGen gen;
    TFX_TRY(make_rowgen(gen, problem_type, data_type, ndata_components, 1, mag_field));
    return build_kernel_any(ctx, gen, ndata, xd, yd, zd, column_weight, compression_type, rate, problem_weight, data_weight, 0, 0,
                            nnz_out, error_sum_out, nnz_hist_out, &ctx->rowstore());
}

int tfx_rowstore_build_comp(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int nmodel_components, int64_t ndata,
                            const double *xd, const double *yd, const double *zd, const double *column_weight, const double *mag_field,
                            int compression_type, double rate, double problem_weight, const double *data_weight, int64_t *nnz_out,
                            double *error_sum_out, int32_t *nnz_hist_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    RowGen gen;
    TFX_TRY(make_rowgen(gen, problem_type, data_type, ndata_components, nmodel_components, mag_field));
    return build_kernel_any(ctx, gen, ndata, xd, yd, zd, column_weight, compression_type, rate, problem_weight, data_weight, 0, 0,
                            nnz_out, error_sum_out, nnz_hist_out, &ctx->rowstore());
}

// counts[r*nparts + d] = entries of row r whose CELL lies in [bounds[d], bounds[d+1]), summed over the model components
// (component k holds its cells at columns k*N + cell)
__device__ __forceinline__ int rs_lower_bound(const int32_t *c, int n, int64_t key)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)c[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void k_rs_bounds(const int32_t *__restrict__ cols, const int32_t *__restrict__ nel, int64_t stride, int64_t nrows,
                            const int64_t *__restrict__ bounds, int nparts, int ncm, int64_t N, int32_t *__restrict__ counts)
{
    const int64_t r = blockIdx.x;
    const int32_t *c = cols + r * stride;
    const int n = nel[r];
    for (int d = threadIdx.x; d < nparts; d += blockDim.x) {
        int cnt = 0;
        for (int k = 0; k < ncm; ++k)
            cnt += rs_lower_bound(c, n, bounds[d + 1] + k * N) - rs_lower_bound(c, n, bounds[d] + k * N);
        counts[r * nparts + d] = cnt;
    }
}

int tfx_rowstore_counts(tfx_ctx *ctx, int nparts, const int64_t *bounds, int32_t *counts_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !bounds || !counts_out) return fail(TFX_E_ARG, "tfx_rowstore_counts: null argument");
    RowStore &rs = ctx->rowstore();
    if (rs.nrows == 0) return fail(TFX_E_STATE, "no row store");
    if (nparts < 1 || nparts > 1024) return fail(TFX_E_ARG, "nparts out of range");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DBuf<int64_t> db;
    DBuf<int32_t> dc;
    TFX_TRY(db.alloc(nparts + 1));
    TFX_TRY(dc.alloc((size_t)(rs.nrows * nparts)));
    TFX_HIP(hipMemcpyAsync(db.p, bounds, (nparts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_rs_bounds, dim3((unsigned)rs.nrows), dim3(64), 0, s, rs.cols.p, rs.nel.p, rs.stride, rs.nrows, db.p, nparts, rs.ncm, rs.N, dc.p);
    TFX_HIP(hipGetLastError());
    TFX_TRY(copy_any(counts_out, dc.p, (size_t)(rs.nrows * nparts) * sizeof(int32_t), s));
    return 0;
}

// cells [col_begin, col_end) of rows [row_begin, row_begin + nrows), one piece per (row, model component): offsets by an
// exclusive scan over the pieces (row-major, component inside), then a copy
__global__ void k_rs_seg(const int32_t *__restrict__ cols, const int32_t *__restrict__ nel, int64_t stride, int64_t row_begin,
                         int nrows, int64_t col_begin, int64_t col_end, int ncm, int64_t N, int32_t *__restrict__ p0,
                         int32_t *__restrict__ cnt)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nrows * ncm) return;
    const int r = q / ncm, k = q - r * ncm;
    const int32_t *c = cols + (row_begin + r) * stride;
    const int n = nel[row_begin + r];
    const int a = rs_lower_bound(c, n, col_begin + k * N);
    p0[q] = a;
    cnt[q] = rs_lower_bound(c, n, col_end + k * N) - a;
}

__global__ void k_rs_scan(const int32_t *__restrict__ cnt, int nrows, int64_t *__restrict__ off)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t run = 0;
        for (int r = 0; r < nrows; ++r) { off[r] = run; run += cnt[r]; }
        off[nrows] = run;
    }
}

__global__ void k_rs_copy(const int32_t *__restrict__ cols, const float *__restrict__ vals, int64_t stride, int64_t row_begin,
                          const int32_t *__restrict__ p0, const int32_t *__restrict__ cnt, const int64_t *__restrict__ off,
                          int64_t col_begin, int64_t col_end, int ncm, int64_t N, int32_t *__restrict__ ocols, float *__restrict__ ovals)
{
    const int q = blockIdx.y;                       // piece (row, component)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt[q]) return;
    const int r = q / ncm, k = q - r * ncm;
    const int64_t src = (row_begin + r) * stride + p0[q] + j;
    // the owner's columns: component k at k*(col_end - col_begin) + (cell - col_begin)
    ocols[off[q] + j] = (int32_t)(cols[src] - k * N - col_begin + k * (col_end - col_begin));
    ovals[off[q] + j] = vals[src];
}

int tfx_rowstore_pack(tfx_ctx *ctx, int64_t row_begin, int64_t nrows, int64_t col_begin, int64_t col_end, int32_t *cols_dev_out,
                      float *vals_dev_out, int64_t capacity, int64_t *n_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !n_out) return fail(TFX_E_ARG, "tfx_rowstore_pack: null argument");
    RowStore &rs = ctx->rowstore();
    if (rs.nrows == 0) return fail(TFX_E_STATE, "no row store");
    if (row_begin < 0 || nrows <= 0 || row_begin + nrows > rs.nrows) return fail(TFX_E_ARG, "row range outside the row store");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int nr = (int)nrows, np = nr * rs.ncm;      // pieces
    DBuf<int32_t> p0, cnt;
    DBuf<int64_t> off;
    TFX_TRY(p0.alloc(np));
    TFX_TRY(cnt.alloc(np));
    TFX_TRY(off.alloc(np + 1));
    hipLaunchKernelGGL(k_rs_seg, dim3((np + 255) / 256), dim3(256), 0, s, rs.cols.p, rs.nel.p, rs.stride, row_begin, nr, col_begin,
                       col_end, rs.ncm, rs.N, p0.p, cnt.p);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(64), 0, s, cnt.p, np, off.p);
    int64_t total = 0;
    TFX_HIP(hipMemcpyAsync(&total, off.p + np, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    *n_out = total;
    if (total > capacity) return fail(TFX_E_ARG, "tfx_rowstore_pack: %lld entries do not fit the buffer (%lld)", (long long)total, (long long)capacity);
    if (total > 0) {
        if (!cols_dev_out || !vals_dev_out) return fail(TFX_E_ARG, "null output buffer");
        hipLaunchKernelGGL(k_rs_copy, dim3((unsigned)((rs.stride / rs.ncm + 255) / 256), np), dim3(256), 0, s, rs.cols.p, rs.vals.p,
                           rs.stride, row_begin, p0.p, cnt.p, off.p, col_begin, col_end, rs.ncm, rs.N, cols_dev_out, vals_dev_out);
        TFX_HIP(hipGetLastError());
    }
    TFX_HIP(hipStreamSynchronize(s));
    return 0;
}

int tfx_rowstore_free(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    (void)hipStreamSynchronize(ctx->stream);
    ctx->rowstore().cols.release();
    ctx->rowstore().vals.release();
    ctx->rowstore().nel.release();
    ctx->rowstore().nrows = 0;
    return 0;
}

// ---- assembling this rank's matrix from row pieces that arrive from other ranks ------------------------------------
int tfx_matrix_begin(tfx_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz_upper)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    TFX_HIP(hipSetDevice(ctx->device));
    ctx->target = &ctx->selmat();
    return matrix_begin(ctx, nrows, ncols, nnz_upper);
}

// rows [row_begin, row_begin + nr) (row_begin a multiple of the row-block size 2048, nr <= 2048), consecutive in the DEVICE
// buffers cols_dev (0-based local columns, ascending within a row) / vals_dev; nel_host[r] = entries of row r
int tfx_matrix_append_rows(tfx_ctx *ctx, int64_t row_begin, int64_t nr, const int32_t *cols_dev, const float *vals_dev,
                           const int32_t *nel_host)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !nel_host) return fail(TFX_E_ARG, "tfx_matrix_append_rows: null argument");
    TFX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<int64_t> ho((size_t)nr);
    int64_t run = 0, maxlen = 0;
    for (int64_t r = 0; r < nr; ++r) { ho[(size_t)r] = run; run += nel_host[r]; maxlen = std::max<int64_t>(maxlen, nel_host[r]); }
    DBuf<int32_t> dn;
    DBuf<int64_t> dof;
    TFX_TRY(dn.alloc((size_t)nr));
    TFX_TRY(dof.alloc((size_t)nr));
    TFX_HIP(hipMemcpyAsync(dn.p, nel_host, (size_t)nr * sizeof(int32_t), hipMemcpyHostToDevice, s));
    TFX_HIP(hipMemcpyAsync(dof.p, ho.data(), (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, s));
    ctx->target = &ctx->selmat();
    // the contract is blocks of up to RB_MAX rows starting at a multiple of RB_MAX; the matrix's own row block may be a smaller
    // power of two (small matrices get smaller tiles)
    if (row_begin % RB_MAX != 0 || nr > RB_MAX || nr <= 0)
        return fail(TFX_E_ARG, "tfx_matrix_append_rows: rows [%lld, +%lld) are not a block of up to %d rows at a multiple of %d",
                    (long long)row_begin, (long long)nr, RB_MAX, RB_MAX);
    const int rbm = ctx->selmat().RB;
    for (int64_t r0 = 0; r0 < nr; r0 += rbm)
        TFX_TRY(matrix_append_rows(ctx, row_begin + r0, std::min<int64_t>(rbm, nr - r0), cols_dev, vals_dev, dn.p + r0, dof.p + r0, maxlen));
    ctx->selmat().nnz += run;
    return 0;
}

int tfx_matrix_finish(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    ctx->target = &ctx->selmat();
    const int64_t nnz = ctx->selmat().nnz;
    TFX_TRY(matrix_finish(ctx));
    ctx->selmat().nnz = nnz;
    return 0;
}

int tfx_build_kernel_grav(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd,
                          const double *column_weight, int compression_type, double rate, double problem_weight,
                          const double *data_weight, int64_t col_begin, int64_t col_end, int64_t *nnz_out,
                          double *error_sum_out, int32_t *nnz_hist_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    return build_kernel_any(ctx, RowGen{}, ndata, xd, yd, zd, column_weight, compression_type, rate, problem_weight, data_weight,
                            col_begin, col_end, nnz_out, error_sum_out, nnz_hist_out);
}

int tfx_build_kernel_mag(tfx_ctx *ctx, int64_t ndata, const double *xd, const double *yd, const double *zd,
                         const double *column_weight, double incl, double decl, double azim, double intensity,
                         int compression_type, double rate, double problem_weight, const double *data_weight,
                         int64_t col_begin, int64_t col_end, int64_t *nnz_out, double *error_sum_out, int32_t *nnz_hist_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    RowGen gen;
    gen.kind = GEN_MAG;
    gen.mf = make_mag_field(incl, decl, azim, intensity);
    return build_kernel_any(ctx, gen, ndata, xd, yd, zd, column_weight, compression_type, rate, problem_weight, data_weight,
                            col_begin, col_end, nnz_out, error_sum_out, nnz_hist_out);
}

int tfx_build_kernel(tfx_ctx *ctx, int problem_type, int data_type, int ndata_components, int nmodel_components, int64_t ndata,
                     const double *xd, const double *yd, const double *zd, const double *column_weight, const double *mag_field,
                     int compression_type, double rate, double problem_weight, const double *data_weight, int64_t col_begin,
                     int64_t col_end, int64_t *nnz_out, double *error_sum_out, int32_t *nnz_hist_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    RowGen gen;
    TFX_TRY(make_rowgen(gen, problem_type, data_type, ndata_components, nmodel_components, mag_field));
    return build_kernel_any(ctx, gen, ndata, xd, yd, zd, column_weight, compression_type, rate, problem_weight, data_weight,
                            col_begin, col_end, nnz_out, error_sum_out, nnz_hist_out);
}

}  // extern "C"
