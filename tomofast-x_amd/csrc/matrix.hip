// matrix.hip - device-resident sensitivity matrix in the column-tiled compressed layout, and the two products
// that dominate an LSQR iteration:
//   forward  b (+)= S x    (t_sparse_matrix%add_mult_vector,       src/inversion/sparse_matrix.f90:316-329)
//   adjoint  b (+)= S^T x  (t_sparse_matrix%add_trans_mult_vector, src/inversion/sparse_matrix.f90:391-405)
//
// Layout (DESIGN.md "Data layout in HBM").  The reference keeps CSR with 4-byte values + 4-byte columns
// (8 B per non-zero).  Here S is cut into tiles of (RB <= 2048 rows) x (TC <= 4096 columns).  Inside a tile the
// entries are in (row, column) order, padded to chunks of 512 entries (64 lanes x 8 entries), in three streams:
//     vals[]    float  : the value exactly as the reference stores it (inside a chunk in val_pos() order)     4      B / entry
//     slots[]   12 bit : LDS slot of the column inside the tile (col_slot(column), common.h); a lane's 8 slots are
//                        three consecutive dwords, so a wave fetches a chunk's slots with one dwordx3 load   1.5    B / entry
//     rowmask[] 1 bit  : "first entry of a new row inside this tile", stored k-major: word k of a chunk holds the
//                        flags of entry k of all 64 lanes - exactly the wave mask a ballot would produce, so the
//                        kernels read it with scalar loads and use it as mbcnt operand / exec mask directly   0.125  B / entry
// i.e. 5.625 B per non-zero.  A row that is empty inside a tile but lies between two non-empty rows carries one marker
// entry (flag set, value 0).  chunk_row0[] gives each chunk the local row of the entry preceding it, so any wave can
// start at any chunk.
//
// Both products stream every tile once, coalesced, and keep the vector side of the product in LDS:
//   forward: the x tile (TC doubles, 32 KB) is staged in LDS once for up to FWD_GROUP_MAX row blocks (a "super block"),
//            whose row sums are accumulated in LDS (2 x RB doubles, 32 KB);
//   adjoint: by default the forward kernel on a transposed copy of the tiles (matrix_build_transpose); without the copy the u rows of
//            the block are staged in LDS and the column sums are accumulated there exactly, in 64-bit integers (k_spmv_adj).
// Row membership comes from the row-start masks with mbcnt prefix counts; short row segments are
// summed inside a lane, the segment tails are merged across lanes with one segmented wave reduction per chunk.
// Both products give the same bits on every run (DESIGN.md 4 "Reproducibility").
#include "common.h"
#include <algorithm>
#include <chrono>
#include <numeric>

namespace tfx {

thread_local std::string g_last_error;
thread_local tfx_ctx *g_alloc_ctx = nullptr;
thread_local int g_no_evict = 0;

bool evict_adjoint_copies(tfx_ctx *ctx)
{
    bool freed = false;
    (void)hipSetDevice(ctx->device);
    for (TiledMatrix *m : {&ctx->mat, &ctx->mat2, &ctx->cons}) {
        if (m->pre && ctx->adj_copy != 1 && ctx->pre_take != m->pre.get()) {      // storage set aside for a copy that does not exist yet
            m->drop_prealloc();
            freed = true;
        }
        // (never the object an allocation is being made FOR: a copy whose work lists matrix_finish is rebuilding is ctx->target)
        if (m->T && m->T->evictable && m->T != ctx->target) {
            if (!freed) (void)hipDeviceSynchronize();      // (nothing may still be reading a copy)
            const size_t bytes = m->T->device_bytes();
            delete m->T;
            m->T = nullptr;
            m->vmax_stale = true;
            freed = true;
            fprintf(stderr, "[tfx] out of device memory: gave up the transposed copy of a matrix (%.1f GB); its adjoint runs on the tiles of S\n",
                    (double)bytes / 1e9);
        }
    }
    return freed;
}

size_t TiledMatrix::device_bytes() const
{
    return rec.bytes() + chunk_row0.bytes() + tiles.bytes() + fwd.bytes() + adj.bytes() +
           fwd_order.bytes() + adj_order.bytes() + fwd_partial.bytes() + adj_partial.bytes() + adj_nslots.bytes() +
           adj_pbase.bytes() + fwd_nslots.bytes() + fwd_pbase.bytes() + dense.bytes() + dense_partial.bytes() + tile_bound.bytes() +
           (T ? T->device_bytes() : 0);
}

void TiledMatrix::release_storage()
{
    rec.release(); chunk_row0.release(); tiles.release(); fwd.release(); adj.release();
    fwd_order.release(); adj_order.release(); fwd_partial.release(); adj_partial.release();
    fwd_nslots.release(); fwd_pbase.release(); adj_nslots.release(); adj_pbase.release();
    dense.release(); dense_partial.release(); tile_bound.release();
    vmax_stale = true;
    drop_prealloc();
    delete T;
    T = nullptr;
    h_tiles.clear(); h_fwd.clear(); h_adj.clear();
    is_dense = false;
    n_entries = cap_entries = 0;
    valid = false;
}

int copy_any(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return 0;
    TFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
    TFX_HIP(hipStreamSynchronize(s));
    return 0;
}

// LDS index swizzle: a lane reads 8 consecutive entries, so for dense column runs lanes L and L+4 would hit the
// same bank pair (stride 8 doubles).  XOR-ing bits 3..5 into bits 0..2 makes 32 consecutive lanes conflict-free
// and is a bijection inside every aligned group of 64 indices (TC is a multiple of 64).

// ------------------------------------------------------------------------------------------------------------
// Conversion: row block in ELL form (cols ascending, 0-based local) -> tiles
// ------------------------------------------------------------------------------------------------------------

// pos[r][t] = index of the first entry of row r with column >= t*TC  (t = 0..ntc), one wave per row.
__global__ void k_tile_pos(const int32_t *__restrict__ cols, const int32_t *__restrict__ nel,
                           const int64_t *__restrict__ rowoff, int nr, int ntc, int TC, int32_t *__restrict__ pos)
{
    int r = blockIdx.x;
    if (r >= nr) return;
    const int32_t *c = cols + rowoff[r];
    int n = nel[r];
    for (int t = threadIdx.x; t <= ntc; t += blockDim.x) {
        int64_t key = (int64_t)t * TC;
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((int64_t)c[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        pos[(int64_t)r * (ntc + 1) + t] = lo;
    }
}

// One block per column tile: segment lengths (with empty-row markers) and their exclusive scan over the rows.
// segoff[t][r] (r = 0..nr), first/last non-empty row.
__global__ void k_tile_scan(const int32_t *__restrict__ pos, int nr, int ntc, int32_t *__restrict__ segoff,
                            int32_t *__restrict__ first_ne, int32_t *__restrict__ last_ne)
{
    extern __shared__ int32_t sm[];     // nr + 1 ints + 2
    int t = blockIdx.x;
    __shared__ int s_first, s_last;
    if (threadIdx.x == 0) { s_first = 0x7fffffff; s_last = -1; }
    __syncthreads();
    for (int r = threadIdx.x; r < nr; r += blockDim.x) {
        int cnt = pos[(int64_t)r * (ntc + 1) + t + 1] - pos[(int64_t)r * (ntc + 1) + t];
        sm[r] = cnt;
        if (cnt > 0) { atomicMin(&s_first, r); atomicMax(&s_last, r); }
    }
    __syncthreads();
    int f = s_first, l = s_last;
    // serial-per-thread blocked scan (nr <= 2048, blockDim = 256 -> 8 per thread)
    int per = (nr + blockDim.x - 1) / blockDim.x;
    int b = threadIdx.x * per, e = min(b + per, nr);
    int sum = 0;
    for (int r = b; r < e; ++r) {
        int len = sm[r];
        if (len == 0 && r > f && r < l) len = 1;      // marker for an empty row between non-empty ones
        sm[r] = len;
        sum += len;
    }
    __shared__ int part[1024];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { int v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    int32_t *so = segoff + (int64_t)t * (nr + 1);
    for (int r = b; r < e; ++r) { so[r] = run; run += sm[r]; }
    if (e == nr && b < nr) so[nr] = run;
    if (nr == 0 && threadIdx.x == 0) so[0] = 0;
    if (threadIdx.x == 0) { first_ne[t] = (l >= 0) ? f : -1; last_ne[t] = l; }
}

// One entry into the three streams (the destination range was zeroed: bits are OR-ed in, neighbours belong to other threads).
__device__ __forceinline__ void put_entry(char *__restrict__ rec, int64_t e, uint32_t slot, bool rowstart, float v)
{
    const int64_t chunk = e >> 9;
    const int i = (int)(e & (CHUNK - 1)), lane = i >> 3, k = i & 7;
    chunk_vals(rec, chunk)[val_pos(e) & (CHUNK - 1)] = v;
    if (slot) {
        uint32_t *w = chunk_slots(rec, chunk) + lane * 3;
        const int bit = 12 * k, wi = bit >> 5, sh = bit & 31;
        atomicOr(w + wi, slot << sh);
        if (sh > 20) atomicOr(w + wi + 1, slot >> (32 - sh));
    }
    if (rowstart) atomicOr(chunk_masks(rec, chunk) + k, 1ull << lane);
}

// The same through a 16-bit staging array (one plain store per entry; k_pack_slots packs it afterwards): the row block's scatter
// writes every slot exactly once, so no atomics are needed for the slots - only the (rare) row-start bits stay atomic.
__device__ __forceinline__ void put_entry16(uint16_t *__restrict__ tmp16, int64_t base, char *__restrict__ rec, int64_t e, uint32_t slot,
                                            bool rowstart, float v)
{
    chunk_vals(rec, e >> 9)[val_pos(e) & (CHUNK - 1)] = v;
    tmp16[e - base] = (uint16_t)slot;
    if (rowstart) atomicOr(chunk_masks(rec, e >> 9) + (e & 7), 1ull << ((e & (CHUNK - 1)) >> 3));
}

// 8 staged 16-bit slots of a lane -> its three dwords of 12-bit slots
__global__ void k_pack_slots(const uint16_t *__restrict__ tmp16, int64_t base, int64_t nlanes, char *__restrict__ rec)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlanes; i += (int64_t)gridDim.x * blockDim.x) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 w = *reinterpret_cast<const u32x4 *>(tmp16 + i * 8);           // 8 x uint16
        const uint32_t s0 = w.x & 0xfffu, s1 = (w.x >> 16) & 0xfffu, s2 = w.y & 0xfffu, s3 = (w.y >> 16) & 0xfffu;
        const uint32_t s4 = w.z & 0xfffu, s5 = (w.z >> 16) & 0xfffu, s6 = w.w & 0xfffu, s7 = (w.w >> 16) & 0xfffu;
        const int64_t e = base + i * 8;
        uint32_t *o = chunk_slots(rec, e >> 9) + ((e & (CHUNK - 1)) >> 3) * 3;
        o[0] = s0 | (s1 << 12) | (s2 << 24);
        o[1] = (s2 >> 8) | (s3 << 4) | (s4 << 16) | (s5 << 28);
        o[2] = (s5 >> 4) | (s6 << 8) | (s7 << 20);
    }
}

// Scatter the entries of one row block into their tiles.  grid = (ceil(maxlen/256), nr).
__global__ void k_tile_scatter(const int32_t *__restrict__ cols, const float *__restrict__ vals,
                               const int32_t *__restrict__ nel, const int64_t *__restrict__ rowoff, int ntc, int TC, int nr,
                               const int32_t *__restrict__ pos, const int32_t *__restrict__ segoff,
                               const int64_t *__restrict__ tile_off, uint16_t *__restrict__ tmp16, int64_t base, char *__restrict__ rec)
{
    int r = blockIdx.y;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nel[r]) return;
    const int64_t src = rowoff[r] + j;
    int32_t c = cols[src];
    int t = c / TC;
    int p0 = pos[(int64_t)r * (ntc + 1) + t];
    int64_t dst = tile_off[t] + segoff[(int64_t)t * (nr + 1) + r] + (j - p0);
    put_entry16(tmp16, base, rec, dst, (uint32_t)col_slot(c - t * TC), j == p0, vals[src]);
}

// Markers for empty rows strictly between the first and last non-empty row of a tile.  grid = (ntc), block 256.
__global__ void k_tile_markers(const int32_t *__restrict__ pos, int nr, int ntc, const int32_t *__restrict__ segoff,
                               const int32_t *__restrict__ first_ne, const int32_t *__restrict__ last_ne,
                               const int64_t *__restrict__ tile_off, char *__restrict__ rec)
{
    int t = blockIdx.x;
    int f = first_ne[t], l = last_ne[t];
    if (f < 0) return;
    for (int r = f + 1 + threadIdx.x; r < l; r += blockDim.x) {
        int cnt = pos[(int64_t)r * (ntc + 1) + t + 1] - pos[(int64_t)r * (ntc + 1) + t];
        if (cnt == 0) put_entry(rec, tile_off[t] + segoff[(int64_t)t * (nr + 1) + r], 0u, true, 0.0f);
    }
}

// chunk_row0[chunk] = local row of the entry just before the chunk (first chunk: first_ne - 1).
__global__ void k_chunk_row0(int nr, const int32_t *__restrict__ segoff, const int32_t *__restrict__ first_ne,
                             const int64_t *__restrict__ tile_off, const int32_t *__restrict__ tile_nchunks,
                             int32_t *__restrict__ chunk_row0)
{
    int t = blockIdx.x;
    int nch = tile_nchunks[t];
    if (nch == 0) return;
    const int32_t *so = segoff + (int64_t)t * (nr + 1);
    int total = so[nr];
    int64_t cbase = tile_off[t] / CHUNK;
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        int row;
        if (c == 0) row = first_ne[t] - 1;
        else {
            int q = c * CHUNK - 1;              // entry before the chunk
            if (q >= total) q = total - 1;      // padding region: stay on the last row
            // largest r with so[r] <= q and segment r non-empty: upper_bound(q) - 1
            int lo = 0, hi = nr;                // so[0..nr]
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (so[mid + 1] <= q) lo = mid + 1;
                else hi = mid;
            }
            row = lo;
        }
        chunk_row0[cbase + c] = row;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Dense storage for an uncompressed kernel (forward.matrixCompression.type = 0): fp32 [nrows][ld], 4 B per entry.
// ------------------------------------------------------------------------------------------------------------
constexpr int DN_CHUNK = 8192;          // columns per workgroup (x chunk = 64 KB of LDS)
constexpr int DN_THREADS = 1024;

int matrix_begin_dense(tfx_ctx *ctx, int64_t nrows, int64_t ncols)
{
    TiledMatrix &m = *ctx->target;
    m.release_storage();            // (whatever the slot held before: tiles, work lists, a transposed copy)
    if (nrows <= 0 || ncols <= 0) return fail(TFX_E_ARG, "matrix_begin_dense: empty matrix");
    m.nrows = nrows;
    m.ncols = ncols;
    m.is_dense = true;
    m.ld = (ncols + 3) / 4 * 4;
    m.nnz = nrows * ncols;
    TFX_TRY(m.dense.alloc((size_t)(nrows * m.ld)));
    const int64_t nchunks = (ncols + DN_CHUNK - 1) / DN_CHUNK;
    TFX_TRY(m.dense_partial.alloc((size_t)(nchunks * nrows)));
    m.h_tiles.clear();
    m.h_fwd.clear();
    m.h_adj.clear();
    m.n_entries = 0;
    return 0;
}

// forward: workgroup = one column chunk, its x slice in LDS; a wave per row, partial[chunk][row] = dot over the chunk
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(DN_THREADS) void k_dense_fwd(const float *__restrict__ A, int64_t ld, int64_t nrows, int64_t ncols,
                                                           const double *__restrict__ x, double *__restrict__ partial)
{
    __shared__ double xs[DN_CHUNK];
    const int64_t c0 = (int64_t)blockIdx.x * DN_CHUNK;
    const int nc = (int)min((int64_t)DN_CHUNK, ncols - c0);
    for (int i = threadIdx.x; i < DN_CHUNK; i += DN_THREADS) xs[i] = (i < nc) ? x[c0 + i] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc4 = nc & ~3;
    for (int64_t r = wave; r < nrows; r += DN_THREADS / 64) {
        const float *row = A + r * ld + c0;
        double acc = 0.0;
        for (int i = lane * 4; i < nc4; i += 256) {
            const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(row + i));      // streamed once per product
            acc = fma((double)v.x, xs[i], acc);
            acc = fma((double)v.y, xs[i + 1], acc);
            acc = fma((double)v.z, xs[i + 2], acc);
            acc = fma((double)v.w, xs[i + 3], acc);
        }
        if (lane < nc - nc4) acc = fma((double)row[nc4 + lane], xs[nc4 + lane], acc);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
        if (lane == 0) partial[(int64_t)blockIdx.x * nrows + r] = acc;
    }
}

__global__ void k_dense_fwd_reduce(const double *__restrict__ partial, int nchunks, int64_t nrows, double *__restrict__ b, int add)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    double s = add ? b[r] : 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * nrows + r];
    b[r] = s;
}

// adjoint: workgroup = one column chunk over all rows; a thread owns two runs of 4 columns, one in each half of the chunk (each of
// its two 16-byte loads is then at a 16-byte lane stride: a 32-byte record per lane makes both loads of a wave touch every cache
// line of the row piece - tools/read_bw_probe.hip); u[r] is a wave-uniform scalar load
__global__ __launch_bounds__(DN_THREADS) void k_dense_adj(const float *__restrict__ A, int64_t ld, int64_t nrows, int64_t ncols,
                                                           const double *__restrict__ u, double *__restrict__ y)
{
    static_assert(DN_CHUNK == 8 * DN_THREADS, "two runs of four columns per thread");
    const int64_t ca = (int64_t)blockIdx.x * DN_CHUNK + (int64_t)threadIdx.x * 4, cb = ca + 4 * DN_THREADS;
    if (ca >= ncols) return;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int na = (int)min((int64_t)4, ncols - ca), nb = (int)max((int64_t)0, min((int64_t)4, ncols - cb));
    const bool full = na == 4 && nb == 4;
#pragma unroll 4
    for (int64_t r = 0; r < nrows; ++r) {
        const double ur = u[r];
        const float *p = A + r * ld;
        if (full) {
            const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p + ca)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p + cb));
            acc[0] = fma((double)a.x, ur, acc[0]); acc[1] = fma((double)a.y, ur, acc[1]);
            acc[2] = fma((double)a.z, ur, acc[2]); acc[3] = fma((double)a.w, ur, acc[3]);
            acc[4] = fma((double)b.x, ur, acc[4]); acc[5] = fma((double)b.y, ur, acc[5]);
            acc[6] = fma((double)b.z, ur, acc[6]); acc[7] = fma((double)b.w, ur, acc[7]);
        } else {
            for (int k = 0; k < na; ++k) acc[k] = fma((double)p[ca + k], ur, acc[k]);
            for (int k = 0; k < nb; ++k) acc[4 + k] = fma((double)p[cb + k], ur, acc[4 + k]);
        }
    }
    for (int k = 0; k < na; ++k) y[ca + k] += acc[k];
    for (int k = 0; k < nb; ++k) y[cb + k] += acc[4 + k];
}

__global__ void k_dense_scale_rows(float *__restrict__ A, int64_t ld, int64_t nrows, int64_t ncols, const float *__restrict__ scale)
{
    for (int64_t r = blockIdx.y; r < nrows; r += gridDim.y) {
        const float f = scale[r];
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x) A[r * ld + c] *= f;
    }
}

// Tile shape and capacity (in entries) of a tiled matrix.  Large matrices use the largest tile (RB_MAX x TC_MAX: least staging per
// entry).  A tile is never split between workgroups, so a small or medium matrix gets smaller tiles - enough of them for ~2
// workgroups per CU, but not finer than 16 chunks of entries (a workgroup's 16 waves): narrower column tiles first (less x staging,
// less LDS, more workgroups per CU), then lower row blocks.
static void tile_shape(const tfx_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz_upper, int &TC, int &RB, int &nrb, int &ntc, int64_t &cap)
{
    // RB is a power of two (64 .. RB_MAX): the public append entry point takes blocks of RB_MAX rows and splits them
    int64_t rb_full = 64;
    while (rb_full < RB_MAX && rb_full < nrows) rb_full *= 2;
    const int64_t tc_full = std::min<int64_t>(TC_MAX, (ncols + 63) / 64 * 64);
    const double density = std::min(1.0, (double)nnz_upper / ((double)nrows * (double)ncols));
    const double per_tile = std::max<double>(16.0 * CHUNK, (double)nnz_upper / (2.0 * std::max(1, ctx->num_cu)));
    const double area = per_tile / std::max(density, 1e-12);
    int64_t tc = 256;
    while (tc < tc_full && (double)tc * (double)rb_full < area) tc *= 2;
    tc = std::min(tc, tc_full);
    int64_t rb = 64;
    while (rb < rb_full && (double)rb * (double)tc < area) rb *= 2;
    TC = (int)tc;
    RB = (int)rb;
    nrb = (int)((nrows + RB - 1) / RB);
    ntc = (int)((ncols + TC - 1) / TC);
    // marker entries: a row that is empty inside a tile between two non-empty rows of that tile.  At most one per (row, column
    // tile); a tile needs two real entries to have any, and at most RB - 2 of them
    const int64_t markers = std::min<int64_t>((int64_t)nrows * ntc, (nnz_upper / 2 + 1) * (int64_t)RB);
    cap = nnz_upper + markers + (int64_t)nrb * ntc * CHUNK + CHUNK;
    cap = (cap + CHUNK - 1) / CHUNK * CHUNK;
}

int matrix_begin(tfx_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz_upper)
{
    TiledMatrix &m = *ctx->target;
    m.release_storage();
    m.nrows = nrows;
    m.ncols = ncols;
    m.nnz = 0;
    if (nrows <= 0 || ncols <= 0) return fail(TFX_E_ARG, "matrix_begin: empty matrix %lld x %lld", (long long)nrows, (long long)ncols);
    int64_t cap = 0;
    tile_shape(ctx, nrows, ncols, nnz_upper, m.TC, m.RB, m.nrb, m.ntc, cap);
    if (ctx->pre_take) {
        // the transposed copy: its storage was set aside when S was begun - take it over when it is large enough
        TiledMatrix::Prealloc *pre = ctx->pre_take;
        ctx->pre_take = nullptr;
        // (`cap` here counts the empty-row markers of S as entries, which the copy drops: the storage was sized on the kernel's entry
        // bound with 2 % head room; matrix_append_panel checks the capacity as it goes, and a copy that does not fit after all is given up)
        if ((double)pre->rec.n >= 0.97 * (double)(cap / CHUNK) * REC_BYTES && (double)pre->row0.n >= 0.97 * (double)(cap / CHUNK)) {
            std::swap(m.rec.p, pre->rec.p); std::swap(m.rec.n, pre->rec.n);
            std::swap(m.chunk_row0.p, pre->row0.p); std::swap(m.chunk_row0.n, pre->row0.n);
            m.cap_entries = std::min<int64_t>((int64_t)(m.rec.n / REC_BYTES), (int64_t)m.chunk_row0.n) * CHUNK;
            m.n_entries = 0;
            m.h_tiles.clear();
            return 0;
        }
        pre->rec.release();
        pre->row0.release();
    }
    TFX_TRY(m.rec.alloc((size_t)(cap / CHUNK) * REC_BYTES));
    TFX_TRY(m.chunk_row0.alloc((size_t)(cap / CHUNK)));
    m.cap_entries = cap;
    m.n_entries = 0;
    m.h_tiles.clear();
    // A large sparse kernel whose transposed copy will fit: the copy's storage is set aside now
    if (!m.is_transpose_copy && ctx->adj_copy != 0 && m.rec.bytes() >= ((size_t)1 << 30) && (&m == &ctx->mat || &m == &ctx->mat2)) {
        int tc, rb, nrb, ntc;
        int64_t capT = 0;
        tile_shape(ctx, ncols, nrows, nnz_upper + nnz_upper / 50, tc, rb, nrb, ntc, capT);
        const size_t need = (size_t)(capT / CHUNK) * (REC_BYTES + sizeof(int32_t));
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b > (double)need + 60e9) {      // (60 GB: the build's own buffers, 16 GB, and the copy's panel scratch - a panel of a
                                                                                                              // kernel with a dense band of columns was seen to need 30 GB)
            // on a helper thread (its thread-local allocation context is empty: no eviction can be triggered from there, and an
            // eviction on this thread waits for it in ~Prealloc): the build computes its first row blocks meanwhile
            m.pre.reset(new TiledMatrix::Prealloc());
            TiledMatrix::Prealloc *pre = m.pre.get();
            const int dev = ctx->device;
            const size_t nrec = (size_t)(capT / CHUNK) * REC_BYTES, nrow0 = (size_t)(capT / CHUNK);
            auto set_aside = [pre, dev, nrec, nrow0]() noexcept {
                try {
                    (void)hipSetDevice(dev);
                    if (pre->rec.alloc(nrec) != 0 || pre->row0.alloc(nrow0) != 0) {
                        pre->rec.release();
                        pre->row0.release();
                    }
                } catch (...) {             // (e.g. bad_alloc while fail() formats its message: the set-aside is optional)
                    pre->rec.release();
                    pre->row0.release();
                }
                (void)hipGetLastError();
            };
            try {
                pre->pending = std::async(std::launch::async, set_aside);
            } catch (const std::exception &) {      // (no thread to be had: allocate here and now, without the eviction retry)
                NoEvict optional;
                set_aside();
            }
        }
        (void)hipGetLastError();
    }
    return 0;
}

// Appends the tiles of one row block.  Row r of the block has d_nel[r] entries starting at d_cols/d_vals +
// d_rowoff[r] (columns ascending, 0-based local); maxlen >= max d_nel.  row_begin must be a multiple of RB, nr <= RB.
// Everything is queued on the ctx stream; the inputs may be reused by work queued on that stream afterwards.
int matrix_append_rows(tfx_ctx *ctx, int64_t row_begin, int64_t nr64, const int32_t *d_cols, const float *d_vals,
                       const int32_t *d_nel, const int64_t *d_rowoff, int64_t maxlen)
{
    TiledMatrix &m = *ctx->target;
    hipStream_t s = ctx->stream;
    int nr = (int)nr64;
    if (row_begin % m.RB != 0 || nr > m.RB || nr <= 0)
        return fail(TFX_E_ARG, "matrix_append_rows: rows [%lld, +%d) not aligned to row block %d", (long long)row_begin, nr, m.RB);
    int rb = (int)(row_begin / m.RB);
    const int ntc = m.ntc;
    // scratch of the conversion lives in the ctx (no allocation per row block once it has grown to size)
    tfx_ctx::AppendScratch &sc = ctx->append;
    const int32_t *s_cols = d_cols;
    const float *s_vals = d_vals;
    const int32_t *s_nel = d_nel;
    const int64_t *s_off = d_rowoff;
    TFX_TRY(sc.pos.ensure((size_t)nr * (ntc + 1)));
    TFX_TRY(sc.segoff.ensure((size_t)std::max(1, ntc) * (nr + 1)));
    TFX_TRY(sc.first_ne.ensure(std::max(1, ntc)));
    TFX_TRY(sc.last_ne.ensure(std::max(1, ntc)));
    TFX_TRY(sc.tile_nch.ensure(std::max(1, ntc)));
    TFX_TRY(sc.tile_off.ensure(std::max(1, ntc)));
    if (ntc > 0) {
        hipLaunchKernelGGL(k_tile_pos, dim3(nr), dim3(256), 0, s, s_cols, s_nel, s_off, nr, ntc, m.TC, sc.pos.p);
        hipLaunchKernelGGL(k_tile_scan, dim3(ntc), dim3(256), (size_t)(nr + 1) * sizeof(int32_t), s, sc.pos.p, nr, ntc,
                           sc.segoff.p, sc.first_ne.p, sc.last_ne.p);
        TFX_HIP(hipGetLastError());
    }
    // tile totals -> host
    sc.h_segoff_last.assign((size_t)std::max(1, ntc), 0);
    if (ntc > 0)
        TFX_HIP(hipMemcpy2DAsync(sc.h_segoff_last.data(), sizeof(int32_t), sc.segoff.p + nr, (size_t)(nr + 1) * sizeof(int32_t),
                                 sizeof(int32_t), ntc, hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    sc.h_off.resize((size_t)std::max(1, ntc));
    sc.h_nch.resize((size_t)std::max(1, ntc));
    int64_t cur = m.n_entries;
    for (int t = 0; t < ntc; ++t) {
        int32_t cnt = sc.h_segoff_last[t];
        int32_t nch = (cnt + CHUNK - 1) / CHUNK;
        sc.h_off[t] = cur;
        sc.h_nch[t] = nch;
        if (cnt > 0) {
            TileMeta tm;
            tm.off = cur;
            tm.nchunks = nch;
            tm.cnt = cnt;
            tm.t = t;
            tm.rb = rb;
            tm.kind = 0;
            tm.aux = 0;
            m.h_tiles.push_back(tm);
        }
        cur += (int64_t)nch * CHUNK;
    }
    if (cur > m.cap_entries)
        return fail(TFX_E_STATE, "tiled matrix capacity exceeded (%lld > %lld entries)", (long long)cur, (long long)m.cap_entries);
    if (ntc > 0 && cur > m.n_entries) {
        // zero the destination range (padding: slot 0 / no flag / value 0; the scatter ORs its bits in)
        const int64_t c0 = m.n_entries / CHUNK, c1 = cur / CHUNK;
        TFX_HIP(hipMemsetAsync(m.rec.p + c0 * REC_BYTES, 0, (size_t)(c1 - c0) * REC_BYTES, s));
        TFX_TRY(sc.tmp16.ensure((size_t)(cur - m.n_entries)));
        TFX_HIP(hipMemsetAsync(sc.tmp16.p, 0, (size_t)(cur - m.n_entries) * sizeof(uint16_t), s));
        TFX_HIP(hipMemcpyAsync(sc.tile_off.p, sc.h_off.data(), ntc * sizeof(int64_t), hipMemcpyHostToDevice, s));
        TFX_HIP(hipMemcpyAsync(sc.tile_nch.p, sc.h_nch.data(), ntc * sizeof(int32_t), hipMemcpyHostToDevice, s));
        const int64_t smax = maxlen;
        if (smax > 0)
            hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)((smax + 255) / 256), nr), dim3(256), 0, s, s_cols, s_vals,
                               s_nel, s_off, ntc, m.TC, nr, sc.pos.p, sc.segoff.p, sc.tile_off.p, sc.tmp16.p, m.n_entries, m.rec.p);
        hipLaunchKernelGGL(k_pack_slots, dim3((unsigned)std::min<int64_t>(8192, ((cur - m.n_entries) / 8 + 255) / 256)), dim3(256), 0, s,
                           sc.tmp16.p, m.n_entries, (cur - m.n_entries) / 8, m.rec.p);
        hipLaunchKernelGGL(k_tile_markers, dim3(ntc), dim3(256), 0, s, sc.pos.p, nr, ntc, sc.segoff.p, sc.first_ne.p, sc.last_ne.p,
                           sc.tile_off.p, m.rec.p);
        hipLaunchKernelGGL(k_chunk_row0, dim3(ntc), dim3(256), 0, s, nr, sc.segoff.p, sc.first_ne.p, sc.tile_off.p, sc.tile_nch.p,
                           m.chunk_row0.p);
        TFX_HIP(hipGetLastError());
    }
    TFX_HIP(hipStreamSynchronize(s));    // the host-side staging vectors above are reused by the next call
    m.n_entries = cur;
    return 0;
}

// Cuts the tile list into work items.  Forward: the key is the super block (fwd_group consecutive row blocks that share the
// staged x tile); inside a super block the tiles are ordered by (column tile, row block); every item owns one partial-sum
// tile (pidx) of fwd_group * RB doubles, the items of super block sb are pbase[sb] .. pbase[sb]+nslots[sb]-1.
// Adjoint: the key is the column tile; slot 0 of a column tile adds straight into y, slots >= 1 own partial tiles
// pbase[t] .. pbase[t]+nslots[t]-2 of TC doubles.
static void build_items(const std::vector<TileMeta> &tiles, bool forward, int group, int64_t target, std::vector<WorkItem> &items,
                        std::vector<int32_t> &order, int &npartial, std::vector<int32_t> &nslots,
                        std::vector<int32_t> &pbase, int nkeys)
{
    auto key_of = [&](const TileMeta &t) { return forward ? t.rb / group : t.t; };
    auto size_of = [&](const TileMeta &t) -> int64_t { return (int64_t)t.nchunks * CHUNK; };
    std::vector<int32_t> idx(tiles.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) {
        const int ka = key_of(tiles[a]), kb = key_of(tiles[b]);
        if (ka != kb) return ka < kb;
        if (forward) {
            if (tiles[a].t != tiles[b].t) return tiles[a].t < tiles[b].t;
            return tiles[a].rb < tiles[b].rb;
        }
        return tiles[a].rb < tiles[b].rb;
    });
    order = idx;
    items.clear();
    nslots.assign(nkeys, 0);
    pbase.assign(nkeys, 0);
    npartial = 0;
    std::vector<int64_t> item_sz;
    size_t i = 0;
    while (i < idx.size()) {
        int key = key_of(tiles[idx[i]]);
        size_t j = i;
        while (j < idx.size() && key_of(tiles[idx[j]]) == key) ++j;
        // split [i, j) into runs of about `target` entries
        int slot = 0;
        size_t b = i;
        pbase[key] = npartial;
        while (b < j) {
            const int64_t first = size_of(tiles[idx[b]]);
            if (first > target + target / 2 && tiles[idx[b]].nchunks >= 32) {
                // a tile much heavier than an item's share (wavelet-compressed rows put most entries of a small kernel into a
                // few column tiles): its chunks are dealt to several items - every chunk knows its first row (chunk_row0), and
                // the items add up through their partial tiles like any other items of the key
                const int nch = tiles[idx[b]].nchunks;
                const int nparts = (int)std::min<int64_t>((first + target - 1) / target, nch / 16);
                for (int q = 0; q < nparts; ++q) {
                    WorkItem w;
                    w.begin = (int32_t)b;
                    w.end = (int32_t)b + 1;
                    w.key = key;
                    w.slot = slot;
                    w.pidx = forward ? npartial++ : (slot == 0 ? -1 : npartial++);
                    w.cb = (int32_t)((int64_t)nch * q / nparts);
                    w.ce = (int32_t)((int64_t)nch * (q + 1) / nparts);
                    ++slot;
                    items.push_back(w);
                    item_sz.push_back(first * (w.ce - w.cb) / std::max(1, nch));
                }
                b += 1;
                continue;
            }
            int64_t acc = 0;
            size_t e = b;
            while (e < j && (acc == 0 || acc + size_of(tiles[idx[e]]) <= target)) {
                acc += size_of(tiles[idx[e]]);
                ++e;
            }
            WorkItem w;
            w.begin = (int32_t)b;
            w.end = (int32_t)e;
            w.key = key;
            w.slot = slot;
            w.pidx = forward ? npartial++ : (slot == 0 ? -1 : npartial++);
            w.cb = 0;
            w.ce = -1;
            ++slot;
            items.push_back(w);
            item_sz.push_back(acc);
            b = e;
        }
        nslots[key] = slot;
        i = j;
    }
    // largest first: the dispatcher hands out workgroups in order, so this approximates LPT scheduling
    std::vector<int32_t> perm(items.size());
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return item_sz[a] > item_sz[b]; });
    std::vector<WorkItem> sorted(items.size());
    for (size_t k = 0; k < perm.size(); ++k) sorted[k] = items[perm[k]];
    items.swap(sorted);
}

int matrix_finish(tfx_ctx *ctx)
{
    TiledMatrix &m = *ctx->target;
    if (m.is_dense) {
        m.valid = true;
        return 0;
    }
    hipStream_t s = ctx->stream;
    int64_t real = 0;
    for (auto &t : m.h_tiles) real += t.cnt;     // includes empty-row markers; exact nnz is set by the caller
    TFX_TRY(m.tiles.alloc(std::max<size_t>(1, m.h_tiles.size())));
    if (!m.h_tiles.empty())
        TFX_HIP(hipMemcpyAsync(m.tiles.p, m.h_tiles.data(), m.h_tiles.size() * sizeof(TileMeta), hipMemcpyHostToDevice, s));
    // about 16 work items per CU (measured at the headline size: 8 -> 38.1, 16 -> 37.3, 32 -> 38.0 ms per iteration), never finer than 16 chunks
    int64_t target = std::max<int64_t>((int64_t)16 * CHUNK, m.n_entries / std::max(1, ctx->num_cu * ctx->items_per_cu));
    // Forward super blocks: one staged x tile serves fwd_group row blocks.  Worth it when the matrix is tall (full-height row
    // blocks, several of them) and big enough that a super block still splits into many items; small systems keep group 1.
    // Two row blocks per group, not four: the x tile (32 KB) plus two row blocks of sums (32 KB) let TWO workgroups of 16 waves share
    // a CU (the sparse kernels are compiled for 8 waves per SIMD: 28 VGPRs, 72 SGPRs), which the three-stream read pattern needs to
    // reach its 6.7 TB/s (tools/read_bw_probe.hip); measured per iteration, groups of 4 / 2 / 1: headline 37.9 / 37.8 / 37.7 ms on one
    // box, 37.7 / 36.3 / 35.9 on another; config 3 (3.4e7 columns, where staging costs most) 43.5 / 42.3 / 42.4 ms.
    m.fwd_group = 1;
    if (m.RB == RB_MAX && m.nrb >= 2 && m.n_entries >= (int64_t)64 * target) m.fwd_group = 2;
    if (ctx->fwd_group_override > 0) m.fwd_group = std::max(1, std::min(ctx->fwd_group_override, FWD_GROUP_MAX));
    const int nsb = (m.nrb + m.fwd_group - 1) / m.fwd_group;
    std::vector<int32_t> fo, ao, fns, ans, fpb, apb;
    int nfp = 0, nap = 0;
    // Every forward item writes a partial tile of fwd_group * RB row sums that k_fwd_reduce reads back: for a matrix with few rows
    // per entry that traffic rivals the matrix itself (1e6 cells x 4096 data: 4096 items x 32 KB = 134 MB written and read per
    // product against 480 MB of tiles).  The forward list is therefore cut no finer than keeps the partial tiles at ~10 % of the
    // matrix bytes (never coarser than 2 items per CU).
    int64_t target_fwd = target;
    {
        const double part_bytes = (double)m.fwd_group * m.RB * sizeof(double);
        const double max_items = std::max(2.0 * std::max(1, ctx->num_cu), 0.1 * ((double)m.n_entries / CHUNK * REC_BYTES) / part_bytes);
        target_fwd = std::max<int64_t>(target, (int64_t)((double)m.n_entries / max_items));
    }
    build_items(m.h_tiles, true, m.fwd_group, target_fwd, m.h_fwd, fo, nfp, fns, fpb, nsb);
    build_items(m.h_tiles, false, 1, target, m.h_adj, ao, nap, ans, apb, m.ntc);
    TFX_TRY(m.fwd.alloc(std::max<size_t>(1, m.h_fwd.size())));
    TFX_TRY(m.adj.alloc(std::max<size_t>(1, m.h_adj.size())));
    TFX_TRY(m.fwd_order.alloc(std::max<size_t>(1, fo.size())));
    TFX_TRY(m.adj_order.alloc(std::max<size_t>(1, ao.size())));
    TFX_TRY(m.adj_nslots.alloc(std::max<size_t>(1, ans.size())));
    TFX_TRY(m.adj_pbase.alloc(std::max<size_t>(1, apb.size())));
    TFX_TRY(m.fwd_nslots.alloc(std::max<size_t>(1, fns.size())));
    TFX_TRY(m.fwd_pbase.alloc(std::max<size_t>(1, fpb.size())));
    if (!m.h_fwd.empty()) {
        TFX_HIP(hipMemcpyAsync(m.fwd.p, m.h_fwd.data(), m.h_fwd.size() * sizeof(WorkItem), hipMemcpyHostToDevice, s));
        TFX_HIP(hipMemcpyAsync(m.adj.p, m.h_adj.data(), m.h_adj.size() * sizeof(WorkItem), hipMemcpyHostToDevice, s));
        TFX_HIP(hipMemcpyAsync(m.fwd_order.p, fo.data(), fo.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        TFX_HIP(hipMemcpyAsync(m.adj_order.p, ao.data(), ao.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    TFX_HIP(hipMemcpyAsync(m.adj_nslots.p, ans.data(), ans.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    TFX_HIP(hipMemcpyAsync(m.adj_pbase.p, apb.data(), apb.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    TFX_HIP(hipMemcpyAsync(m.fwd_nslots.p, fns.data(), fns.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    TFX_HIP(hipMemcpyAsync(m.fwd_pbase.p, fpb.data(), fpb.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    TFX_TRY(m.fwd_partial.alloc(std::max<size_t>(1, (size_t)nfp * m.fwd_group * m.RB)));
    TFX_TRY(m.adj_partial.alloc(std::max<size_t>(1, (size_t)nap * m.TC)));
    TFX_HIP(hipStreamSynchronize(s));
    m.adj_has_partials = nap > 0;
    m.fwd_avg_nslots = nsb > 0 ? (double)nfp / (double)nsb : 0.0;
    m.vmax_stale = true;
    m.nnz = real;
    m.valid = true;
    if (!m.is_transpose_copy) {
        if (m.T) {                       // "refinish": the copy keeps its tiles, its work lists follow the current knobs
            TiledMatrix *keep = ctx->target;
            ctx->target = m.T;
            const int rc = matrix_finish(ctx);
            ctx->target = keep;
            TFX_TRY(rc);
        } else {
            const int rc = matrix_build_transpose(ctx, m);
            m.drop_prealloc();           // (storage set aside for a copy that was not made after all)
            if (rc) {                    // (only adj_copy = 1: the copy was demanded and does not fit) - the finish failed as a whole
                m.valid = false;
                return rc;
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// The two matrix kernels
// ------------------------------------------------------------------------------------------------------------
constexpr int SPMV_THREADS = 1024;

struct ChunkRegs {
    uint32_t w[3];    // 8 packed 12-bit slots
    float v[8];
};

__device__ __forceinline__ void load_chunk(const char *__restrict__ rec, int64_t chunk, int lane, ChunkRegs &c)
{
    // the matrix is streamed exactly once per product: non-temporal loads keep it from displacing the x / u tiles in L2
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const uint32_t *sp = chunk_slots(rec, chunk) + lane * 3;          // three dword loads, merged into one dwordx3 by the compiler
    c.w[0] = __builtin_nontemporal_load(sp);
    c.w[1] = __builtin_nontemporal_load(sp + 1);
    c.w[2] = __builtin_nontemporal_load(sp + 2);
    const float *vp = chunk_vals(rec, chunk) + lane * 4;              // val_pos(): entries k = 0..3 of all lanes, then k = 4..7
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(vp));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(vp + CHUNK / 2));
    c.v[0] = a.x; c.v[1] = a.y; c.v[2] = a.z; c.v[3] = a.w;
    c.v[4] = b.x; c.v[5] = b.y; c.v[6] = b.z; c.v[7] = b.w;
}

// slot k of the lane's eight (12 bits each in three dwords)
template <int K>
__device__ __forceinline__ uint32_t slot_of(const ChunkRegs &c)
{
    constexpr int bit = 12 * K, wi = bit >> 5, sh = bit & 31;
    if constexpr (sh <= 20) return (c.w[wi] >> sh) & 0xfffu;
    else return ((c.w[wi] >> sh) | (c.w[wi + 1] << (32 - sh))) & 0xfffu;
}

// The eight row-start masks of a chunk (wave-uniform address -> scalar loads), and the number of row starts in all lanes
// below this one (all 8 entries of those lanes): mbcnt straight on the stored masks, no ballots.
struct ChunkMasks { uint64_t m[8]; };
__device__ __forceinline__ int load_masks(const char *__restrict__ rec, int64_t chunk, ChunkMasks &mk)
{
    const uint64_t *p = chunk_masks(rec, chunk);
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mk.m[k] = p[k];
        acc = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk.m[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk.m[k], acc));
    }
    return acc;
}
#define ROWSTART_K(mk, k) __builtin_amdgcn_inverse_ballot_w64((mk).m[k])

// ---- segmented reduction helpers (DPP: data moves between lanes on the VALU, not through the LDS crossbar)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// sum += sum of the lane D positions up inside the same 16-lane row, when that lane ends on the same output row
template <int D>
__device__ __forceinline__ void seg_step(double &sum, int cur)
{
    constexpr int CTRL = 0x100 + D;                                           // row_shl:D  (lane i reads lane i + D of its row)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(sum), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(sum), CTRL, 0xf, 0xf, false);
    const int oc = __builtin_amdgcn_update_dpp(-2, cur, CTRL, 0xf, 0xf, false);   // lanes without a source keep -2 (never a row)
    if (oc == cur) sum += __hiloint2double(hi, lo);
}

// A work item walks its tiles in groups of up to FWD_GROUP_MAX tiles that share the LDS-staged vector(s): forward - the tiles of
// one column tile inside the item's super block (one staged x tile, row sums of all row blocks of the super block in LDS);
// adjoint - the tiles of consecutive row blocks inside one aligned super block (their u rows staged together).  The waves of the
// workgroup deal out the chunks of the whole group round-robin, so a barrier and a partly filled last round are paid once per
// group, not once per tile.  Everything here is wave-uniform (scalar registers).
struct TileGroup {
    int ng;                          // tiles in the group
    int total;                       // chunks in the group
    int pre[FWD_GROUP_MAX];          // first group-chunk of tile j (INT_MAX beyond ng)
    int64_t cbase[FWD_GROUP_MAX];    // first chunk of tile j in the streams (+ cb)
    int lrb[FWD_GROUP_MAX];          // row block of tile j relative to the super block
};

template <bool FORWARD>
__device__ __forceinline__ void make_group(const WorkItem &it, const int32_t *__restrict__ order, const TileMeta *__restrict__ tiles,
                                           int ti, int GROUP, TileGroup &g, TileMeta &first)
{
    first = tiles[order[ti]];
    const int sb = first.rb / GROUP;
    g.ng = 1;
    g.pre[0] = 0;
    g.cbase[0] = first.off / CHUNK;
    g.lrb[0] = first.rb - sb * GROUP;
    int run = first.nchunks;
    if (it.ce >= 0) {                  // a heavy tile shared by several items: this one takes its units [cb, ce)
        g.cbase[0] += it.cb;
        run = it.ce - it.cb;
    }
#pragma unroll
    for (int j = 1; j < FWD_GROUP_MAX; ++j) {
        g.pre[j] = 0x7fffffff;
        g.cbase[j] = 0;
        g.lrb[j] = 0;
        if (g.ng == j && ti + j < it.end && it.ce < 0) {
            const TileMeta tm = tiles[order[ti + j]];
            const bool same = FORWARD ? (tm.t == first.t) : (tm.rb / GROUP == sb);
            if (same) {
                g.ng = j + 1;
                g.pre[j] = run;
                g.cbase[j] = tm.off / CHUNK;
                g.lrb[j] = tm.rb - sb * GROUP;
                run += tm.nchunks;
            }
        }
    }
    g.total = run;
}

// group-chunk q -> chunk index in the streams and the tile's row block in the super block
__device__ __forceinline__ int64_t locate_chunk(const TileGroup &g, int q, int &lrb)
{
    int64_t c = g.cbase[0] + q;
    lrb = g.lrb[0];
#pragma unroll
    for (int j = 1; j < FWD_GROUP_MAX; ++j)
        if (q >= g.pre[j]) { c = g.cbase[j] + (q - g.pre[j]); lrb = g.lrb[j]; }
    return c;
}

// pointers of the matrix streams (passed by value to the product kernels)
struct MatPtrs {
    const WorkItem *items;
    const int32_t *order;
    const TileMeta *tiles;
    const char *rec;
    const int32_t *chunk_row0;
};

// (the streams are separate `const __restrict__` kernel arguments, not members of a struct: only then does the compiler know that
// nothing in the kernel writes them and reads the wave-uniform ones - masks, offsets, tile records - with scalar loads)
#define MAT_PARAMS                                                                                                         \
    const WorkItem *__restrict__ items, const int32_t *__restrict__ order, const TileMeta *__restrict__ tiles,             \
    const char *__restrict__ rec, const int32_t *__restrict__ chunk_row0
#define MAT_ARGS(mp) (mp).items, (mp).order, (mp).tiles, (mp).rec, (mp).chunk_row0

// Segmented sum over the lanes of a wave: lanes with equal keys are contiguous; on return the FIRST lane of every run of equal keys
// holds the run's total (the other lanes partial sums), *first says whether the lane is such a first lane.  Inside a 16-lane DPP row:
// 4 shift-and-add steps on the VALU (row_shl, no LDS crossbar); across the four rows: the row heads are read as scalars and carried
// backwards.  The association order is fixed (by lane), so the result is reproducible.
__device__ __forceinline__ double seg_reduce(double sum, int cur, int lane, bool *first)
{
    seg_step<1>(sum, cur);
    seg_step<2>(sum, cur);
    seg_step<4>(sum, cur);
    seg_step<8>(sum, cur);
    {
        const int tc0 = __builtin_amdgcn_readlane(cur, 15), tc1 = __builtin_amdgcn_readlane(cur, 31),
                  tc2 = __builtin_amdgcn_readlane(cur, 47);
        const int hc1 = __builtin_amdgcn_readlane(cur, 16), hc2 = __builtin_amdgcn_readlane(cur, 32),
                  hc3 = __builtin_amdgcn_readlane(cur, 48);
        const double hs1 = readlane_f64(sum, 16), hs2 = readlane_f64(sum, 32), hs3 = readlane_f64(sum, 48);
        // carry[r]: what the lanes of row r whose run reaches the end of the row still miss
        const double carry2 = (hc3 == tc2) ? hs3 : 0.0;
        const double carry1 = (hc2 == tc1) ? hs2 + ((hc2 == tc2) ? carry2 : 0.0) : 0.0;
        const double carry0 = (hc1 == tc0) ? hs1 + ((hc1 == tc1) ? carry1 : 0.0) : 0.0;
        const int row = lane >> 4;
        const int tc = row == 0 ? tc0 : (row == 1 ? tc1 : (row == 2 ? tc2 : -2));
        const double carry = row == 0 ? carry0 : (row == 1 ? carry1 : carry2);
        if (cur == tc) sum += carry;
    }
    const int pcur = __builtin_amdgcn_update_dpp(-3, cur, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    *first = pcur != cur;
    return sum;
}

// One chunk of the forward product: row segments summed inside a lane, the segment tails merged across lanes, one LDS add per
// (row, chunk) piece into o[row].  HEAD: the pieces of row `hr` - the row this wave's run of chunks starts in the middle of, whose
// earlier part belongs to another wave - go to the run's own slot *hp instead (see k_spmv_fwd).  Returns (HEAD only) whether
// the chunk holds a row start, i.e. whether row hr ends here.
template <bool HEAD>
__device__ __forceinline__ bool fwd_chunk(const char *__restrict__ rec, const int32_t *__restrict__ chunk_row0, int64_t ch, int lane,
                                          const double *xs, double *o, int hr, double *hp)
{
    ChunkRegs cr;
    ChunkMasks mk;
    load_chunk(rec, ch, lane, cr);
    int cur = chunk_row0[ch] + load_masks(rec, ch, mk);
    double acc = 0.0;
#define FWD_DST(row) ((HEAD && (row) == hr) ? hp : &o[row])
#define FWD_STEP(K)                                                          \
    if (ROWSTART_K(mk, K)) {                                                 \
        if (cur >= 0 && acc != 0.0) atomicAdd(FWD_DST(cur), acc);            \
        cur += 1;                                                            \
        acc = 0.0;                                                           \
    }                                                                        \
    acc = fma((double)cr.v[K], xs[slot_of<K>(cr)], acc);
    FWD_STEP(0) FWD_STEP(1) FWD_STEP(2) FWD_STEP(3) FWD_STEP(4) FWD_STEP(5) FWD_STEP(6) FWD_STEP(7)
#undef FWD_STEP
    // merge the tails of lanes that end on the same row (equal rows are contiguous lanes): the first lane of every run adds the total
    bool first;
    const double sum = seg_reduce(acc, cur, lane, &first);
    if (first && cur >= 0 && sum != 0.0) atomicAdd(FWD_DST(cur), sum);
#undef FWD_DST
    if constexpr (HEAD) return (mk.m[0] | mk.m[1] | mk.m[2] | mk.m[3] | mk.m[4] | mk.m[5] | mk.m[6] | mk.m[7]) != 0ull;
    return false;
}

// forward: one workgroup = a run of tiles of one super block (GROUP row blocks), ordered by column tile so that a staged x tile
// serves all row blocks of the group; partial[item][row of the super block] = sum over the run.
//
// The sums are REPRODUCIBLE: every floating-point addition happens in an order fixed by the matrix and its work lists, never by the
// timing of the waves (the reference's sums are sequential loops, sparse_matrix.f90:316-329).  The chunks of a tile group are cut into
// RUNS of `RUN` consecutive chunks, dealt to the waves round-robin (the 16 waves of a workgroup read inside one window of 16 RUN
// chunks at a time).  A row whose entries lie inside one run is only ever added to by that run's wave, and a wave's LDS adds are
// performed in program order.  A row that straddles the start of a run is the run's "head row": its pieces inside the run are summed
// in the run's own LDS slot, the run (of another wave) in which the row starts adds its pieces to the row directly, and after the
// group's barrier one wave adds the head sums to their rows in run order (a segmented sum over the lanes in a fixed association).
// (Run 0 has no head row inside the workgroup: when it starts inside a row - an item that owns a range of a heavy tile's chunks -
// the earlier part of the row belongs to another work item, and the items meet in k_fwd_reduce in fixed order.)
constexpr int FWD_MAX_RUNS = 256;       // head slots per tile group (3 KB of LDS); longer groups get longer runs
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 16 ? 8 : 1) void k_spmv_fwd(MAT_PARAMS, const double *__restrict__ x, double *__restrict__ partial,
                                                           int64_t ncols, int TC, int RB, int GROUP, int RUN)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s_head[FWD_MAX_RUNS];
    __shared__ int s_hrow[FWD_MAX_RUNS];
    constexpr int THREADS = WAVES * 64;
    double *xs = lds;          // TC
    double *outs = lds + TC;   // GROUP * RB
    const WorkItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nout = GROUP * RB;
    for (int i = tid; i < nout; i += THREADS) outs[i] = 0.0;
    int nruns_prev = 0;        // runs of the tile group before: their head sums are still to be added
    int ti = it.begin;
    while (true) {
        __syncthreads();
        if (wave == 0) {
            for (int base = 0; base < nruns_prev; base += 64) {
                const int i = base + lane;
                const bool in = i > 0 && i < nruns_prev;
                const int r = in ? s_hrow[i] : -1;
                bool first;
                const double sum = seg_reduce(in ? s_head[i] : 0.0, r, lane, &first);
                if (first && r >= 0 && sum != 0.0) atomicAdd(&outs[r], sum);
            }
        }
        if (ti >= it.end) break;
        TileGroup g;
        TileMeta tm;
        make_group<true>(it, order, tiles, ti, GROUP, g, tm);
        ti += g.ng;
        {
            // x is read at the columns' (bank-folded) slots
            const int64_t col0 = (int64_t)tm.t * TC;
            const int ncol = (int)min((int64_t)TC, ncols - col0);
            for (int i = tid; i < TC; i += THREADS) xs[col_slot(i)] = (i < ncol) ? x[col0 + i] : 0.0;
        }
        __syncthreads();
        // (a short group - small matrices have tiles of a few dozen chunks - is dealt chunk by chunk: runs of RUN would leave waves idle)
        const int run_len = max(g.total < 4 * WAVES * RUN ? 1 : RUN, (g.total + FWD_MAX_RUNS - 1) / FWD_MAX_RUNS);      // (never more than FWD_MAX_RUNS head slots, whatever RUN is)
        const int nruns = (g.total + run_len - 1) / run_len;
        nruns_prev = nruns;
        for (int run = wave; run < nruns; run += WAVES) {
            int q = run * run_len;
            const int q1 = min(q + run_len, g.total);
            if (run > 0) {
                int lrb0;
                const int64_t ch0 = locate_chunk(g, q, lrb0);
                const int hr = chunk_row0[ch0];
                if (lane == 0) { s_hrow[run] = hr >= 0 ? lrb0 * RB + hr : -1; s_head[run] = 0.0; }
                // chunks up to and including the first one with a row start: the only ones that can hold pieces of the head row
                bool open = true;
                while (open && q < q1) {
                    int lrb;
                    const int64_t ch = locate_chunk(g, q, lrb);
                    open = !fwd_chunk<true>(rec, chunk_row0, ch, lane, xs, outs + lrb * RB, lrb == lrb0 ? hr : -5, &s_head[run]);
                    ++q;
                }
            }
            for (; q < q1; ++q) {
                int lrb;
                const int64_t ch = locate_chunk(g, q, lrb);
                fwd_chunk<false>(rec, chunk_row0, ch, lane, xs, outs + lrb * RB, -5, nullptr);
            }
        }
    }
    __syncthreads();
    double *dst = partial + (int64_t)it.pidx * nout;
    for (int i = tid; i < nout; i += THREADS) dst[i] = outs[i];
}

// b[row] = (add ? b[row] : 0) + sum over the partial tiles of the row's super block (fixed order: deterministic).
// A block takes FR_ROWS rows; its FR_GROUPS thread groups each add every FR_GROUPS-th partial tile (small matrices have
// hundreds of partial tiles per row block: a single sequential chain per row would be latency-bound), then the group sums are
// added in group order.  SB = rows per super block = fwd_group * RB.
constexpr int FR_ROWS = 16, FR_GROUPS = 16;
__global__ __launch_bounds__(FR_ROWS * FR_GROUPS) void k_fwd_reduce(const double *__restrict__ partial, const int32_t *__restrict__ nslots,
                                                                    const int32_t *__restrict__ pbase, int SB, int64_t nrows,
                                                                    double *__restrict__ b, int add)
{
    __shared__ double part[FR_GROUPS][FR_ROWS];
    const int lr_in = threadIdx.x % FR_ROWS, g = threadIdx.x / FR_ROWS;
    const int64_t r = (int64_t)blockIdx.x * FR_ROWS + lr_in;
    double s = 0.0;
    if (r < nrows) {
        const int sb = (int)(r / SB), lr = (int)(r - (int64_t)sb * SB);
        const int ns = nslots[sb], p0 = pbase[sb];
        // (a thread's chain of adds is sequential by construction - the order is the result's bits -, its loads are not: eight in flight;
        //  with one load per add the kernel was bound by the latency of ~90 dependent L2 / HBM round trips per thread on a matrix with
        //  4096 rows and 1465 partial tiles per row: 28 us per product at `medium`)
        const double *q = partial + (int64_t)p0 * SB + lr;
        int k = g;
        for (; k + 7 * FR_GROUPS < ns; k += 8 * FR_GROUPS) {
            double a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = q[(int64_t)(k + j * FR_GROUPS) * SB];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += a[j];
        }
        for (; k < ns; k += FR_GROUPS) s += q[(int64_t)k * SB];
    }
    part[g][lr_in] = s;
    __syncthreads();
    if (g == 0 && r < nrows) {
        double t = add ? b[r] : 0.0;
#pragma unroll
        for (int q = 0; q < FR_GROUPS; ++q) t += part[q][lr_in];
        b[r] = t;
    }
}

// The same sums in the same association (hence the same bits) with one thread per row, consecutive threads on consecutive rows: for
// matrices with many rows and few partial tiles per super block - the transposed copy of a wide kernel has 10^7 rows and mostly one
// or two partial tiles per super block, and the blocked kernel above spent 0.28 ms per product there (16 rows per workgroup, most
// of its 16 thread groups idle).
__global__ __launch_bounds__(256) void k_fwd_reduce_flat(const double *__restrict__ partial, const int32_t *__restrict__ nslots,
                                                          const int32_t *__restrict__ pbase, int SB, int64_t nrows, double *__restrict__ b, int add)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        const int sb = (int)(r / SB), lr = (int)(r - (int64_t)sb * SB);
        const int ns = nslots[sb], p0 = pbase[sb];
        double t = add ? b[r] : 0.0;
        const double *pr = partial + (int64_t)p0 * SB + lr;
        if (ns <= FR_GROUPS) {                                // (the common case of a matrix with many rows: every group has at most one tile)
#pragma unroll 4
            for (int q = 0; q < FR_GROUPS; ++q) {
                double s = 0.0;
                if (q < ns) s += pr[(int64_t)q * SB];
                t += s;
            }
        } else {
            for (int q = 0; q < FR_GROUPS; ++q) {
                double s = 0.0;                               // (thread group q of k_fwd_reduce: every FR_GROUPS-th partial tile, in order; four loads in flight)
                int k = q;
                for (; k + 3 * FR_GROUPS < ns; k += 4 * FR_GROUPS) {
                    const double a0 = pr[(int64_t)k * SB], a1 = pr[(int64_t)(k + FR_GROUPS) * SB], a2 = pr[(int64_t)(k + 2 * FR_GROUPS) * SB],
                                 a3 = pr[(int64_t)(k + 3 * FR_GROUPS) * SB];
                    s += a0; s += a1; s += a2; s += a3;
                }
                for (; k < ns; k += FR_GROUPS) s += pr[(int64_t)k * SB];
                t += s;
            }
        }
        b[r] = t;
    }
}

// adjoint on the tiles of S themselves (no transposed copy: the matrix did not fit twice): one workgroup = a run of tiles of one
// column tile; slot 0 adds into y, the others write partials.
//
// The column sums of a tile group are accumulated in LDS by all 16 waves at once, so an fp64 accumulation would depend on the order
// in which the waves' adds arrive.  They are therefore accumulated EXACTLY, in 64-bit integers: the group's u rows are staged times
// 2^k, with k chosen such that no column sum of the group can reach 2^61 in magnitude - the bound is max|u| of the staged rows times
// the largest column sum of |value| of the group's tiles (tile_bound[], computed once per matrix, exactly, in integers: the same
// bits on every run).  Every product value * (u * 2^k) is rounded ONCE to an integer (the fp64 product has 77 significant bits; it
// is split into a multiple of 2^32 and a remainder by two fma's against 1.5 * 2^84 and 1.5 * 2^52, whose low mantissa bits then
// hold the two halves) and the integers are added with 64-bit LDS integer atomics - associative, so the sums do not depend on any
// order and can not overflow.  The grid of that one rounding is 2^-60 .. 2^-61 of the bound, i.e. of (largest column sum of |value|
// x max|u|): for the ~50 entries per column of a tile of a wavelet-compressed kernel about 2^-56 of the largest product, finer than
// the 2^-53 an fp64 accumulation rounds every partial sum to.  At the end of the group every thread converts its own columns back
// (int64 -> fp64, one rounding) and adds them to its running fp64 column sums, in group order.
constexpr double ADJ_MAGIC_LO = 6755399441055744.0;                       // 1.5 * 2^52: unit spacing
constexpr double ADJ_MAGIC_HI = 6755399441055744.0 * 4294967296.0;        // 1.5 * 2^84: spacing 2^32
constexpr int ADJ_COLS_PER_THREAD = TC_MAX / 1024;
__global__ __launch_bounds__(1024, 8) void k_spmv_adj(MAT_PARAMS, const double *__restrict__ tile_bound, const double *__restrict__ u,
                                                       double *__restrict__ y, double *__restrict__ partial, int64_t nrows, int64_t ncols,
                                                       int TC, int RB, int GROUP)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned long long s_umax[16];
    constexpr int THREADS = 1024, WAVES = 16;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(lds);        // TC
    double *us = lds + TC;                                                         // GROUP * RB
    const WorkItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < TC; i += THREADS) acc[i] = 0ull;
    double colsum[ADJ_COLS_PER_THREAD];          // fp64 sums of this thread's slots tid, tid + 1024, ...
#pragma unroll
    for (int j = 0; j < ADJ_COLS_PER_THREAD; ++j) colsum[j] = 0.0;
    int kprev = 0;                               // scale exponent of the group whose integer sums are still in acc[]
    bool pending = false;
    int ti = it.begin;
    while (ti < it.end) {
        TileGroup g;
        TileMeta tm;
        make_group<false>(it, order, tiles, ti, GROUP, g, tm);
        double csum = tile_bound[order[ti]];      // a column of the group collects at most the sum of the tiles' largest column sums
#pragma unroll
        for (int j = 1; j < FWD_GROUP_MAX; ++j)
            if (j < g.ng) csum = csum + tile_bound[order[ti + j]];
        ti += g.ng;
        __syncthreads();
        if (pending) {
            // integer sums of the group before -> this thread's fp64 column sums
#pragma unroll
            for (int j = 0; j < ADJ_COLS_PER_THREAD; ++j) {
                const int i = tid + j * THREADS;
                if (i < TC) {
                    const long long v = (long long)acc[i];
                    if (v != 0) {
                        colsum[j] += ldexp((double)v, -kprev);
                        acc[i] = 0ull;
                    }
                }
            }
        }
        // the u rows of the row blocks the group touches: [first tile's block, last tile's block] inside the super block
        int last = g.lrb[0];
#pragma unroll
        for (int j = 1; j < FWD_GROUP_MAX; ++j)
            if (j < g.ng) last = g.lrb[j];
        const int lo = g.lrb[0] * RB, hi = (last + 1) * RB;
        const int64_t row0 = (int64_t)(tm.rb / GROUP) * GROUP * RB;
        // max |u| of the staged rows, as the largest bit pattern of |u|: ordered like the values, and a NaN sorts above everything
        unsigned long long umb = 0ull;
        for (int i = lo + tid; i < hi; i += THREADS) {
            const double v = (row0 + i < nrows) ? u[row0 + i] : 0.0;
            us[i] = v;
            umb = max(umb, (unsigned long long)__double_as_longlong(fabs(v)));
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) umb = max(umb, (unsigned long long)__shfl_xor((long long)umb, d));
        if (lane == 0) s_umax[wave] = umb;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WAVES; ++w) umb = max(umb, s_umax[w]);
        const double umax = __longlong_as_double((long long)umb);
        const double bound = umax * csum;                          // >= |any column sum of the group| (and >= every single product)
        const bool finite = bound < __builtin_huge_val();          // false for inf and NaN (tile_bound is NaN when a value is)
        if (!finite) {
            // u holds an inf / NaN (or the bound overflows): the sums are poisoned like the fp64 sums would be
#pragma unroll
            for (int j = 0; j < ADJ_COLS_PER_THREAD; ++j) colsum[j] = __builtin_nan("");
        }
        pending = finite && bound > 0.0;
        int k = 0;
        if (pending) {
            k = 60 - ilogb(bound);                                 // bound < 2^(ilogb + 1)  ->  |sums| < 2^61 (+ the roundings of <= 8192 products)
            for (int i = lo + tid; i < hi; i += THREADS) us[i] = ldexp(us[i], k);
        }
        kprev = k;
        __syncthreads();
        if (!pending) continue;                                    // (wave-uniform) every product of the group is zero
        for (int q = wave; q < g.total; q += WAVES) {
            int lrb;
            const int64_t ch = locate_chunk(g, q, lrb);
            const double *ub = us + lrb * RB;
            ChunkRegs cr;
            ChunkMasks mk;
            load_chunk(rec, ch, lane, cr);
            int cur = chunk_row0[ch] + load_masks(rec, ch, mk);
            double uval = ub[max(cur, 0)];
            // p = value * uval (|p| < 2^61): s1 = p + 1.5 * 2^84 rounds p to a multiple th of 2^32 and holds th / 2^32 in its low
            // mantissa bits; p - th is exact to 2^-22 (fma) and |p - th| <= 2^31, s3 = (p - th) + 1.5 * 2^52 rounds it to an integer
            // held in ITS low mantissa bits; the 64-bit integer is th + that: low dword = low dword of s3, high dword = low dword of
            // s1 + the (sign-extending) high dword of s3's integer
#define ADJ_STEP(K)                                                                                          \
            if (ROWSTART_K(mk, K)) {                                                                         \
                cur += 1;                                                                                    \
                uval = ub[cur];                                                                              \
            }                                                                                                \
            if (cr.v[K] != 0.0f) {                                                                           \
                const double vd = (double)cr.v[K];                                                           \
                const double s1 = fma(vd, uval, ADJ_MAGIC_HI);                                               \
                const double th = s1 - ADJ_MAGIC_HI;                                                         \
                const double s3 = fma(vd, uval, -th) + ADJ_MAGIC_LO;                                         \
                const uint32_t qlo = (uint32_t)__double2loint(s3);                                           \
                const uint32_t qhi = (uint32_t)__double2loint(s1) + ((uint32_t)__double2hiint(s3) - 0x43380000u); \
                atomicAdd(&acc[slot_of<K>(cr)], ((unsigned long long)qhi << 32) | qlo);                      \
            }
            ADJ_STEP(0) ADJ_STEP(1) ADJ_STEP(2) ADJ_STEP(3) ADJ_STEP(4) ADJ_STEP(5) ADJ_STEP(6) ADJ_STEP(7)
#undef ADJ_STEP
        }
    }
    __syncthreads();
    const int64_t col0 = (int64_t)it.key * TC;
    const int ncol = (int)min((int64_t)TC, ncols - col0);
#pragma unroll
    for (int j = 0; j < ADJ_COLS_PER_THREAD; ++j) {
        const int i = tid + j * THREADS;           // slot i holds column col_slot(i) (an involution)
        if (i < TC) {
            double sum = colsum[j];
            if (pending) {
                const long long v = (long long)acc[i];
                if (v != 0) sum += ldexp((double)v, -kprev);
            }
            const int c = col_slot(i);
            if (it.slot == 0) {
                if (c < ncol) y[col0 + c] += sum;
            } else {
                partial[(int64_t)it.pidx * TC + c] = sum;
            }
        }
    }
}

// tile_bound[tile] = the largest column sum of |value| inside the tile, rounded up - what a column of the adjoint product can collect
// from the tile per unit of max|u|.  Exact integer arithmetic (|value| scaled to 30 bits below the tile's largest, rounded up, summed in
// 64-bit LDS integers), so the bound - and with it the scale of the adjoint's integer sums - has the same bits on every run.  NaN when
// a value is NaN / inf.
__global__ __launch_bounds__(256) void k_tile_bound(const TileMeta *__restrict__ tiles, int ntiles, const char *__restrict__ rec, int TC,
                                                     double *__restrict__ tile_bound)
{
    extern __shared__ unsigned long long cs[];      // TC column sums
    __shared__ unsigned long long part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const TileMeta tm = tiles[ti];
        const int64_t cbase = tm.off / CHUNK;
        uint32_t m = 0u;                            // bit pattern of |value|: ordered like the values, a NaN sorts above everything
        for (int c = wave; c < tm.nchunks; c += 4) {
            const float *v = chunk_vals(rec, cbase + c);
            for (int i = lane; i < CHUNK; i += 64) m = max(m, __float_as_uint(v[i]) & 0x7fffffffu);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
        if (lane == 0) part[wave] = m;
        for (int i = threadIdx.x; i < TC; i += blockDim.x) cs[i] = 0ull;
        __syncthreads();
        const uint32_t vmb = (uint32_t)max(max(part[0], part[1]), max(part[2], part[3]));
        __syncthreads();
        const float vmax = __uint_as_float(vmb);
        if (!(vmax < __builtin_huge_valf())) {       // NaN or inf among the values
            if (threadIdx.x == 0) tile_bound[ti] = __builtin_nan("");
            continue;
        }
        if (vmax == 0.0f) {
            if (threadIdx.x == 0) tile_bound[ti] = 0.0;
            continue;
        }
        const int ev = ilogbf(vmax) + 1;             // |value| < 2^ev
        for (int c = wave; c < tm.nchunks; c += 4) {
            ChunkRegs cr;
            load_chunk(rec, cbase + c, lane, cr);
            const uint32_t sl[8] = {slot_of<0>(cr), slot_of<1>(cr), slot_of<2>(cr), slot_of<3>(cr), slot_of<4>(cr), slot_of<5>(cr), slot_of<6>(cr), slot_of<7>(cr)};
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (cr.v[k] != 0.0f) atomicAdd(&cs[sl[k]], (unsigned long long)ceil(ldexp((double)fabsf(cr.v[k]), 30 - ev)));
        }
        __syncthreads();
        unsigned long long mx = 0ull;
        for (int i = threadIdx.x; i < TC; i += blockDim.x) mx = max(mx, cs[i]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mx = max(mx, (unsigned long long)__shfl_xor((long long)mx, d));
        __syncthreads();
        if (lane == 0) part[wave] = mx;
        __syncthreads();
        if (threadIdx.x == 0) tile_bound[ti] = ldexp((double)max(max(part[0], part[1]), max(part[2], part[3])), ev - 30);
        __syncthreads();
    }
}

// vals[e] *= scale[row of e] for every stored entry: what read_sensitivity_kernel does to a row on reload
// (sensitivity_gravmag.F90:834-843: the file holds the unscaled kernel, the matrix problem_weight * data_weight(row) times it).
__global__ __launch_bounds__(256) void k_scale_rows(const TileMeta *__restrict__ tiles, int ntiles, char *__restrict__ rec,
                                                     const int32_t *__restrict__ chunk_row0,
                                                     const float *__restrict__ scale, int64_t nrows, int RB)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int ti = blockIdx.y; ti < ntiles; ti += gridDim.y) {
        const TileMeta tm = tiles[ti];
        const int64_t row0 = (int64_t)tm.rb * RB;
        const int64_t cbase = tm.off / CHUNK;
        for (int c = blockIdx.x * 4 + wave; c < tm.nchunks; c += gridDim.x * 4) {
            ChunkMasks mk;
            int cur = chunk_row0[cbase + c] + load_masks(rec, cbase + c, mk);
            float *vp = chunk_vals(rec, cbase + c) + lane * 4;          // val_pos()
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((mk.m[k] >> lane) & 1ull) cur += 1;
                const int64_t r = row0 + max(cur, 0);
                float *v = vp + (k >> 2) * (CHUNK / 2) + (k & 3);
                if (r < nrows) *v = *v * scale[r];
            }
        }
    }
}

// y[col] += sum over the partial tiles (slots >= 1) of the column tile, fixed order
__global__ void k_adj_reduce(const double *__restrict__ partial, const int32_t *__restrict__ nslots,
                             const int32_t *__restrict__ pbase, int TC, int64_t ncols, double *__restrict__ y)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const int t = (int)(c / TC), lc = (int)(c - (int64_t)t * TC);
    const int ns = nslots[t];
    if (ns <= 1) return;
    const int p0 = pbase[t];
    double s = y[c];
    for (int k = 0; k < ns - 1; ++k) s += partial[(int64_t)(p0 + k) * TC + lc];
    y[c] = s;
}

// hipFuncAttributeMaxDynamicSharedMemorySize applies per (kernel, device) for the whole process, so the bookkeeping is process-wide and
// keyed by (device, kernel variant): it holds the running maximum, and the attribute is only ever raised - a second ctx on the same
// device with smaller tiles can not lower what an older ctx relies on, and another device registers for itself.
constexpr int LDS_MAX_DEVICES = 64, LDS_VARIANTS = 4;
static size_t g_lds_attr[LDS_MAX_DEVICES][LDS_VARIANTS];
static int set_lds_limit(tfx_ctx *ctx, int which, const void *fn, size_t bytes)
{
    const int dev = ctx->device;
    if (dev < 0 || dev >= LDS_MAX_DEVICES) return fail(TFX_E_ARG, "device %d out of range", dev);
    if (bytes <= g_lds_attr[dev][which]) return 0;
    TFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g_lds_attr[dev][which] = bytes;
    return 0;
}

// HIP events around every launch of the two product kernels (tfx_profile_enable).  The events are only recorded here - no host
// synchronisation inside the measured region; prof_drain turns the finished pairs into totals when they are asked for.
void prof_begin(tfx_ctx *ctx)
{
    if (!ctx->profile) return;
    tfx_ctx::ProfPair pp{nullptr, nullptr, 0};
    if (!ctx->prof_free.empty()) {
        pp = ctx->prof_free.back();
        ctx->prof_free.pop_back();
    } else {
        (void)hipEventCreate(&pp.a);
        (void)hipEventCreate(&pp.b);
    }
    (void)hipEventRecord(pp.a, ctx->stream);
    ctx->prof_pending.push_back(pp);
}
void prof_drain(tfx_ctx *ctx)
{
    for (auto &pp : ctx->prof_pending) {
        if (!pp.b) continue;
        (void)hipEventSynchronize(pp.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, pp.a, pp.b) == hipSuccess) {
            ctx->prof_ms[pp.which] += ms;
            ctx->prof_n[pp.which] += 1;
        }
        ctx->prof_free.push_back(pp);
    }
    ctx->prof_pending.clear();
}
void prof_end(tfx_ctx *ctx, int which)
{
    if (!ctx->profile || ctx->prof_pending.empty()) return;
    tfx_ctx::ProfPair &pp = ctx->prof_pending.back();
    pp.which = which;
    (void)hipEventRecord(pp.b, ctx->stream);
    if (ctx->prof_pending.size() >= 4096) prof_drain(ctx);       // bound the pool
}

int spmv_dev(tfx_ctx *ctx, const double *d_x, double *d_b, int add) { return spmv_dev(ctx, ctx->selmat(), d_x, d_b, add); }
int spmtv_dev(tfx_ctx *ctx, const double *d_x, double *d_b, int add) { return spmtv_dev(ctx, ctx->selmat(), d_x, d_b, add); }

// Diagnostics: how many chunks have all their non-zero values within `span` binades (candidates for a shared-exponent value format)
__global__ void k_chunk_exponent_span(const char *__restrict__ rec, int64_t nchunks, int span, unsigned long long *__restrict__ count,
                                      unsigned int *__restrict__ hist /* [34] span histogram */)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = w; c < nchunks; c += nw) {
        int emin = 1 << 30, emax = -(1 << 30);
        for (int k = 0; k < 8; ++k) {
            const uint32_t b = __float_as_uint(chunk_vals(rec, c)[lane * 8 + k]);
            const int e = (int)((b >> 23) & 0xff);
            if ((b & 0x7fffffffu) != 0) { emin = min(emin, e); emax = max(emax, e); }
        }
        for (int d = 32; d > 0; d >>= 1) { emin = min(emin, __shfl_xor(emin, d)); emax = max(emax, __shfl_xor(emax, d)); }
        if (lane == 0) {
            const int sp = emax >= emin ? emax - emin : 0;
            if (sp <= span) atomicAdd(count, 1ull);
            atomicAdd(&hist[min(sp, 33)], 1u);
        }
    }
}

int chunk_exponent_stats(tfx_ctx *ctx, TiledMatrix &m, int span, int64_t *fit, int64_t *total, unsigned int *hist34)
{
    const int64_t nch = m.n_entries / CHUNK;
    DBuf<unsigned long long> cnt;
    DBuf<unsigned int> dh;
    TFX_TRY(cnt.alloc(1));
    TFX_TRY(dh.alloc(34));
    TFX_HIP(hipMemsetAsync(cnt.p, 0, 8, ctx->stream));
    TFX_HIP(hipMemsetAsync(dh.p, 0, 34 * 4, ctx->stream));
    if (nch > 0) hipLaunchKernelGGL(k_chunk_exponent_span, dim3(4096), dim3(256), 0, ctx->stream, m.rec.p, nch, span, cnt.p, dh.p);
    unsigned long long h = 0;
    TFX_TRY(copy_any(&h, cnt.p, 8, ctx->stream));
    TFX_TRY(copy_any(hist34, dh.p, 34 * 4, ctx->stream));
    *fit = (int64_t)h;
    *total = nch;
    return 0;
}

static MatPtrs mat_ptrs(const TiledMatrix &m, bool forward)
{
    MatPtrs p;
    p.items = forward ? m.fwd.p : m.adj.p;
    p.order = forward ? m.fwd_order.p : m.adj_order.p;
    p.tiles = m.tiles.p;
    p.rec = m.rec.p;
    p.chunk_row0 = m.chunk_row0.p;
    return p;
}

// b (+)= M x with the forward kernel on the tiles of M (M = S for the forward product, M = the transposed copy for the adjoint);
// prof_slot: which profiling slot the launch is timed in (-1: none)
static int forward_product(tfx_ctx *ctx, TiledMatrix &m, const double *d_x, double *d_b, int add, int prof_slot)
{
    hipStream_t s = ctx->stream;
    const bool prof = prof_slot >= 0;
    const int SB = m.fwd_group * m.RB;
    const size_t lds = (size_t)(m.TC + SB) * sizeof(double);
    if (!m.h_fwd.empty()) {
        const MatPtrs mp = mat_ptrs(m, true);
        if (prof) prof_begin(ctx);
        TFX_TRY(set_lds_limit(ctx, 0, (const void *)k_spmv_fwd<16>, lds));
        hipLaunchKernelGGL((k_spmv_fwd<16>), dim3((unsigned)m.h_fwd.size()), dim3(SPMV_THREADS), lds, s, MAT_ARGS(mp), d_x,
                           m.fwd_partial.p, m.ncols, m.TC, m.RB, m.fwd_group, ctx->fwd_run);
        if (prof) prof_end(ctx, prof_slot);
        TFX_HIP(hipGetLastError());
    }
    // one thread per row where a row has few partial tiles (the transposed copy of a wide kernel: 10^6..10^7 rows, one or two tiles each);
    // 16 threads per row where it has many (S itself at the headline size and a rank's share of it: 99 856 rows x 164 tiles - the one-thread
    // form spent 87 us per product there, profiles/README.md round 5); same association, same bits
    if (m.nrows >= (int64_t)1 << 20 || (m.nrows >= 8192 && m.fwd_avg_nslots <= 16.0))
        hipLaunchKernelGGL(k_fwd_reduce_flat, dim3((unsigned)std::min<int64_t>((m.nrows + 255) / 256, (int64_t)ctx->num_cu * 32)), dim3(256), 0, s,
                           m.fwd_partial.p, m.fwd_nslots.p, m.fwd_pbase.p, SB, m.nrows, d_b, add);
    else
        hipLaunchKernelGGL(k_fwd_reduce, dim3((unsigned)((m.nrows + FR_ROWS - 1) / FR_ROWS)), dim3(FR_ROWS * FR_GROUPS), 0, s, m.fwd_partial.p,
                           m.fwd_nslots.p, m.fwd_pbase.p, SB, m.nrows, d_b, add);
    TFX_HIP(hipGetLastError());
    return 0;
}

int spmv_dev(tfx_ctx *ctx, TiledMatrix &m, const double *d_x, double *d_b, int add)
{
    if (!m.valid) return fail(TFX_E_STATE, "spmv: no matrix");
    const bool prof = (&m == &ctx->mat || &m == &ctx->mat2);
    hipStream_t s = ctx->stream;
    if (m.is_dense) {
        const int nchunks = (int)((m.ncols + DN_CHUNK - 1) / DN_CHUNK);
        if (prof) prof_begin(ctx);
        hipLaunchKernelGGL(k_dense_fwd, dim3(nchunks), dim3(DN_THREADS), 0, s, m.dense.p, m.ld, m.nrows, m.ncols, d_x, m.dense_partial.p);
        if (prof) prof_end(ctx, 0);
        hipLaunchKernelGGL(k_dense_fwd_reduce, dim3((unsigned)((m.nrows + 255) / 256)), dim3(256), 0, s, m.dense_partial.p, nchunks,
                           m.nrows, d_b, add);
        TFX_HIP(hipGetLastError());
        return 0;
    }
    return forward_product(ctx, m, d_x, d_b, add, prof ? 0 : -1);
}

int spmtv_dev(tfx_ctx *ctx, TiledMatrix &m, const double *d_x, double *d_b, int add)
{
    if (!m.valid) return fail(TFX_E_STATE, "spmtv: no matrix");
    const bool prof = (&m == &ctx->mat || &m == &ctx->mat2);
    hipStream_t s = ctx->stream;
    if (m.T && m.T->valid) return forward_product(ctx, *m.T, d_x, d_b, add, prof ? 1 : -1);      // the adjoint as a forward product on S^T
    if (!add) TFX_HIP(hipMemsetAsync(d_b, 0, (size_t)m.ncols * sizeof(double), s));
    if (m.is_dense) {
        const int nchunks = (int)((m.ncols + DN_CHUNK - 1) / DN_CHUNK);
        if (prof) prof_begin(ctx);
        hipLaunchKernelGGL(k_dense_adj, dim3(nchunks), dim3(DN_THREADS), 0, s, m.dense.p, m.ld, m.nrows, m.ncols, d_x, d_b);
        if (prof) prof_end(ctx, 1);
        TFX_HIP(hipGetLastError());
        return 0;
    }
    const size_t lds = (size_t)(m.TC + m.fwd_group * m.RB) * sizeof(double);
    if (!m.h_adj.empty()) {
        const MatPtrs mp = mat_ptrs(m, false);
        if (m.vmax_stale) {
            const int nt = (int)m.h_tiles.size();
            TFX_TRY(m.tile_bound.ensure((size_t)nt));
            hipLaunchKernelGGL(k_tile_bound, dim3((unsigned)std::min(nt, 65536)), dim3(256), (size_t)m.TC * sizeof(unsigned long long), s, m.tiles.p, nt,
                               m.rec.p, m.TC, m.tile_bound.p);
            TFX_HIP(hipGetLastError());
            m.vmax_stale = false;
        }
        if (prof) prof_begin(ctx);
        TFX_TRY(set_lds_limit(ctx, 2, (const void *)k_spmv_adj, lds));
        hipLaunchKernelGGL(k_spmv_adj, dim3((unsigned)m.h_adj.size()), dim3(SPMV_THREADS), lds, s, MAT_ARGS(mp), m.tile_bound.p, d_x, d_b,
                           m.adj_partial.p, m.nrows, m.ncols, m.TC, m.RB, m.fwd_group);
        if (prof) prof_end(ctx, 1);
        TFX_HIP(hipGetLastError());
        if (m.adj_has_partials)
            hipLaunchKernelGGL(k_adj_reduce, dim3((unsigned)((m.ncols + 255) / 256)), dim3(256), 0, s, m.adj_partial.p, m.adj_nslots.p,
                               m.adj_pbase.p, m.TC, m.ncols, d_b);
        TFX_HIP(hipGetLastError());
    }
    return 0;
}

// vals[e] *= scale[column of e]: tfx_matrix_scale_rows on the transposed copy (its columns are the rows of S)
__global__ __launch_bounds__(256) void k_scale_cols(const TileMeta *__restrict__ tiles, int ntiles, char *__restrict__ rec,
                                                     const float *__restrict__ scale, int64_t ncols, int TC)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int ti = blockIdx.y; ti < ntiles; ti += gridDim.y) {
        const TileMeta tm = tiles[ti];
        const int64_t cbase = tm.off / CHUNK, col0 = (int64_t)tm.t * TC;
        for (int c = blockIdx.x * 4 + wave; c < tm.nchunks; c += gridDim.x * 4) {
            ChunkRegs cr;
            load_chunk(rec, cbase + c, lane, cr);
            float *vp = chunk_vals(rec, cbase + c) + lane * 4;
            const uint32_t sl[8] = {slot_of<0>(cr), slot_of<1>(cr), slot_of<2>(cr), slot_of<3>(cr), slot_of<4>(cr), slot_of<5>(cr), slot_of<6>(cr), slot_of<7>(cr)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t col = col0 + col_slot((int)sl[k]);
                if (cr.v[k] != 0.0f && col < ncols) vp[(k >> 2) * (CHUNK / 2) + (k & 3)] = cr.v[k] * scale[col];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Transposed copy of a tiled matrix (the adjoint product as a forward product on S^T)
// ------------------------------------------------------------------------------------------------------------
// S^T is made PANEL by panel: a panel is a range of rows of S^T (= columns of S, whole row blocks of S^T) times a range of column
// tiles of S^T (= a range of rows of S).  For a wide kernel (10^7 columns, 10^5 rows) a panel is a band of ~4.5e5 rows of S^T times
// all its 25 column tiles - 23 panels of ~9e8 entries at the headline size, each a handful of launches over thousands of tiles (round
// 3 walked the 4864 row blocks of S^T one at a time: 60 000 small launches, 15 000 host synchronisations, 6.5 s).  Per panel:
//   (1) count: per tile of S that reaches into the panel, the entries per column (LDS histogram), by row block of S;
//   (2) prefix over the row blocks of S and scan over the columns: every (column, row block of S) knows where its run starts in the
//       column's row of S^T;
//   (3) fill: a workgroup takes one tile, slice of 64 rows by slice, marks (column, row) in an LDS bitmap; the rank of an entry inside
//       its (column, row block) run is the number of marked rows below it - the rows of S^T come out with ascending column indices
//       without a sort;
//   (4) the packed rows become tiles of S^T (matrix_append_panel: the tile conversion of matrix_append_rows for many row blocks at once).
// Zero-valued entries (markers, padding, stored zeros) are dropped: they add nothing to S^T x.

// every entry of a chunk as (local row, global column, value): the same decode as the product kernels
template <typename F>
__device__ __forceinline__ void walk_chunk(const char *__restrict__ rec, const int32_t *__restrict__ chunk_row0, int64_t ch, int lane,
                                           int64_t col0, F &&f)
{
    ChunkRegs cr;
    ChunkMasks mk;
    load_chunk(rec, ch, lane, cr);
    int cur = chunk_row0[ch] + load_masks(rec, ch, mk);
    const uint32_t sl[8] = {slot_of<0>(cr), slot_of<1>(cr), slot_of<2>(cr), slot_of<3>(cr), slot_of<4>(cr), slot_of<5>(cr), slot_of<6>(cr), slot_of<7>(cr)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((mk.m[k] >> lane) & 1ull) cur += 1;
        f(cur, col0 + col_slot((int)sl[k]), cr.v[k]);
    }
}

// a panel of the transposition: columns [c0, c1) and rows [R0, R1) of S; rb_lo = first row block of S that reaches into it
struct TrPanel {
    int64_t c0, c1, R0, R1;
    int rb_lo;
    int64_t npc;          // c1 - c0
};

// (1) cnt[(rb - rb_lo) * npc + (col - c0)] = entries of the tile in that column (one tile owns its (row block, columns): plain stores)
__global__ __launch_bounds__(1024) void k_tr_count(const TileMeta *__restrict__ tiles, const int32_t *__restrict__ tids, const char *__restrict__ rec,
                                                    const int32_t *__restrict__ chunk_row0, int TC, int RB, TrPanel pn, int32_t *__restrict__ cnt)
{
    extern __shared__ int32_t hist[];
    const TileMeta tm = tiles[tids[blockIdx.x]];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < TC; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int64_t cbase = tm.off / CHUNK, col0 = (int64_t)tm.t * TC, row0 = (int64_t)tm.rb * RB;
    for (int q = wave; q < tm.nchunks; q += 16) {
        walk_chunk(rec, chunk_row0, cbase + q, lane, col0, [&](int lrow, int64_t col, float val) {
            const int64_t row = row0 + lrow;
            if (val != 0.0f && col >= pn.c0 && col < pn.c1 && row >= pn.R0 && row < pn.R1) atomicAdd(&hist[(int)(col - col0)], 1);
        });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TC; i += blockDim.x) {
        const int64_t col = col0 + i;
        if (hist[i] && col >= pn.c0 && col < pn.c1) cnt[(int64_t)(tm.rb - pn.rb_lo) * pn.npc + (col - pn.c0)] = hist[i];
    }
}

// (2a) per column: exclusive prefix of the counts over the row blocks of S (in place), nel[j] = the column's total
__global__ __launch_bounds__(256) void k_tr_prefix(int32_t *__restrict__ cnt, int nsub, int64_t npc, int32_t *__restrict__ nel)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npc; j += (int64_t)gridDim.x * blockDim.x) {
        int run = 0;
        for (int sb = 0; sb < nsub; ++sb) {
            const int v = cnt[(int64_t)sb * npc + j];
            cnt[(int64_t)sb * npc + j] = run;
            run += v;
        }
        nel[j] = run;
    }
}

// (2b) exclusive scan of nel[0..n) into off[0..n] (64-bit) in three launches: block sums, scan of the block sums, offsets
constexpr int SCAN_BLOCK = 4096;      // elements per block of 1024 threads
__global__ __launch_bounds__(1024) void k_scan_sums(const int32_t *__restrict__ v, int64_t n, int64_t *__restrict__ bsum, int32_t *__restrict__ bmax)
{
    __shared__ long long part[1024];
    __shared__ int pmax[1024];
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    long long s = 0;
    int mx = 0;
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) { s += v[i0 + k]; mx = max(mx, v[i0 + k]); }
    part[threadIdx.x] = s;
    pmax[threadIdx.x] = mx;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) { part[threadIdx.x] += part[threadIdx.x + d]; pmax[threadIdx.x] = max(pmax[threadIdx.x], pmax[threadIdx.x + d]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { bsum[blockIdx.x] = part[0]; bmax[blockIdx.x] = pmax[0]; }
}

__global__ __launch_bounds__(1024) void k_scan_top(int64_t *__restrict__ bsum, const int32_t *__restrict__ bmax, int nb, int64_t *__restrict__ totals)
{
    // one block: exclusive scan of the block sums in place (nb is a few thousand), totals = {sum, largest element}
    __shared__ long long part[1024];
    __shared__ int pmax[1024];
    const int per = (nb + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    long long s = 0;
    int mx = 0;
    for (int b = b0; b < b0 + per && b < nb; ++b) { s += bsum[b]; mx = max(mx, bmax[b]); }
    part[threadIdx.x] = s;
    pmax[threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        int m = 0;
        for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = run; run += v; m = max(m, pmax[i]); }
        totals[0] = run;
        totals[1] = m;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int b = b0; b < b0 + per && b < nb; ++b) { const long long v = bsum[b]; bsum[b] = run; run += v; }
}

__global__ __launch_bounds__(1024) void k_scan_apply(const int32_t *__restrict__ v, int64_t n, const int64_t *__restrict__ bsum, int64_t *__restrict__ off)
{
    __shared__ long long part[1024];
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    int e[4];
    long long s = 0;
    for (int k = 0; k < 4; ++k) { e[k] = (i0 + k < n) ? v[i0 + k] : 0; s += e[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 thread sums
    for (int d = 1; d < 1024; d <<= 1) {
        const long long add = ((int)threadIdx.x >= d) ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = bsum[blockIdx.x] + part[threadIdx.x] - s;
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n) off[i0 + k] = run;
        run += e[k];
        if (i0 + k == n - 1) off[n] = run;
    }
}

// (3) one workgroup (4 waves) per tile of S.  The tile is walked in slices of 64 rows (its entries are in (row, column) order, so a
// slice is a range of chunks): the entries of the slice mark (column, row) in bm[column] - one 64-bit word per column of the tile -,
// then every entry reads its rank inside its column: the entries of the column in earlier slices (colcnt[column]) + the marked rows
// below it in this slice.  The tile is read about twice (round 3 / the first panel version read it 16 times, one 256-column strip of
// a [256][2048]-bit map at a time: 1.3 s of the 3.7 s the copy took at the headline size).
constexpr int TRF_THREADS = 256;
__global__ __launch_bounds__(TRF_THREADS) void k_tr_fill(const TileMeta *__restrict__ tiles, const int32_t *__restrict__ tids, const char *__restrict__ rec,
                                                         const int32_t *__restrict__ chunk_row0, int TC, int RB, TrPanel pn,
                                                         const int32_t *__restrict__ base /* [row blocks of S in the panel][npc] */,
                                                         const int64_t *__restrict__ rowoff, int32_t *__restrict__ tcols, float *__restrict__ tvals)
{
    extern __shared__ unsigned long long bm[];         // [TC] row bitmaps of the current slice, then [TC] uint32 column counts
    uint32_t *colcnt = reinterpret_cast<uint32_t *>(bm + TC);
    const TileMeta tm = tiles[tids[blockIdx.x]];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WAVES = TRF_THREADS / 64;
    const int64_t cbase = tm.off / CHUNK, col0 = (int64_t)tm.t * TC, row0 = (int64_t)tm.rb * RB;
    const int32_t *A = chunk_row0 + cbase;              // A[q] = local row of the entry before chunk q (non-decreasing)
    const int n = tm.nchunks;
    for (int i = threadIdx.x; i < TC; i += TRF_THREADS) { bm[i] = 0ull; colcnt[i] = 0u; }
    __syncthreads();
    const int first_row = max(A[0], 0);
    for (int lo = (first_row >> 6) << 6; lo < RB; lo += 64) {
        const int hi = lo + 63;
        // chunks that hold rows of [lo, hi]: from the first chunk whose last row is >= lo (the row before chunk q + 1) to the last
        // chunk whose first possible row (the row before it) is <= hi
        int qa, qb;
        {
            int l = 0, h = n - 1;                       // first q with (q == n - 1 or A[q + 1] >= lo)
            while (l < h) {
                const int mid = (l + h) >> 1;
                if (A[mid + 1] >= lo) h = mid;
                else l = mid + 1;
            }
            qa = l;
            l = qa; h = n;                              // first q with A[q] > hi
            while (l < h) {
                const int mid = (l + h) >> 1;
                if (A[mid] > hi) h = mid;
                else l = mid + 1;
            }
            qb = l;
        }
        if (qa >= qb) continue;                         // (uniform) no chunk reaches into this slice
        auto in_panel = [&](int lrow, int64_t col, float val) {
            const int64_t row = row0 + lrow;
            return val != 0.0f && lrow >= lo && lrow <= hi && col >= pn.c0 && col < pn.c1 && row >= pn.R0 && row < pn.R1;
        };
        for (int q = qa + wave; q < qb; q += WAVES)
            walk_chunk(rec, chunk_row0, cbase + q, lane, col0, [&](int lrow, int64_t col, float val) {
                if (in_panel(lrow, col, val)) atomicOr(&bm[(int)(col - col0)], 1ull << (lrow & 63));
            });
        __syncthreads();
        for (int q = qa + wave; q < qb; q += WAVES)
            walk_chunk(rec, chunk_row0, cbase + q, lane, col0, [&](int lrow, int64_t col, float val) {
                if (!in_panel(lrow, col, val)) return;
                const int c = (int)(col - col0);
                const int rank = (int)colcnt[c] + __popcll(bm[c] & ((1ull << (lrow & 63)) - 1ull));
                const int64_t jb = col - pn.c0;
                const int64_t dst = rowoff[jb] + base[(int64_t)(tm.rb - pn.rb_lo) * pn.npc + jb] + rank;
                tcols[dst] = (int32_t)(row0 + lrow);
                tvals[dst] = val;
            });
        __syncthreads();
        for (int i = threadIdx.x; i < TC; i += TRF_THREADS) {
            const unsigned long long w = bm[i];
            if (w) { colcnt[i] += (uint32_t)__popcll(w); bm[i] = 0ull; }
        }
        __syncthreads();
    }
}

// ---- (4) tile conversion of a panel: rows [r0, r0 + nrp) (nbp row blocks) x column tiles [t0, t0 + nt) -----------------------
// pos[r][k] (k = 0..nt) = first entry of packed row r with column >= (t0 + k) * TC; one wave per row
__global__ __launch_bounds__(256) void k_panel_pos(const int32_t *__restrict__ cols, const int32_t *__restrict__ nel, const int64_t *__restrict__ rowoff,
                                                    int64_t nrp, int t0, int nt, int TC, int32_t *__restrict__ pos)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = w; r < nrp; r += nw) {
        const int32_t *c = cols + rowoff[r];
        const int n = nel[r];
        for (int k = lane; k <= nt; k += 64) {
            int lo = 0;
            if (k == nt) lo = n;
            else if (k > 0) {
                const int64_t key = (int64_t)(t0 + k) * TC;
                int hi = n;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int64_t)c[mid] < key) lo = mid + 1;
                    else hi = mid;
                }
            }
            pos[r * (nt + 1) + k] = lo;
        }
    }
}

// grid = (nt, nbp): segment lengths (with empty-row markers) of tile (row block b, column tile k) and their exclusive scan over the
// rows of the block -> segoff[(b * nt + k) * (RB + 1) + r], first / last non-empty row, total
__global__ __launch_bounds__(256) void k_panel_scan(const int32_t *__restrict__ pos, int64_t nrp, int nt, int RB, int32_t *__restrict__ segoff,
                                                     int32_t *__restrict__ first_ne, int32_t *__restrict__ last_ne, int32_t *__restrict__ total)
{
    extern __shared__ int32_t sm[];     // RB ints
    __shared__ int s_first, s_last;
    __shared__ int part[256];
    const int k = blockIdx.x, b = blockIdx.y;
    const int64_t rbase = (int64_t)b * RB;
    const int nr = (int)min((int64_t)RB, nrp - rbase);
    if (threadIdx.x == 0) { s_first = 0x7fffffff; s_last = -1; }
    __syncthreads();
    for (int r = threadIdx.x; r < nr; r += blockDim.x) {
        const int32_t *pr = pos + (rbase + r) * (nt + 1) + k;
        const int cnt = pr[1] - pr[0];
        sm[r] = cnt;
        if (cnt > 0) { atomicMin(&s_first, r); atomicMax(&s_last, r); }
    }
    __syncthreads();
    const int f = s_first, l = s_last;
    const int per = (nr + blockDim.x - 1) / blockDim.x;
    const int rb0 = threadIdx.x * per, re = min(rb0 + per, nr);
    int sum = 0;
    for (int r = rb0; r < re; ++r) {
        int len = sm[r];
        if (len == 0 && r > f && r < l) len = 1;      // marker for an empty row between non-empty ones
        sm[r] = len;
        sum += len;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { const int v = part[i]; part[i] = run; run += v; }
        total[b * nt + k] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    int32_t *so = segoff + (int64_t)(b * nt + k) * (RB + 1);
    for (int r = rb0; r < re; ++r) { so[r] = run; run += sm[r]; }
    if (re == nr && rb0 < nr) so[nr] = run;
    if (nr == 0 && threadIdx.x == 0) so[0] = 0;
    if (threadIdx.x == 0) { first_ne[b * nt + k] = (l >= 0) ? f : -1; last_ne[b * nt + k] = l; }
}

// one wave per packed row: its entries go to their tiles
__global__ __launch_bounds__(256) void k_panel_scatter(const int32_t *__restrict__ cols, const float *__restrict__ vals, const int32_t *__restrict__ nel,
                                                        const int64_t *__restrict__ rowoff, int64_t nrp, int t0, int nt, int TC, int RB,
                                                        const int32_t *__restrict__ pos, const int32_t *__restrict__ segoff,
                                                        const int64_t *__restrict__ tile_off, uint16_t *__restrict__ tmp16, int64_t base, char *__restrict__ rec)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = w; r < nrp; r += nw) {
        const int n = nel[r];
        if (n == 0) continue;
        const int64_t o = rowoff[r];
        const int b = (int)(r / RB), rl = (int)(r - (int64_t)b * RB);
        const int32_t *pr = pos + r * (nt + 1);
        for (int j = lane; j < n; j += 64) {
            const int32_t c = cols[o + j];
            const int t = c / TC, k = t - t0;
            const int p0 = pr[k];
            const int64_t dst = tile_off[b * nt + k] + segoff[(int64_t)(b * nt + k) * (RB + 1) + rl] + (j - p0);
            put_entry16(tmp16, base, rec, dst, (uint32_t)col_slot(c - t * TC), j == p0, vals[o + j]);
        }
    }
}

// grid = (nt, nbp): markers for empty rows strictly between the first and last non-empty row of a tile
__global__ __launch_bounds__(256) void k_panel_markers(const int32_t *__restrict__ pos, int64_t nrp, int nt, int RB, const int32_t *__restrict__ segoff,
                                                        const int32_t *__restrict__ first_ne, const int32_t *__restrict__ last_ne,
                                                        const int64_t *__restrict__ tile_off, char *__restrict__ rec)
{
    const int k = blockIdx.x, b = blockIdx.y, id = b * nt + k;
    const int f = first_ne[id], l = last_ne[id];
    if (f < 0) return;
    const int64_t rbase = (int64_t)b * RB;
    for (int r = f + 1 + threadIdx.x; r < l; r += blockDim.x) {
        const int32_t *pr = pos + (rbase + r) * (nt + 1) + k;
        if (pr[1] - pr[0] == 0) put_entry(rec, tile_off[id] + segoff[(int64_t)id * (RB + 1) + r], 0u, true, 0.0f);
    }
}

// grid = (nt, nbp): chunk_row0[chunk] = local row of the entry just before the chunk (first chunk: first_ne - 1)
__global__ __launch_bounds__(256) void k_panel_chunk_row0(int64_t nrp, int nt, int RB, const int32_t *__restrict__ segoff, const int32_t *__restrict__ first_ne,
                                                           const int64_t *__restrict__ tile_off, const int32_t *__restrict__ tile_nchunks,
                                                           int32_t *__restrict__ chunk_row0)
{
    const int k = blockIdx.x, b = blockIdx.y, id = b * nt + k;
    const int nch = tile_nchunks[id];
    if (nch == 0) return;
    const int nr = (int)min((int64_t)RB, nrp - (int64_t)b * RB);
    const int32_t *so = segoff + (int64_t)id * (RB + 1);
    const int total = so[nr];
    const int64_t cbase = tile_off[id] / CHUNK;
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        int row;
        if (c == 0) row = first_ne[id] - 1;
        else {
            int q = c * CHUNK - 1;              // entry before the chunk
            if (q >= total) q = total - 1;      // padding region: stay on the last row
            int lo = 0, hi = nr;                // largest r with so[r] <= q and segment r non-empty
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (so[mid + 1] <= q) lo = mid + 1;
                else hi = mid;
            }
            row = lo;
        }
        chunk_row0[cbase + c] = row;
    }
}

// Appends the tiles of rows [r0, r0 + nrp) (r0 a multiple of RB) restricted to the column tiles [t0, t0 + nt): the rows lie packed one
// behind the other (d_rowoff[r + 1] = d_rowoff[r] + d_nel[r]; columns ascending, all inside the column tiles of the panel), `total`
// entries in all.  The tiles are stored row block by row block, column tile by column tile.
static int matrix_append_panel(tfx_ctx *ctx, int64_t r0, int64_t nrp, int t0, int nt, const int32_t *d_cols, const float *d_vals,
                               const int32_t *d_nel, const int64_t *d_rowoff, int64_t total)
{
    TiledMatrix &m = *ctx->target;
    hipStream_t s = ctx->stream;
    if (r0 % m.RB != 0 || nrp <= 0 || nt <= 0 || t0 < 0 || t0 + nt > m.ntc)
        return fail(TFX_E_ARG, "matrix_append_panel: bad panel (rows %lld + %lld, column tiles %d + %d)", (long long)r0, (long long)nrp, t0, nt);
    const int RB = m.RB, nbp = (int)((nrp + RB - 1) / RB), rb0 = (int)(r0 / RB);
    const int64_t ntl = (int64_t)nbp * nt;
    tfx_ctx::AppendScratch &sc = ctx->append;
    TFX_TRY(sc.pos.ensure((size_t)(nrp * (nt + 1))));
    TFX_TRY(sc.segoff.ensure((size_t)(ntl * (RB + 1))));
    TFX_TRY(sc.first_ne.ensure((size_t)ntl));
    TFX_TRY(sc.last_ne.ensure((size_t)ntl));
    TFX_TRY(sc.tile_nch.ensure((size_t)ntl));
    TFX_TRY(sc.tile_off.ensure((size_t)ntl));
    TFX_TRY(sc.tile_total.ensure((size_t)ntl));
    const unsigned row_grid = (unsigned)std::min<int64_t>((nrp + 3) / 4, (int64_t)ctx->num_cu * 64);
    hipLaunchKernelGGL(k_panel_pos, dim3(row_grid), dim3(256), 0, s, d_cols, d_nel, d_rowoff, nrp, t0, nt, m.TC, sc.pos.p);
    hipLaunchKernelGGL(k_panel_scan, dim3(nt, nbp), dim3(256), (size_t)RB * sizeof(int32_t), s, sc.pos.p, nrp, nt, RB, sc.segoff.p, sc.first_ne.p,
                       sc.last_ne.p, sc.tile_total.p);
    TFX_HIP(hipGetLastError());
    sc.h_segoff_last.assign((size_t)ntl, 0);
    TFX_HIP(hipMemcpyAsync(sc.h_segoff_last.data(), sc.tile_total.p, (size_t)ntl * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    sc.h_off.resize((size_t)ntl);
    sc.h_nch.resize((size_t)ntl);
    int64_t cur = m.n_entries;
    for (int b = 0; b < nbp; ++b)
        for (int k = 0; k < nt; ++k) {
            const size_t id = (size_t)b * nt + k;
            const int32_t cnt = sc.h_segoff_last[id];
            const int32_t nch = (cnt + CHUNK - 1) / CHUNK;
            sc.h_off[id] = cur;
            sc.h_nch[id] = nch;
            if (cnt > 0) {
                TileMeta tm;
                tm.off = cur;
                tm.nchunks = nch;
                tm.cnt = cnt;
                tm.t = t0 + k;
                tm.rb = rb0 + b;
                tm.kind = 0;
                tm.aux = 0;
                m.h_tiles.push_back(tm);
            }
            cur += (int64_t)nch * CHUNK;
        }
    if (cur > m.cap_entries)
        return fail(TFX_E_STATE, "tiled matrix capacity exceeded (%lld > %lld entries)", (long long)cur, (long long)m.cap_entries);
    if (cur > m.n_entries) {
        const int64_t ch0 = m.n_entries / CHUNK, ch1 = cur / CHUNK;
        TFX_HIP(hipMemsetAsync(m.rec.p + ch0 * REC_BYTES, 0, (size_t)(ch1 - ch0) * REC_BYTES, s));
        TFX_TRY(sc.tmp16.ensure((size_t)(cur - m.n_entries)));
        TFX_HIP(hipMemsetAsync(sc.tmp16.p, 0, (size_t)(cur - m.n_entries) * sizeof(uint16_t), s));
        TFX_HIP(hipMemcpyAsync(sc.tile_off.p, sc.h_off.data(), (size_t)ntl * sizeof(int64_t), hipMemcpyHostToDevice, s));
        TFX_HIP(hipMemcpyAsync(sc.tile_nch.p, sc.h_nch.data(), (size_t)ntl * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (total > 0)
            hipLaunchKernelGGL(k_panel_scatter, dim3(row_grid), dim3(256), 0, s, d_cols, d_vals, d_nel, d_rowoff, nrp, t0, nt, m.TC, RB, sc.pos.p,
                               sc.segoff.p, sc.tile_off.p, sc.tmp16.p, m.n_entries, m.rec.p);
        hipLaunchKernelGGL(k_pack_slots, dim3((unsigned)std::min<int64_t>(65536, ((cur - m.n_entries) / 8 + 255) / 256)), dim3(256), 0, s,
                           sc.tmp16.p, m.n_entries, (cur - m.n_entries) / 8, m.rec.p);
        hipLaunchKernelGGL(k_panel_markers, dim3(nt, nbp), dim3(256), 0, s, sc.pos.p, nrp, nt, RB, sc.segoff.p, sc.first_ne.p, sc.last_ne.p,
                           sc.tile_off.p, m.rec.p);
        hipLaunchKernelGGL(k_panel_chunk_row0, dim3(nt, nbp), dim3(256), 0, s, nrp, nt, RB, sc.segoff.p, sc.first_ne.p, sc.tile_off.p, sc.tile_nch.p,
                           m.chunk_row0.p);
        TFX_HIP(hipGetLastError());
    }
    TFX_HIP(hipStreamSynchronize(s));    // the host-side staging vectors above are reused by the next call
    m.n_entries = cur;
    return 0;
}

int matrix_build_transpose(tfx_ctx *ctx, TiledMatrix &m)
{
    if (m.is_dense || m.is_transpose_copy || m.h_tiles.empty() || ctx->adj_copy == 0) return 0;
    hipStream_t s = ctx->stream;
    if (m.pre) m.pre->wait();
    const bool have_storage = m.pre && m.pre->rec.p && m.pre->row0.p;       // set aside by matrix_begin
    if (ctx->adj_copy == 2 && !have_storage) {
        if (m.n_entries < ctx->adj_copy_min_nnz) return 0;
        size_t free_b = 0, total_b = 0;
        TFX_HIP(hipStreamSynchronize(s));
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
        // the copy (a little larger than the original: its own padding and markers) + the conversion scratch of a panel
        if ((double)free_b < 1.06 * (double)m.rec.bytes() + std::min(16e9, 0.2 * (double)m.rec.bytes() + 1e8)) return 0;
    }
    if (ctx->adj_copy == 2 && have_storage) {
        // the storage is there, but a panel of the transposition needs its scratch as well ((column, value) pairs + slots of up to three
        // nominal panels' worth of entries: a dense band of columns): without room for it the copy is given up NOW, not after the panel's allocation has failed
        size_t free_b = 0, total_b = 0;
        TFX_HIP(hipStreamSynchronize(s));
        const double scratch = std::min((double)m.n_entries, 3.0 * ctx->tr_panel_entries) * 10.0 + 1.0e9;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < scratch) {
            (void)hipGetLastError();
            fprintf(stderr, "[tfx] no transposed copy for the adjoint (%.1f GB free, the transposition needs %.1f GB of scratch): the adjoint runs on the tiles of S\n",
                    (double)free_b / 1e9, scratch / 1e9);
            m.drop_prealloc();
            return 0;
        }
    }
    const auto t_begin = std::chrono::steady_clock::now();
    TiledMatrix *T = new TiledMatrix();
    T->is_transpose_copy = true;
    T->evictable = ctx->adj_copy == 2;
    TiledMatrix *keep = ctx->target;
    auto give_up = [&](int rc) {
        // the copy is an optimisation: when the device has no room for it after all, the adjoint stays on the tiles of S
        ctx->target = keep;
        delete T;
        (void)hipGetLastError();
        if (ctx->adj_copy == 1) return rc;
        fprintf(stderr, "[tfx] no transposed copy for the adjoint (%s): the adjoint runs on the tiles of S\n", g_last_error.c_str());
        return 0;
    };
    ctx->target = T;
    ctx->pre_take = have_storage ? m.pre.get() : nullptr;
    int rc = matrix_begin(ctx, m.ncols, m.nrows, std::max<int64_t>(1, m.nnz));
    ctx->pre_take = nullptr;
    m.drop_prealloc();
    if (rc) return give_up(rc);
    const double t_alloc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    // Panels: bands of rows of S^T x all its column tiles when a band's per-row tile index (rows x (column tiles + 1) ints) fits the
    // budget - the usual case -, otherwise full height x a few column tiles, otherwise bands x single column tiles.  The band height /
    // the number of column tiles are cut so that a panel holds about PANEL_ENTRIES entries (estimated from the mean density; the scratch
    // grows to what a panel really holds).
    const double PANEL_ENTRIES = ctx->tr_panel_entries, POS_BUDGET = ctx->tr_pos_budget;      // entries (8 B each of scratch + 2 B) ; ints of pos[] / segoff[]
    const int RBt = T->RB, TCt = T->TC;
    const int64_t nrt = m.ncols;                                          // rows of S^T
    const double per_tile_col = (double)std::max<int64_t>(1, m.nnz) / (double)T->ntc;     // entries per column tile of S^T, all rows
    int nt_panel;
    int64_t band_rows;                                                    // rows of S^T per panel (multiple of RBt)
    const int64_t align = std::max<int64_t>(RBt, m.TC);                   // (powers of two) a band holds whole column tiles of S: every tile of S is read by one band
    if ((double)(T->ntc + 1) * (double)align <= POS_BUDGET) {
        // Bands of rows of S^T x ALL its column tiles: the tiles of S^T are then stored row block by row block like those of S (a
        // work item of the forward kernel walks the column tiles of one super block: with column-tile-major storage its tiles lay
        // 4.6 GB apart at the headline size, and the product on the copy ran 2 % behind the product on S)
        nt_panel = T->ntc;
        const double per_row = (double)std::max<int64_t>(1, m.nnz) / (double)nrt;
        int64_t rows = (int64_t)std::min(PANEL_ENTRIES / per_row, POS_BUDGET / (double)(T->ntc + 1));
        rows = std::max<int64_t>(align, rows / align * align);
        band_rows = std::min<int64_t>(rows, (nrt + align - 1) / align * align);
    } else if ((double)nrt * 2.0 <= POS_BUDGET) {
        band_rows = ((nrt + RBt - 1) / RBt) * RBt;
        nt_panel = (int)std::max(1.0, std::min({(double)T->ntc, POS_BUDGET / (double)nrt - 1.0, PANEL_ENTRIES / per_tile_col}));
        if (nt_panel == 1 && per_tile_col > 1.5 * PANEL_ENTRIES) {       // even one column tile is too much at full height: bands
            const int64_t nb = (int64_t)std::ceil(per_tile_col / PANEL_ENTRIES);
            band_rows = std::max<int64_t>(RBt, ((nrt / nb + RBt - 1) / RBt) * RBt);
        }
    } else {
        nt_panel = 1;
        band_rows = std::max<int64_t>(RBt, (int64_t)(POS_BUDGET / 2.0) / RBt * RBt);
    }
    // tiles of S by row block (the panels of one band of columns walk them row range by row range)
    std::vector<std::vector<int32_t>> by_rb((size_t)m.nrb);
    for (size_t i = 0; i < m.h_tiles.size(); ++i) by_rb[(size_t)m.h_tiles[i].rb].push_back((int32_t)i);
    tfx_ctx::TransposeScratch &sc = ctx->trs;
    const size_t fill_lds = (size_t)m.TC * (sizeof(unsigned long long) + sizeof(uint32_t));
    const bool timing = getenv("TFX_BUILD_TIMING") != nullptr;
    double t_count = 0, t_fill = 0, t_append = 0;
    int npanels = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    rc = [&]() -> int {
        TFX_TRY(sc.totals.ensure(2));
        TFX_HIP(hipFuncSetAttribute((const void *)k_tr_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fill_lds));
        std::vector<int32_t> tids;
        // (the order of the tiles in the streams is free - every tile carries its offset -, so any panel order gives the same matrix)
        for (int64_t r0 = 0; r0 < nrt; r0 += band_rows) {
            const int64_t r1 = std::min(nrt, r0 + band_rows), npc = r1 - r0;
            const int ct_lo = (int)(r0 / m.TC), ct_hi = (int)((r1 - 1) / m.TC);              // column tiles of S the band touches
            for (int t0 = 0; t0 < T->ntc; t0 += nt_panel) {
                const auto tp0 = now();
                const int nt = std::min(nt_panel, T->ntc - t0);
                TrPanel pn;
                pn.c0 = r0; pn.c1 = r1; pn.npc = npc;
                pn.R0 = (int64_t)t0 * TCt; pn.R1 = std::min<int64_t>(m.nrows, (int64_t)(t0 + nt) * TCt);
                pn.rb_lo = (int)(pn.R0 / m.RB);
                const int rb_hi = (int)((pn.R1 - 1) / m.RB), nsub = rb_hi - pn.rb_lo + 1;
                tids.clear();
                for (int rb = pn.rb_lo; rb <= rb_hi; ++rb)
                    for (int32_t i : by_rb[(size_t)rb])
                        if (m.h_tiles[(size_t)i].t >= ct_lo && m.h_tiles[(size_t)i].t <= ct_hi) tids.push_back(i);
                if (tids.empty()) continue;
                ++npanels;
                TFX_TRY(sc.tids.ensure(tids.size()));
                TFX_HIP(hipMemcpyAsync(sc.tids.p, tids.data(), tids.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
                TFX_TRY(sc.cnt.ensure((size_t)(nsub * npc)));
                TFX_TRY(sc.nel.ensure((size_t)npc));
                TFX_TRY(sc.rowoff.ensure((size_t)npc + 1));
                const int nsb = (int)((npc + SCAN_BLOCK - 1) / SCAN_BLOCK);
                TFX_TRY(sc.bsum.ensure((size_t)nsb));
                TFX_TRY(sc.bmax.ensure((size_t)nsb));
                TFX_HIP(hipMemsetAsync(sc.cnt.p, 0, (size_t)(nsub * npc) * sizeof(int32_t), s));
                hipLaunchKernelGGL(k_tr_count, dim3((unsigned)tids.size()), dim3(1024), (size_t)m.TC * sizeof(int32_t), s, m.tiles.p, sc.tids.p, m.rec.p,
                                   m.chunk_row0.p, m.TC, m.RB, pn, sc.cnt.p);
                hipLaunchKernelGGL(k_tr_prefix, dim3((unsigned)std::min<int64_t>((npc + 255) / 256, 16384)), dim3(256), 0, s, sc.cnt.p, nsub, npc, sc.nel.p);
                hipLaunchKernelGGL(k_scan_sums, dim3(nsb), dim3(1024), 0, s, sc.nel.p, npc, sc.bsum.p, sc.bmax.p);
                hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, s, sc.bsum.p, sc.bmax.p, nsb, sc.totals.p);
                hipLaunchKernelGGL(k_scan_apply, dim3(nsb), dim3(1024), 0, s, sc.nel.p, npc, sc.bsum.p, sc.rowoff.p);
                TFX_HIP(hipGetLastError());
                int64_t totals[2] = {0, 0};
                TFX_HIP(hipMemcpyAsync(totals, sc.totals.p, sizeof(totals), hipMemcpyDeviceToHost, s));
                TFX_HIP(hipStreamSynchronize(s));
                const auto tp1 = now();
                t_count += secs(tp0, tp1);
                if (totals[0] == 0) continue;                  // nothing of S in this panel
                if ((size_t)totals[0] > sc.tcols.n) {          // (grown with head room: the panels differ in density, and a re-allocation of GBs costs 0.1 s)
                    const size_t want = (size_t)((double)totals[0] * 1.25) + 1024;
                    TFX_TRY(sc.tcols.ensure(want));
                    TFX_TRY(sc.tvals.ensure(want));
                }
                hipLaunchKernelGGL(k_tr_fill, dim3((unsigned)tids.size()), dim3(TRF_THREADS), fill_lds, s, m.tiles.p,
                                   sc.tids.p, m.rec.p, m.chunk_row0.p, m.TC, m.RB, pn, sc.cnt.p, sc.rowoff.p, sc.tcols.p, sc.tvals.p);
                TFX_HIP(hipGetLastError());
                if (timing) (void)hipStreamSynchronize(s);
                const auto tp2 = now();
                t_fill += secs(tp1, tp2);
                TFX_TRY(matrix_append_panel(ctx, r0, npc, t0, nt, sc.tcols.p, sc.tvals.p, sc.nel.p, sc.rowoff.p, totals[0]));
                t_append += secs(tp2, now());
            }
        }
        const auto t3 = now();
        TFX_TRY(matrix_finish(ctx));
        if (timing)
            fprintf(stderr, "[tfx] transposed copy: allocation %.2f s; %d panels (%lld rows x %d column tiles each): count + scan %.2f s, fill %.2f s, tile conversion %.2f s; work lists %.2f s\n",
                    t_alloc, npanels, (long long)band_rows, nt_panel, t_count, t_fill, t_append, secs(t3, now()));
        return 0;
    }();
    // the conversion scratch is as large as the densest panel: give it back
    const auto t_rel = std::chrono::steady_clock::now();
    sc.tcols.release(); sc.tvals.release(); sc.cnt.release(); sc.nel.release(); sc.rowoff.release();
    ctx->append.pos.release(); ctx->append.segoff.release(); ctx->append.tmp16.release();
    if (timing) fprintf(stderr, "[tfx] transposed copy: scratch released in %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rel).count());
    if (rc) return give_up(rc);
    ctx->target = keep;
    m.T = T;
    (void)hipStreamSynchronize(s);
    m.copy_build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

// rows of the selected matrix times a per-row factor (fp32 product of the stored value and the factor, like the reference's
// sensit_compressed * real(problem_weight * data_weight, MATRIX_PRECISION), sensitivity_gravmag.F90:834-843)
int scale_rows_dev(tfx_ctx *ctx, TiledMatrix &m, const float *d_scale)
{
    if (!m.valid) return fail(TFX_E_STATE, "scale_rows: no matrix");
    hipStream_t s = ctx->stream;
    if (m.is_dense) {
        hipLaunchKernelGGL(k_dense_scale_rows, dim3(256, (unsigned)std::min<int64_t>(m.nrows, 65535)), dim3(256), 0, s, m.dense.p, m.ld,
                           m.nrows, m.ncols, d_scale);
        TFX_HIP(hipGetLastError());
        return 0;
    }
    const int nt = (int)m.h_tiles.size();
    if (nt == 0) return 0;
    m.vmax_stale = true;
    hipLaunchKernelGGL(k_scale_rows, dim3(8, (unsigned)std::min(nt, 32768)), dim3(256), 0, s, m.tiles.p, nt, m.rec.p,
                       m.chunk_row0.p, d_scale, m.nrows, m.RB);
    TFX_HIP(hipGetLastError());
    if (m.T && m.T->valid && !m.T->h_tiles.empty()) {        // the same factors, by column, on the transposed copy (same fp32 products)
        TiledMatrix &t = *m.T;
        hipLaunchKernelGGL(k_scale_cols, dim3(8, (unsigned)std::min((int)t.h_tiles.size(), 32768)), dim3(256), 0, s, t.tiles.p,
                           (int)t.h_tiles.size(), t.rec.p, d_scale, t.ncols, t.TC);
        TFX_HIP(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// t_sparse_matrix%normalize_columns (src/inversion/sparse_matrix.f90:414-443): column_norm(j) = sqrt(sum_k sa(k)**2) with the square
// formed in fp32 and added in fp64, then sa(k) = real(sa(k) / column_norm(j), 4) for the columns whose norm is not zero.
// The sum is formed EXACTLY here (so it has the same bits on every run whatever order the entries arrive in): pass 1 finds the largest
// square of a column, pass 2 adds every square as two 64-bit integers on a grid fixed by that maximum (units of 2^-40 and 2^-85 of the
// maximum's binade for up to 2^17 rows; every further doubling of the row count coarsens both words by one bit - `shift` =
// max(0, ceil(log2 nrows) - 17) - so neither can overflow for any row count: hi < 2^(41 - shift) nrows <= 2^58, lo < 2^(45 - shift) nrows
// <= 2^62; bits below 2^-(85 - 2 shift) of the largest square are dropped - 2^-65 at 2^27 rows, still far below fp64's 2^-53), pass 3
// rounds once.  The reference's sequential fp64 sum differs from the exact one by at most nrows * 2^-53 relative.
// A column that holds a NaN or an infinite square (a value beyond 1.8e19 overflows sa**2 in fp32, as it does in the reference) gets the
// norm the reference's sum would have - NaN resp. +Inf - and its entries the quotients by it.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsq_max(const TileMeta *__restrict__ tiles, int ntiles, const char *__restrict__ rec, int TC, int64_t ncols,
                                                    uint32_t *__restrict__ qmax)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int ti = blockIdx.y; ti < ntiles; ti += gridDim.y) {
        const TileMeta tm = tiles[ti];
        const int64_t cbase = tm.off / CHUNK, col0 = (int64_t)tm.t * TC;
        for (int c = blockIdx.x * 4 + wave; c < tm.nchunks; c += gridDim.x * 4) {
            ChunkRegs cr;
            load_chunk(rec, cbase + c, lane, cr);
            const uint32_t sl[8] = {slot_of<0>(cr), slot_of<1>(cr), slot_of<2>(cr), slot_of<3>(cr), slot_of<4>(cr), slot_of<5>(cr), slot_of<6>(cr), slot_of<7>(cr)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t col = col0 + col_slot((int)sl[k]);
                const float q = cr.v[k] * cr.v[k];                       // sa(k)**2 in MATRIX_PRECISION (:427)
                // (non-negative floats order like their bit patterns; +Inf sits above every finite square and a NaN - sign cleared - above
                // +Inf, so the maximum also says whether the column holds a non-finite square)
                if (!(q <= 0.0f) && col < ncols) atomicMax(qmax + col, __float_as_uint(q) & 0x7fffffffu);
            }
        }
    }
}

__device__ __forceinline__ void colsq_split(float q, uint32_t qmax_bits, int shift, unsigned long long &hi, unsigned long long &lo)
{
    const int e = (int)((qmax_bits >> 23) & 0xffu) - 127;            // binade of the column's largest square (denormal: -127, still a valid grid)
    const double t = ldexp((double)q, 40 - shift - e);               // exact; < 2^(41 - shift)
    const double th = floor(t);
    hi = (unsigned long long)th;
    lo = (unsigned long long)floor(ldexp(t - th, 45 - shift));       // (t - th is exact, < 1)
}

__global__ __launch_bounds__(256) void k_colsq_sum(const TileMeta *__restrict__ tiles, int ntiles, const char *__restrict__ rec, int TC, int64_t ncols,
                                                    const uint32_t *__restrict__ qmax, unsigned long long *__restrict__ acc, int shift)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int ti = blockIdx.y; ti < ntiles; ti += gridDim.y) {
        const TileMeta tm = tiles[ti];
        const int64_t cbase = tm.off / CHUNK, col0 = (int64_t)tm.t * TC;
        for (int c = blockIdx.x * 4 + wave; c < tm.nchunks; c += gridDim.x * 4) {
            ChunkRegs cr;
            load_chunk(rec, cbase + c, lane, cr);
            const uint32_t sl[8] = {slot_of<0>(cr), slot_of<1>(cr), slot_of<2>(cr), slot_of<3>(cr), slot_of<4>(cr), slot_of<5>(cr), slot_of<6>(cr), slot_of<7>(cr)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t col = col0 + col_slot((int)sl[k]);
                const float q = cr.v[k] * cr.v[k];
                if (q > 0.0f && col < ncols) {
                    const uint32_t qb = qmax[col];
                    if ((qb >> 23) == 0xffu) continue;                   // a non-finite square in this column: the norm is NaN / Inf whatever the rest adds
                    unsigned long long hi, lo;
                    colsq_split(q, qb, shift, hi, lo);
                    if (hi) atomicAdd(acc + 2 * col, hi);
                    if (lo) atomicAdd(acc + 2 * col + 1, lo);
                }
            }
        }
    }
}

__global__ void k_colsq_finish(const uint32_t *__restrict__ qmax, const unsigned long long *__restrict__ acc, int64_t ncols, double *__restrict__ norm,
                               int shift)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    const uint32_t qb = qmax[j];
    if (qb == 0u) { norm[j] = 0.0; return; }
    if ((qb >> 23) == 0xffu) {                 // sum(...) of the reference is NaN (a NaN square) or +Inf (an overflowed one); sqrt keeps either
        norm[j] = (qb & 0x7fffffu) ? __longlong_as_double(0x7ff8000000000000ll) : __longlong_as_double(0x7ff0000000000000ll);
        return;
    }
    const int e = (int)((qb >> 23) & 0xffu) - 127;
    // hi < 2^58, lo < 2^62 in units 2^-(45 - shift) of hi's: one fp64 sum of the two words, one rounding each for the conversions
    const double sum = ldexp((double)acc[2 * j] + ldexp((double)acc[2 * j + 1], -(45 - shift)), e - (40 - shift));
    norm[j] = sqrt(sum);                                                                    // :432
}

// sa(k) = real(sa(k) / column_norm(j), MATRIX_PRECISION) (:439); BY_ROW: the transposed copy, whose rows are the columns of S
template <bool BY_ROW>
__global__ __launch_bounds__(256) void k_div_by_norm(const TileMeta *__restrict__ tiles, int ntiles, char *__restrict__ rec,
                                                      const int32_t *__restrict__ chunk_row0, int TC, int RB, int64_t nlimit,
                                                      const double *__restrict__ norm)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int ti = blockIdx.y; ti < ntiles; ti += gridDim.y) {
        const TileMeta tm = tiles[ti];
        const int64_t cbase = tm.off / CHUNK, col0 = (int64_t)tm.t * TC, row0 = (int64_t)tm.rb * RB;
        for (int c = blockIdx.x * 4 + wave; c < tm.nchunks; c += gridDim.x * 4) {
            ChunkRegs cr;
            ChunkMasks mk;
            load_chunk(rec, cbase + c, lane, cr);
            int cur = chunk_row0[cbase + c] + load_masks(rec, cbase + c, mk);
            float *vp = chunk_vals(rec, cbase + c) + lane * 4;
            const uint32_t sl[8] = {slot_of<0>(cr), slot_of<1>(cr), slot_of<2>(cr), slot_of<3>(cr), slot_of<4>(cr), slot_of<5>(cr), slot_of<6>(cr), slot_of<7>(cr)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((mk.m[k] >> lane) & 1ull) cur += 1;
                const int64_t j = BY_ROW ? row0 + max(cur, 0) : col0 + col_slot((int)sl[k]);
                if (cr.v[k] != 0.0f && j < nlimit) {
                    const double nj = norm[j];
                    if (nj != 0.0) vp[(k >> 2) * (CHUNK / 2) + (k & 3)] = (float)((double)cr.v[k] / nj);
                }
            }
        }
    }
}

// dense storage: one thread per column, rows in the reference's order (the same sequential fp64 sum)
__global__ void k_dense_normalize_columns(float *__restrict__ A, int64_t ld, int64_t nrows, int64_t ncols, double *__restrict__ norm)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double sum = 0.0;
    for (int64_t r = 0; r < nrows; ++r) {
        const float v = A[r * ld + j];
        sum += (double)(v * v);
    }
    const double nj = sqrt(sum);
    norm[j] = nj;
    if (nj != 0.0)
        for (int64_t r = 0; r < nrows; ++r) A[r * ld + j] = (float)((double)A[r * ld + j] / nj);
}

int normalize_columns_dev(tfx_ctx *ctx, TiledMatrix &m, double *d_norm)
{
    if (!m.valid) return fail(TFX_E_STATE, "normalize_columns: no matrix");
    hipStream_t s = ctx->stream;
    if (m.is_dense) {
        hipLaunchKernelGGL(k_dense_normalize_columns, dim3((unsigned)((m.ncols + 255) / 256)), dim3(256), 0, s, m.dense.p, m.ld, m.nrows, m.ncols, d_norm);
        TFX_HIP(hipGetLastError());
        return 0;
    }
    int shift = 0;                           // the integer grid coarsens by one bit per doubling of the row count beyond 2^17
    while (((int64_t)1 << (17 + shift)) < m.nrows) ++shift;
    if (shift > 24) return fail(TFX_E_ARG, "normalize_columns: more than 2^41 rows");
    const int nt = (int)m.h_tiles.size();
    DBuf<uint32_t> qmax;
    DBuf<unsigned long long> acc;
    TFX_TRY(qmax.alloc((size_t)m.ncols));
    TFX_TRY(acc.alloc((size_t)m.ncols * 2));
    TFX_HIP(hipMemsetAsync(qmax.p, 0, qmax.bytes(), s));
    TFX_HIP(hipMemsetAsync(acc.p, 0, acc.bytes(), s));
    const dim3 grid(8, (unsigned)std::max(1, std::min(nt, 32768)));
    if (nt > 0) {
        hipLaunchKernelGGL(k_colsq_max, grid, dim3(256), 0, s, m.tiles.p, nt, m.rec.p, m.TC, m.ncols, qmax.p);
        hipLaunchKernelGGL(k_colsq_sum, grid, dim3(256), 0, s, m.tiles.p, nt, m.rec.p, m.TC, m.ncols, qmax.p, acc.p, shift);
    }
    hipLaunchKernelGGL(k_colsq_finish, dim3((unsigned)((m.ncols + 255) / 256)), dim3(256), 0, s, qmax.p, acc.p, m.ncols, d_norm, shift);
    if (nt > 0)
        hipLaunchKernelGGL(k_div_by_norm<false>, grid, dim3(256), 0, s, m.tiles.p, nt, m.rec.p, m.chunk_row0.p, m.TC, m.RB, m.ncols, d_norm);
    TFX_HIP(hipGetLastError());
    m.vmax_stale = true;
    if (m.T && m.T->valid && !m.T->h_tiles.empty()) {        // the same quotients on the transposed copy: its rows are the columns of S
        TiledMatrix &t = *m.T;
        const int ntt = (int)t.h_tiles.size();
        hipLaunchKernelGGL(k_div_by_norm<true>, dim3(8, (unsigned)std::min(ntt, 32768)), dim3(256), 0, s, t.tiles.p, ntt, t.rec.p, t.chunk_row0.p, t.TC,
                           t.RB, t.nrows, d_norm);
        TFX_HIP(hipGetLastError());
        t.vmax_stale = true;
    }
    TFX_HIP(hipStreamSynchronize(s));        // (qmax / acc are freed on return)
    return 0;
}


}  // namespace tfx
