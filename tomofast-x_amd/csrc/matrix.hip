// matrix.hip - device-resident sensitivity matrix in the column-tiled compressed layout, and the two products
// that dominate an LSQR iteration:
//   forward  b (+)= S x    (t_sparse_matrix%add_mult_vector,       src/inversion/sparse_matrix.f90:316-329)
//   adjoint  b (+)= S^T x  (t_sparse_matrix%add_trans_mult_vector, src/inversion/sparse_matrix.f90:391-405)
//
// Layout (DESIGN.md "Data layout in HBM").  The reference keeps CSR with 4-byte values + 4-byte columns
// (8 B per non-zero).  Here S is cut into tiles of (RB <= 2048 rows) x (TC <= 16384 columns).  Inside a tile the
// entries are in (row, column) order and stored as streams:
//     vals[]  float : the value exactly as the reference stores it
//     d8[]    uint8 : 1..254 = column delta to the previous entry of the same row; 255 = "exception": the in-tile
//                     column is the next uint16 of the chunk's slice of exc[]; 0 = exception that also starts a new row
//     exc[]   uint16: absolute in-tile columns of the exception entries (row starts, gaps >= 255, first entry of a chunk)
// i.e. 5 B per non-zero + 2 B per exception (2-3 % of the entries for wavelet-compressed kernels: ~5.06 B/nnz).
// A row that is empty inside a tile but lies between two non-empty rows carries one marker entry (row start, value 0).
// A tile is padded to a multiple of 512 entries (one chunk = 64 lanes x 8 entries: a lane fetches its 8 deltas with one
// 8-byte load and its 8 values with two 16-byte loads).  cmeta[] gives each chunk the local row of the entry preceding
// it and its exception slice, so any wave can start at any chunk.
//
// Both products stream every tile once, coalesced, and keep the vector side of the product in LDS:
//   forward: the x tile (TC doubles, 128 KB) is staged in LDS, row sums are accumulated in LDS (RB doubles);
//   adjoint: the u rows of the block are staged in LDS, the column sums live in LDS (TC doubles) and are
//            written once.
// Decoding a chunk: exception / row-start ranks from ballot + mbcnt prefix counts, exception columns fetched with
// wave shuffles from two registers that hold the chunk's exc slice, columns from an in-lane running sum plus one
// segmented wave scan for the carry into each lane.  Short row segments are summed inside a lane, the segment tails
// are merged across lanes with one segmented wave reduction per chunk.
#include "common.h"
#include <algorithm>
#include <numeric>

namespace tfx {

thread_local std::string g_last_error;

size_t TiledMatrix::device_bytes() const
{
    return d8.bytes() + vals.bytes() + exc.bytes() + cmeta.bytes() + tiles.bytes() + fwd.bytes() + adj.bytes() +
           fwd_order.bytes() + adj_order.bytes() + fwd_partial.bytes() + adj_partial.bytes() + adj_nslots.bytes() +
           adj_pbase.bytes() + fwd_nslots.bytes() + fwd_pbase.bytes();
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
__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 3) & 7); }

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

// Scatter the entries of one row block into their tiles.  grid = (ceil(maxlen/256), nr).
__global__ void k_tile_scatter(const int32_t *__restrict__ cols, const float *__restrict__ vals,
                               const int32_t *__restrict__ nel, const int64_t *__restrict__ rowoff, int ntc, int TC, int nr,
                               const int32_t *__restrict__ pos, const int32_t *__restrict__ segoff,
                               const int64_t *__restrict__ tile_off, uint16_t *__restrict__ codes,
                               float *__restrict__ ovals)
{
    int r = blockIdx.y;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nel[r]) return;
    const int64_t src = rowoff[r] + j;
    int32_t c = cols[src];
    int t = c / TC;
    int p0 = pos[(int64_t)r * (ntc + 1) + t];
    int64_t dst = tile_off[t] + segoff[(int64_t)t * (nr + 1) + r] + (j - p0);
    codes[dst] = (uint16_t)((c - t * TC) | (j == p0 ? ROWSTART : 0));
    ovals[dst] = vals[src];
}

// Markers for empty rows strictly between the first and last non-empty row of a tile.  grid = (ntc), block 256.
__global__ void k_tile_markers(const int32_t *__restrict__ pos, int nr, int ntc, const int32_t *__restrict__ segoff,
                               const int32_t *__restrict__ first_ne, const int32_t *__restrict__ last_ne,
                               const int64_t *__restrict__ tile_off, uint16_t *__restrict__ codes,
                               float *__restrict__ ovals)
{
    int t = blockIdx.x;
    int f = first_ne[t], l = last_ne[t];
    if (f < 0) return;
    for (int r = f + 1 + threadIdx.x; r < l; r += blockDim.x) {
        int cnt = pos[(int64_t)r * (ntc + 1) + t + 1] - pos[(int64_t)r * (ntc + 1) + t];
        if (cnt == 0) {
            int64_t dst = tile_off[t] + segoff[(int64_t)t * (nr + 1) + r];
            codes[dst] = ROWSTART;
            ovals[dst] = 0.0f;
        }
    }
}

// chunk_row0[chunk] = local row of the entry just before the chunk (first chunk: first_ne - 1).
__global__ void k_chunk_row0(int nr, const int32_t *__restrict__ segoff, const int32_t *__restrict__ first_ne,
                             const int64_t *__restrict__ tile_off, const int32_t *__restrict__ tile_nchunks,
                             ChunkMeta *__restrict__ cmeta)
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
        cmeta[cbase + c].row0 = row;
    }
}


// ---- build-time uint16 codes (ROWSTART | column) -> stored byte stream + exception list ---------------------------
// One workgroup per column tile of the row block, one wave per chunk, a lane owns 8 consecutive entries.
__device__ __forceinline__ uint32_t code16(const uint4 &w, int k)
{
    const uint32_t v = (k < 2) ? w.x : (k < 4) ? w.y : (k < 6) ? w.z : w.w;
    return (v >> ((k & 1) * 16)) & 0xffffu;
}

// pass 1: d8 bytes + exception count per chunk; pass 2 (write_exc != 0): exception columns into exc[]
__global__ __launch_bounds__(256) void k_encode_d8(const int64_t *__restrict__ tile_off, const int32_t *__restrict__ tile_nchunks,
                                                   const int32_t *__restrict__ tile_cnt, const uint16_t *__restrict__ codes,
                                                   int64_t codes_base, uint8_t *__restrict__ d8, ChunkMeta *__restrict__ cmeta,
                                                   uint16_t *__restrict__ exc, int write_exc)
{
    const int t = blockIdx.x;
    const int nch = tile_nchunks[t];
    if (nch == 0) return;
    const int64_t off = tile_off[t];
    const int cnt = tile_cnt[t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int c = wave; c < nch; c += nwave) {
        const int64_t base = off + (int64_t)c * CHUNK;
        const uint4 w = *reinterpret_cast<const uint4 *>(codes + (base - codes_base) + lane * 8);
        const int nvalid = min(CHUNK, cnt - c * CHUNK);
        // column of the entry just before this lane's first one
        int prev = (int)(code16(w, 7) & COLMASK);
        prev = __shfl_up(prev, 1);
        uint32_t lo = 0, hi = 0;
        int nx = 0;
        uint32_t xcol[8];
        unsigned xmask = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t code = code16(w, k);
            const int col = (int)(code & COLMASK);
            const int idx = lane * 8 + k;
            const bool valid = idx < nvalid;
            const bool rs = (code & ROWSTART) != 0;
            const int delta = col - prev;
            const bool ex = valid && (idx == 0 || rs || delta > D8_MAX_DELTA || delta <= 0);
            const uint32_t byte = !valid ? 1u : (ex ? (rs ? (uint32_t)D8_EXC_ROWSTART : (uint32_t)D8_EXC) : (uint32_t)delta);
            if (k < 4) lo |= byte << (8 * k);
            else hi |= byte << (8 * (k - 4));
            xcol[k] = (uint32_t)col;
            if (ex) { xmask |= 1u << k; nx += 1; }
            prev = col;
        }
        // exclusive prefix of the exception counts over the lanes
        int incl = nx;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        const int total = __shfl(incl, 63);
        const int64_t ci = base / CHUNK;
        if (!write_exc) {
            *reinterpret_cast<uint2 *>(d8 + base + lane * 8) = make_uint2(lo, hi);
            if (lane == 0) cmeta[ci].nexc = total;
        } else {
            int pos = incl - nx;
            uint16_t *e = exc + cmeta[ci].exc_off;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (xmask & (1u << k)) e[pos++] = (uint16_t)xcol[k];
        }
    }
}

// exclusive scan of cmeta[c0 .. c0+n).nexc into exc_off (starting at `start`); *total_out = sum.  One block of 1024.
__global__ void k_exc_scan(ChunkMeta *__restrict__ cmeta, int64_t c0, int64_t n, int64_t start, int64_t *__restrict__ total_out)
{
    __shared__ int64_t part[1024];
    const int64_t per = (n + blockDim.x - 1) / blockDim.x;
    const int64_t b = (int64_t)threadIdx.x * per, e = min(b + per, n);
    int64_t s = 0;
    for (int64_t i = b; i < e; ++i) s += cmeta[c0 + i].nexc;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { const int64_t v = part[i]; part[i] = run; run += v; }
        *total_out = run;
    }
    __syncthreads();
    int64_t run = start + part[threadIdx.x];
    for (int64_t i = b; i < e; ++i) { cmeta[c0 + i].exc_off = run; run += cmeta[c0 + i].nexc; }
}

int matrix_begin(tfx_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz_upper)
{
    TiledMatrix &m = ctx->mat;
    m.valid = false;
    m.nrows = nrows;
    m.ncols = ncols;
    m.nnz = 0;
    if (nrows <= 0 || ncols <= 0) return fail(TFX_E_ARG, "matrix_begin: empty matrix %lld x %lld", (long long)nrows, (long long)ncols);
    m.TC = (int)std::min<int64_t>(TC_MAX, (ncols + 63) / 64 * 64);
    m.RB = (int)std::min<int64_t>(RB_MAX, (nrows + 63) / 64 * 64);
    m.ntc = (int)((ncols + m.TC - 1) / m.TC);
    m.nrb = (int)((nrows + m.RB - 1) / m.RB);
    int64_t markers = std::min<int64_t>(nnz_upper, (int64_t)nrows * m.ntc);
    int64_t cap = nnz_upper + markers + (int64_t)m.nrb * m.ntc * CHUNK + CHUNK;
    cap = (cap + CHUNK - 1) / CHUNK * CHUNK;
    TFX_TRY(m.d8.alloc((size_t)cap));
    TFX_TRY(m.vals.alloc((size_t)cap));
    TFX_TRY(m.cmeta.alloc((size_t)(cap / CHUNK)));
    TFX_TRY(m.exc.alloc((size_t)std::max<int64_t>(4096, cap / 16)));     // grown on demand
    m.n_entries = 0;
    m.n_exc = 0;
    m.h_tiles.clear();
    return 0;
}

// Appends the tiles of one row block.  Row r of the block has d_nel[r] entries starting at d_cols/d_vals +
// d_rowoff[r] (columns ascending, 0-based local); maxlen >= max d_nel.  row_begin must be a multiple of RB, nr <= RB.
int matrix_append_rows(tfx_ctx *ctx, int64_t row_begin, int64_t nr64, const int32_t *d_cols, const float *d_vals,
                       const int32_t *d_nel, const int64_t *d_rowoff, int64_t maxlen)
{
    TiledMatrix &m = ctx->mat;
    hipStream_t s = ctx->stream;
    int nr = (int)nr64;
    if (row_begin % m.RB != 0 || nr > m.RB || nr <= 0)
        return fail(TFX_E_ARG, "matrix_append_rows: rows [%lld, +%d) not aligned to row block %d", (long long)row_begin, nr, m.RB);
    int rb = (int)(row_begin / m.RB);
    int ntc = m.ntc;
    DBuf<int32_t> pos, segoff, first_ne, last_ne, tile_nch;
    DBuf<int64_t> tile_off;
    TFX_TRY(pos.alloc((size_t)nr * (ntc + 1)));
    TFX_TRY(segoff.alloc((size_t)ntc * (nr + 1)));
    TFX_TRY(first_ne.alloc(ntc));
    TFX_TRY(last_ne.alloc(ntc));
    TFX_TRY(tile_nch.alloc(ntc));
    TFX_TRY(tile_off.alloc(ntc));
    hipLaunchKernelGGL(k_tile_pos, dim3(nr), dim3(256), 0, s, d_cols, d_nel, d_rowoff, nr, ntc, m.TC, pos.p);
    hipLaunchKernelGGL(k_tile_scan, dim3(ntc), dim3(256), (size_t)(nr + 1) * sizeof(int32_t), s, pos.p, nr, ntc,
                       segoff.p, first_ne.p, last_ne.p);
    TFX_HIP(hipGetLastError());
    // tile totals -> host
    std::vector<int32_t> h_segoff_last(ntc), h_first(ntc);
    TFX_HIP(hipMemcpy2DAsync(h_segoff_last.data(), sizeof(int32_t), segoff.p + nr, (size_t)(nr + 1) * sizeof(int32_t),
                             sizeof(int32_t), ntc, hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> h_off(ntc);
    std::vector<int32_t> h_nch(ntc);
    int64_t cur = m.n_entries;
    int64_t nnz_block = 0;
    for (int t = 0; t < ntc; ++t) {
        int32_t cnt = h_segoff_last[t];
        int32_t nch = (cnt + CHUNK - 1) / CHUNK;
        h_off[t] = cur;
        h_nch[t] = nch;
        if (cnt > 0) {
            TileMeta tm;
            tm.off = cur;
            tm.nchunks = nch;
            tm.cnt = cnt;
            tm.t = t;
            tm.rb = rb;
            m.h_tiles.push_back(tm);
        }
        cur += (int64_t)nch * CHUNK;
        nnz_block += cnt;
    }
    if ((size_t)cur > m.d8.n)
        return fail(TFX_E_STATE, "tiled matrix capacity exceeded (%lld > %zu entries)", (long long)cur, m.d8.n);
    const int64_t nloc = cur - m.n_entries;
    if (nloc == 0) return 0;
    // build-time codes of this row block (uint16: ROWSTART | column), then the stored byte stream + exception list
    DBuf<uint16_t> codes;
    DBuf<int32_t> tile_cnt;
    DBuf<int64_t> d_total;
    TFX_TRY(codes.alloc((size_t)nloc));
    TFX_TRY(tile_cnt.alloc(ntc));
    TFX_TRY(d_total.alloc(1));
    uint16_t *codes_v = codes.p - m.n_entries;            // indexed with global entry numbers
    TFX_HIP(hipMemsetAsync(codes.p, 0, (size_t)nloc * sizeof(uint16_t), s));
    TFX_HIP(hipMemsetAsync(m.vals.p + m.n_entries, 0, (size_t)nloc * sizeof(float), s));
    TFX_HIP(hipMemcpyAsync(tile_off.p, h_off.data(), ntc * sizeof(int64_t), hipMemcpyHostToDevice, s));
    TFX_HIP(hipMemcpyAsync(tile_nch.p, h_nch.data(), ntc * sizeof(int32_t), hipMemcpyHostToDevice, s));
    TFX_HIP(hipMemcpyAsync(tile_cnt.p, h_segoff_last.data(), ntc * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (maxlen > 0)
        hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)((maxlen + 255) / 256), nr), dim3(256), 0, s, d_cols, d_vals,
                           d_nel, d_rowoff, ntc, m.TC, nr, pos.p, segoff.p, tile_off.p, codes_v, m.vals.p);
    hipLaunchKernelGGL(k_tile_markers, dim3(ntc), dim3(256), 0, s, pos.p, nr, ntc, segoff.p, first_ne.p, last_ne.p,
                       tile_off.p, codes_v, m.vals.p);
    hipLaunchKernelGGL(k_chunk_row0, dim3(ntc), dim3(256), 0, s, nr, segoff.p, first_ne.p, tile_off.p, tile_nch.p, m.cmeta.p);
    hipLaunchKernelGGL(k_encode_d8, dim3(ntc), dim3(256), 0, s, tile_off.p, tile_nch.p, tile_cnt.p, codes.p, m.n_entries,
                       m.d8.p, m.cmeta.p, (uint16_t *)nullptr, 0);
    const int64_t c0 = m.n_entries / CHUNK, nchunks = nloc / CHUNK;
    hipLaunchKernelGGL(k_exc_scan, dim3(1), dim3(1024), 0, s, m.cmeta.p, c0, nchunks, m.n_exc, d_total.p);
    TFX_HIP(hipGetLastError());
    int64_t h_total = 0;
    TFX_HIP(hipMemcpyAsync(&h_total, d_total.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    TFX_HIP(hipStreamSynchronize(s));
    if ((size_t)(m.n_exc + h_total) > m.exc.n) {          // grow the exception list (it is ~1 % of the matrix)
        DBuf<uint16_t> bigger;
        TFX_TRY(bigger.alloc((size_t)((m.n_exc + h_total) * 3 / 2 + 4096)));
        if (m.n_exc > 0) TFX_HIP(hipMemcpyAsync(bigger.p, m.exc.p, (size_t)m.n_exc * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        TFX_HIP(hipStreamSynchronize(s));
        std::swap(m.exc.p, bigger.p);
        std::swap(m.exc.n, bigger.n);
    }
    hipLaunchKernelGGL(k_encode_d8, dim3(ntc), dim3(256), 0, s, tile_off.p, tile_nch.p, tile_cnt.p, codes.p, m.n_entries,
                       m.d8.p, m.cmeta.p, m.exc.p, 1);
    TFX_HIP(hipGetLastError());
    TFX_HIP(hipStreamSynchronize(s));    // temporaries are freed on return
    m.n_entries = cur;
    m.n_exc += h_total;
    (void)nnz_block;
    return 0;
}

// Cuts the tile list into work items.  Forward: every item owns one partial-sum tile (pidx) of RB doubles, the
// items of row block rb are pbase[rb] .. pbase[rb]+nslots[rb]-1.  Adjoint: slot 0 of a column tile adds straight
// into y, slots >= 1 own partial tiles pbase[t] .. pbase[t]+nslots[t]-2 of TC doubles.
static void build_items(const std::vector<TileMeta> &tiles, bool forward, int64_t target, std::vector<WorkItem> &items,
                        std::vector<int32_t> &order, int &npartial, std::vector<int32_t> &nslots,
                        std::vector<int32_t> &pbase, int nkeys)
{
    std::vector<int32_t> idx(tiles.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) {
        int ka = forward ? tiles[a].rb : tiles[a].t, kb = forward ? tiles[b].rb : tiles[b].t;
        if (ka != kb) return ka < kb;
        int sa = forward ? tiles[a].t : tiles[a].rb, sb = forward ? tiles[b].t : tiles[b].rb;
        return sa < sb;
    });
    order = idx;
    items.clear();
    nslots.assign(nkeys, 0);
    pbase.assign(nkeys, 0);
    npartial = 0;
    std::vector<int64_t> item_sz;
    size_t i = 0;
    while (i < idx.size()) {
        int key = forward ? tiles[idx[i]].rb : tiles[idx[i]].t;
        size_t j = i;
        while (j < idx.size() && (forward ? tiles[idx[j]].rb : tiles[idx[j]].t) == key) ++j;
        // split [i, j) into runs of about `target` entries
        int slot = 0;
        size_t b = i;
        pbase[key] = npartial;
        while (b < j) {
            int64_t acc = 0;
            size_t e = b;
            while (e < j && (acc == 0 || acc + (int64_t)tiles[idx[e]].nchunks * CHUNK <= target)) {
                acc += (int64_t)tiles[idx[e]].nchunks * CHUNK;
                ++e;
            }
            WorkItem w;
            w.begin = (int32_t)b;
            w.end = (int32_t)e;
            w.key = key;
            w.slot = slot;
            w.pidx = forward ? npartial++ : (slot == 0 ? -1 : npartial++);
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
    TiledMatrix &m = ctx->mat;
    hipStream_t s = ctx->stream;
    int64_t real = 0;
    for (auto &t : m.h_tiles) real += t.cnt;     // includes empty-row markers; exact nnz is set by the caller
    TFX_TRY(m.tiles.alloc(std::max<size_t>(1, m.h_tiles.size())));
    if (!m.h_tiles.empty())
        TFX_HIP(hipMemcpyAsync(m.tiles.p, m.h_tiles.data(), m.h_tiles.size() * sizeof(TileMeta), hipMemcpyHostToDevice, s));
    // about 8 work items per CU, but never finer than 16 chunks
    int64_t target = std::max<int64_t>((int64_t)16 * CHUNK, m.n_entries / std::max(1, ctx->num_cu * 8));
    std::vector<int32_t> fo, ao, fns, ans, fpb, apb;
    int nfp = 0, nap = 0;
    build_items(m.h_tiles, true, target, m.h_fwd, fo, nfp, fns, fpb, m.nrb);
    build_items(m.h_tiles, false, target, m.h_adj, ao, nap, ans, apb, m.ntc);
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
    TFX_TRY(m.fwd_partial.alloc(std::max<size_t>(1, (size_t)nfp * m.RB)));
    TFX_TRY(m.adj_partial.alloc(std::max<size_t>(1, (size_t)nap * m.TC)));
    TFX_HIP(hipStreamSynchronize(s));
    m.adj_has_partials = nap > 0;
    m.nnz = real;
    m.valid = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// The two matrix kernels
// ------------------------------------------------------------------------------------------------------------
constexpr int SPMV_THREADS = 1024;
constexpr int SPMV_WAVES = SPMV_THREADS / 64;

// One chunk decoded for one lane: 8 consecutive entries.
struct Decoded {
    int col[8];        // in-tile column
    float v[8];
    unsigned rsmask;   // bit k: entry k starts a new row
    unsigned vmask;    // bit k: entry k exists (not tile padding)
    int cur;           // local row of the entry preceding this lane's first entry
};

__device__ __forceinline__ void decode_chunk(const uint8_t *__restrict__ d8, const float *__restrict__ vals,
                                             const uint16_t *__restrict__ exc, const ChunkMeta cm, int64_t base, int nvalid,
                                             int lane, Decoded &D)
{
    const uint2 db = *reinterpret_cast<const uint2 *>(d8 + base + lane * 8);
    const float4 va = *reinterpret_cast<const float4 *>(vals + base + lane * 8);
    const float4 vb = *reinterpret_cast<const float4 *>(vals + base + lane * 8 + 4);
    D.v[0] = va.x; D.v[1] = va.y; D.v[2] = va.z; D.v[3] = va.w;
    D.v[4] = vb.x; D.v[5] = vb.y; D.v[6] = vb.z; D.v[7] = vb.w;
    // the chunk's exception columns: two registers per lane cover 128 of them, the rest (rare) is read directly
    const uint16_t *ex = exc + cm.exc_off;
    const int e0 = (lane < cm.nexc) ? (int)ex[lane] : 0;
    const bool two = cm.nexc > 64;                         // wave-uniform
    const int e1 = (two && lane + 64 < cm.nexc) ? (int)ex[lane + 64] : 0;
    int byte[8];
    unsigned xmask = 0, rsmask = 0, vmask = 0;
    int pre_x = 0, pre_r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        byte[k] = (int)(((k < 4 ? db.x : db.y) >> (8 * (k & 3))) & 0xffu);
        const bool valid = lane * 8 + k < nvalid;
        const bool isx = valid && (byte[k] == D8_EXC || byte[k] == D8_EXC_ROWSTART);
        const bool rs = valid && byte[k] == D8_EXC_ROWSTART;
        const unsigned long long mx = __ballot(isx), mr = __ballot(rs);
        pre_x = __builtin_amdgcn_mbcnt_hi((uint32_t)(mx >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mx, pre_x));
        pre_r = __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, pre_r));
        if (valid) vmask |= 1u << k;
        if (isx) xmask |= 1u << k;
        if (rs) rsmask |= 1u << k;
    }
    // running column inside the lane; before the lane's first exception it is relative to the carry from lower lanes
    int acc = 0, eidx = pre_x;
    unsigned absmask = 0;
    bool seen = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int code = __shfl(e0, eidx & 63);
        if (two) { const int c1 = __shfl(e1, eidx & 63); if (eidx >= 64) code = c1; }
        if (xmask & (1u << k)) {
            if (eidx >= 128) code = (int)ex[eidx];
            acc = code;
            seen = true;
            eidx += 1;
        } else if (vmask & (1u << k)) {
            acc += byte[k];
        }
        D.col[k] = acc;
        if (seen) absmask |= 1u << k;
    }
    // carry into each lane: segmented inclusive scan of (seen, acc) over the lanes (lane 0 always starts absolute)
    int f = seen ? 1 : 0, val = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int of = __shfl_up(f, d), ov = __shfl_up(val, d);
        if (lane >= d && !f) { val += ov; f |= of; }
    }
    int carry = __shfl_up(val, 1);
    if (lane == 0) carry = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (!(absmask & (1u << k))) D.col[k] += carry;
    D.rsmask = rsmask;
    D.vmask = vmask;
    D.cur = cm.row0 + pre_r;
}

// forward: one workgroup = a run of tiles of one row block; its partial tile = sum over the run.
__global__ __launch_bounds__(SPMV_THREADS) void k_spmv_fwd(const WorkItem *__restrict__ items,
                                                             const int32_t *__restrict__ order,
                                                             const TileMeta *__restrict__ tiles,
                                                             const uint8_t *__restrict__ d8,
                                                             const float *__restrict__ vals,
                                                             const uint16_t *__restrict__ exc,
                                                             const ChunkMeta *__restrict__ cmeta,
                                                             const double *__restrict__ x, double *__restrict__ partial,
                                                             int64_t ncols, int TC, int RB)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *xs = lds;          // TC
    double *outs = lds + TC;   // RB
    const WorkItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < RB; i += SPMV_THREADS) outs[i] = 0.0;
    for (int ti = it.begin; ti < it.end; ++ti) {
        const TileMeta tm = tiles[order[ti]];
        __syncthreads();
        const int64_t col0 = (int64_t)tm.t * TC;
        const int ncol = (int)min((int64_t)TC, ncols - col0);
        for (int i = tid; i < TC; i += SPMV_THREADS) xs[swz(i)] = (i < ncol) ? x[col0 + i] : 0.0;
        __syncthreads();
        const int64_t cbase = tm.off / CHUNK;
        for (int c = wave; c < tm.nchunks; c += SPMV_WAVES) {
            Decoded D;
            decode_chunk(d8, vals, exc, cmeta[cbase + c], tm.off + (int64_t)c * CHUNK, min(CHUNK, tm.cnt - c * CHUNK), lane, D);
            int cur = D.cur;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (D.rsmask & (1u << k)) {
                    if (cur >= 0 && acc != 0.0) atomicAdd(&outs[cur], acc);
                    cur += 1;
                    acc = 0.0;
                }
                if (D.vmask & (1u << k)) acc = fma((double)D.v[k], xs[swz(D.col[k])], acc);
            }
            // merge the tails of lanes that end on the same row (equal rows are contiguous lanes)
            double sum = acc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double o = __shfl_down(sum, d);
                const int ocur = __shfl_down(cur, d);
                if (lane + d < 64 && ocur == cur) sum += o;
            }
            const int pcur = __shfl_up(cur, 1);
            if ((lane == 0 || pcur != cur) && cur >= 0 && sum != 0.0) atomicAdd(&outs[cur], sum);
        }
    }
    __syncthreads();
    double *dst = partial + (int64_t)it.pidx * RB;
    for (int i = tid; i < RB; i += SPMV_THREADS) dst[i] = outs[i];
}

// b[row] = (add ? b[row] : 0) + sum over the partial tiles of the row's block (fixed order: deterministic)
__global__ void k_fwd_reduce(const double *__restrict__ partial, const int32_t *__restrict__ nslots,
                             const int32_t *__restrict__ pbase, int RB, int64_t nrows, double *__restrict__ b, int add)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int rb = (int)(r / RB), lr = (int)(r - (int64_t)rb * RB);
    const int ns = nslots[rb], p0 = pbase[rb];
    double s = add ? b[r] : 0.0;
    for (int k = 0; k < ns; ++k) s += partial[(int64_t)(p0 + k) * RB + lr];
    b[r] = s;
}

// adjoint: one workgroup = a run of tiles of one column tile; slot 0 adds into y, the others write partial tiles.
__global__ __launch_bounds__(SPMV_THREADS) void k_spmv_adj(const WorkItem *__restrict__ items,
                                                             const int32_t *__restrict__ order,
                                                             const TileMeta *__restrict__ tiles,
                                                             const uint8_t *__restrict__ d8,
                                                             const float *__restrict__ vals,
                                                             const uint16_t *__restrict__ exc,
                                                             const ChunkMeta *__restrict__ cmeta,
                                                             const double *__restrict__ u, double *__restrict__ y,
                                                             double *__restrict__ partial, int64_t nrows, int64_t ncols,
                                                             int TC, int RB)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *acc = lds;        // TC
    double *us = lds + TC;    // RB
    const WorkItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < TC; i += SPMV_THREADS) acc[i] = 0.0;
    for (int ti = it.begin; ti < it.end; ++ti) {
        const TileMeta tm = tiles[order[ti]];
        __syncthreads();
        const int64_t row0 = (int64_t)tm.rb * RB;
        const int nrow = (int)min((int64_t)RB, nrows - row0);
        for (int i = tid; i < RB; i += SPMV_THREADS) us[i] = (i < nrow) ? u[row0 + i] : 0.0;
        __syncthreads();
        const int64_t cbase = tm.off / CHUNK;
        for (int c = wave; c < tm.nchunks; c += SPMV_WAVES) {
            Decoded D;
            decode_chunk(d8, vals, exc, cmeta[cbase + c], tm.off + (int64_t)c * CHUNK, min(CHUNK, tm.cnt - c * CHUNK), lane, D);
            int cur = D.cur;
            double uval = us[max(cur, 0)];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (D.rsmask & (1u << k)) {
                    cur += 1;
                    uval = us[cur];
                }
                const float v = D.v[k];
                if ((D.vmask & (1u << k)) && v != 0.0f) atomicAdd(&acc[swz(D.col[k])], (double)v * uval);
            }
        }
    }
    __syncthreads();
    const int64_t col0 = (int64_t)it.key * TC;
    const int ncol = (int)min((int64_t)TC, ncols - col0);
    if (it.slot == 0) {
        for (int i = tid; i < ncol; i += SPMV_THREADS) y[col0 + i] += acc[swz(i)];
    } else {
        double *dst = partial + (int64_t)it.pidx * TC;
        for (int i = tid; i < TC; i += SPMV_THREADS) dst[i] = acc[swz(i)];
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

static int set_lds_limit(const void *fn, size_t bytes)
{
    TFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

static void prof_begin(tfx_ctx *ctx)
{
    if (ctx->profile) (void)hipEventRecord(ctx->pev0, ctx->stream);
}
static void prof_end(tfx_ctx *ctx, int which)
{
    if (!ctx->profile) return;
    (void)hipEventRecord(ctx->pev1, ctx->stream);
    (void)hipEventSynchronize(ctx->pev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->pev0, ctx->pev1);
    ctx->prof_ms[which] += ms;
    ctx->prof_n[which] += 1;
}

int spmv_dev(tfx_ctx *ctx, const double *d_x, double *d_b, int add)
{
    TiledMatrix &m = ctx->mat;
    if (!m.valid) return fail(TFX_E_STATE, "spmv: no matrix");
    hipStream_t s = ctx->stream;
    const size_t lds = (size_t)(m.TC + m.RB) * sizeof(double);
    if (!m.h_fwd.empty()) {
        static size_t lds_set = 0;
        if (lds > lds_set) { TFX_TRY(set_lds_limit((const void *)k_spmv_fwd, lds)); lds_set = lds; }
        prof_begin(ctx);
        hipLaunchKernelGGL(k_spmv_fwd, dim3((unsigned)m.h_fwd.size()), dim3(SPMV_THREADS), lds, s, m.fwd.p, m.fwd_order.p,
                           m.tiles.p, m.d8.p, m.vals.p, m.exc.p, m.cmeta.p, d_x, m.fwd_partial.p, m.ncols, m.TC, m.RB);
        prof_end(ctx, 0);
        TFX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_fwd_reduce, dim3((unsigned)((m.nrows + 255) / 256)), dim3(256), 0, s, m.fwd_partial.p,
                       m.fwd_nslots.p, m.fwd_pbase.p, m.RB, m.nrows, d_b, add);
    TFX_HIP(hipGetLastError());
    return 0;
}

int spmtv_dev(tfx_ctx *ctx, const double *d_x, double *d_b, int add)
{
    TiledMatrix &m = ctx->mat;
    if (!m.valid) return fail(TFX_E_STATE, "spmtv: no matrix");
    hipStream_t s = ctx->stream;
    const size_t lds = (size_t)(m.TC + m.RB) * sizeof(double);
    if (!add) TFX_HIP(hipMemsetAsync(d_b, 0, (size_t)m.ncols * sizeof(double), s));
    if (!m.h_adj.empty()) {
        static size_t lds_set = 0;
        if (lds > lds_set) { TFX_TRY(set_lds_limit((const void *)k_spmv_adj, lds)); lds_set = lds; }
        prof_begin(ctx);
        hipLaunchKernelGGL(k_spmv_adj, dim3((unsigned)m.h_adj.size()), dim3(SPMV_THREADS), lds, s, m.adj.p, m.adj_order.p,
                           m.tiles.p, m.d8.p, m.vals.p, m.exc.p, m.cmeta.p, d_x, d_b, m.adj_partial.p, m.nrows, m.ncols,
                           m.TC, m.RB);
        prof_end(ctx, 1);
        TFX_HIP(hipGetLastError());
        if (m.adj_has_partials)
            hipLaunchKernelGGL(k_adj_reduce, dim3((unsigned)((m.ncols + 255) / 256)), dim3(256), 0, s, m.adj_partial.p,
                               m.adj_nslots.p, m.adj_pbase.p, m.TC, m.ncols, d_b);
        TFX_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace tfx
