// api.hip - extern "C" entry points of libtfx.so that are not kernels themselves: context, grid upload,
// CSR upload/download, the two products with host-or-device pointers, partitioning, timers.
#include "common.h"
#include <algorithm>
#include <cstdlib>

using namespace tfx;

extern "C" {

const char *tfx_last_error(void) { return g_last_error.c_str(); }

int tfx_create(int device, void *stream, tfx_ctx **out)
{
    if (!out) return fail(TFX_E_ARG, "tfx_create: null output");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(TFX_E_HIP, "tfx_create: no HIP device visible (%s) - the MI355X path has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= count) return fail(TFX_E_ARG, "tfx_create: device %d out of range [0,%d)", device, count);
    TFX_HIP(hipSetDevice(device));
    tfx_ctx *c = new tfx_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (const char *e = getenv("TFX_ADJ_COPY")) c->adj_copy = std::max(0, std::min(2, atoi(e)));   // like tfx_debug_set "adj_copy"
    if (const char *e = getenv("TFX_COMM_INIT_TIMEOUT")) c->comm_init_timeout_s = atof(e);
    if (const char *e = getenv("TFX_TR_PANEL_ENTRIES")) { if (atof(e) > 0) c->tr_panel_entries = atof(e); }     // like the debug keys "tr_panel_entries" /
    if (const char *e = getenv("TFX_TR_POS_BUDGET")) { if (atof(e) > 0) c->tr_pos_budget = atof(e); }           //   "tr_pos_budget": sweeps force many small panels
    if (const char *e = getenv("TFX_FWD_RUN")) c->fwd_run = std::min(16, std::max(1, atoi(e)));
    if (const char *e = getenv("TFX_LSQR_MERGE_TAIL")) c->lsqr_merge_tail = atoi(e) != 0;
    if (const char *e = getenv("TFX_ADJ_COPY_MIN_NNZ")) c->adj_copy_min_nnz = atoll(e);
    if (const char *e = getenv("TFX_CHAIN_UNDER_WAVELET")) c->chain_under_wavelet = atoi(e) != 0;
    if (const char *e = getenv("TFX_BUILD_OVERLAP")) c->build_overlap = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("TFX_GEN_WGS_PER_CU")) c->gen_wgs_per_cu = atoi(e);
    if (const char *e = getenv("TFX_WAVE_PIPE")) c->wave_pipe = std::max(0, std::min(8, atoi(e)));
    if (const char *e = getenv("TFX_GEN_AFTER_WAVELET")) c->gen_after_wavelet = std::max(0, std::min(3, atoi(e)));
    TFX_HIP(hipEventCreate(&c->ev0));
    TFX_HIP(hipEventCreate(&c->ev1));
    TFX_HIP(hipEventCreate(&c->pev0));
    TFX_HIP(hipEventCreate(&c->pev1));
    *out = c;
    return 0;
}

// number of HIP devices visible to this process (rank -> device mapping of a host that launches one process per GPU)
int tfx_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

// synchronous copy between host and device memory (either direction), ordered after the work already queued on the ctx stream -
// what a host-language all-reduce hook needs to stage a device buffer through MPI
int tfx_copy(tfx_ctx *ctx, void *dst, const void *src, int64_t bytes)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !dst || !src || bytes < 0) return fail(TFX_E_ARG, "tfx_copy: bad arguments");
    TFX_HIP(hipSetDevice(ctx->device));
    TFX_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, ctx->stream));
    TFX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// raw device buffers for a host that moves packed matrix pieces itself (tfx_rowstore_pack -> send -> tfx_matrix_append_rows)
int tfx_device_malloc(tfx_ctx *ctx, int64_t bytes, void **ptr_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !ptr_out || bytes < 0) return fail(TFX_E_ARG, "tfx_device_malloc: bad arguments");
    TFX_HIP(hipSetDevice(ctx->device));
    *ptr_out = nullptr;
    if (bytes == 0) return 0;
    hipError_t e = hipMalloc(ptr_out, (size_t)bytes);
    if (e != hipSuccess) return fail(TFX_E_HIP, "hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
    return 0;
}

int tfx_device_free(tfx_ctx *ctx, void *ptr)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    TFX_HIP(hipSetDevice(ctx->device));
    if (ptr) {
        (void)hipStreamSynchronize(ctx->stream);
        TFX_HIP(hipFree(ptr));
    }
    return 0;
}

int lsqr_free(tfx_ctx *ctx);   // lsqr.hip

int tfx_destroy(tfx_ctx *ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    lsqr_free(ctx);
    (void)tfx_comm_destroy(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->pev0) (void)hipEventDestroy(ctx->pev0);
    if (ctx->pev1) (void)hipEventDestroy(ctx->pev1);
    tfx::prof_drain(ctx);
    for (auto &pp : ctx->prof_free) { (void)hipEventDestroy(pp.a); (void)hipEventDestroy(pp.b); }
    delete ctx;
    return 0;
}

int tfx_device_info(tfx_ctx *ctx, char *name, int len, int64_t *hbm_bytes)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    hipDeviceProp_t prop;
    TFX_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && len > 0) {
        snprintf(name, (size_t)len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return prop.multiProcessorCount;
}

int tfx_set_allreduce(tfx_ctx *ctx, tfx_allreduce_fn fn, void *user, int rank, int nranks)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(TFX_E_ARG, "bad rank %d / %d", rank, nranks);
    if (nranks > 1 && !fn) return fail(TFX_E_ARG, "nranks > 1 needs an all-reduce hook");
    ctx->allreduce = fn;
    ctx->allreduce_user = user;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return 0;
}

int tfx_set_grid(tfx_ctx *ctx, int nx, int ny, int nz, const double *X1, const double *X2, const double *Y1,
                 const double *Y2, const double *Z1, const double *Z2)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    if (nx <= 0 || ny <= 0 || nz <= 0) return fail(TFX_E_ARG, "bad grid size %d %d %d", nx, ny, nz);
    TFX_HIP(hipSetDevice(ctx->device));
    const double *src[6] = {X1, X2, Y1, Y2, Z1, Z2};
    int64_t N = (int64_t)nx * ny * nz;
    for (int i = 0; i < 6; ++i) {
        if (!src[i]) return fail(TFX_E_ARG, "tfx_set_grid: null array %d", i);
        TFX_TRY(ctx->grid[i].alloc((size_t)N));
        TFX_TRY(copy_any(ctx->grid[i].p, src[i], (size_t)N * sizeof(double), ctx->stream));
    }
    ctx->nx = nx;
    ctx->ny = ny;
    ctx->nz = nz;
    ctx->N = N;
    return detect_tensor_grid(ctx);
}

int tfx_debug_set(tfx_ctx *ctx, const char *key, int value)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !key) return fail(TFX_E_ARG, "null argument");
    if (!strcmp(key, "force_general_prism")) {
        ctx->force_general_prism = value != 0;
        if (ctx->N) return detect_tensor_grid(ctx);
        return 0;
    }
    if (!strcmp(key, "tensor_grid")) return ctx->tensor_grid ? 1 : 0;     // query
    if (!strcmp(key, "band_select_min_cells")) {                           // value < 0: never use the band select
        ctx->band_min_n = value < 0 ? INT64_MAX : (int64_t)value;
        return 0;
    }
    if (!strcmp(key, "deterministic")) return 0;   // (kept for older hosts) the two products are always reproducible: matrix.hip k_spmv_fwd / k_spmv_adj
    if (!strcmp(key, "adj_copy")) {             // transposed copy for the adjoint of matrices finished from now on: 0 never, 1 always, 2 automatic
        ctx->adj_copy = std::max(0, std::min(2, value));
        return 0;
    }
    if (!strcmp(key, "adj_copy_min_nnz")) {     // automatic mode: matrices of at least this many stored entries get the copy
        ctx->adj_copy_min_nnz = value;
        return 0;
    }
    if (!strcmp(key, "adj_copy_build_ms")) return (int)(1e3 * ctx->selmat().copy_build_s);     // query: wall clock of the selected matrix's copy
    if (!strcmp(key, "tr_panel_entries")) {     // entries per panel of the transposition (0 = default)
        ctx->tr_panel_entries = value > 0 ? (double)value : 9.0e8;
        return 0;
    }
    if (!strcmp(key, "tr_pos_budget")) {        // ints of per-row tile index a panel may use (0 = default): small values force full-height / banded panels
        ctx->tr_pos_budget = value > 0 ? (double)value : 1.5e8;
        return 0;
    }
    if (!strcmp(key, "drop_adj_copy")) {        // gives up the transposed copy of the selected matrix (the adjoint then runs on the tiles of S)
        TiledMatrix &m = ctx->selmat();
        if (m.T) {
            (void)hipDeviceSynchronize();
            delete m.T;
            m.T = nullptr;
            m.vmax_stale = true;
        }
        return 0;
    }
    if (!strcmp(key, "has_adj_copy")) return (ctx->selmat().T && ctx->selmat().T->valid) ? 1 : 0;     // query
    if (!strcmp(key, "chunk_exponent_span")) {  // diagnostics: per mille of the chunks whose non-zero values span <= `value` binades
        int64_t fit = 0, total = 0;
        unsigned int hist[34];
        TFX_TRY(chunk_exponent_stats(ctx, ctx->selmat(), value, &fit, &total, hist));
        fprintf(stderr, "[tfx] chunk exponent spans (binades: chunks):");
        for (int i = 0; i < 34; ++i) if (hist[i]) fprintf(stderr, " %d:%u", i, hist[i]);
        fprintf(stderr, "\n");
        return total > 0 ? (int)(1000.0 * (double)fit / (double)total) : 0;
    }
    if (!strcmp(key, "gen_after_wavelet")) {
        ctx->gen_after_wavelet = std::max(0, std::min(3, value));        // axis passes of the wavelet transform ahead of the next generator
        return 0;
    }
    if (!strcmp(key, "wave_pipe")) {
        ctx->wave_pipe = std::max(0, std::min(8, value));
        return ctx->wave_pipe;
    }
    if (!strcmp(key, "gen_wgs_per_cu")) {
        ctx->gen_wgs_per_cu = value;
        return 0;
    }
    if (!strcmp(key, "chain_under_wavelet")) {
        ctx->chain_under_wavelet = value != 0;
        return 0;
    }
    if (!strcmp(key, "build_overlap")) {        // 0 one stream; 1 overlapped for builds of at least 8 batches; 2 always overlapped
        ctx->build_overlap = std::max(0, std::min(2, value));
        return 0;
    }
    if (!strcmp(key, "items_per_cu")) {         // work items per CU for matrices finished from now on
        ctx->items_per_cu = value > 0 ? value : 16;
        return 0;
    }
    if (!strcmp(key, "refinish")) {             // rebuild the work lists of the selected matrix with the current knobs
        if (!ctx->selmat().valid || ctx->selmat().is_dense) return fail(TFX_E_STATE, "refinish: no tiled matrix");
        const int64_t nnz = ctx->selmat().nnz;
        TiledMatrix *keep = ctx->target;            // (the target may still be the constraint matrix or the other slot: ADVICE r2)
        ctx->target = &ctx->selmat();
        const int rc = matrix_finish(ctx);
        ctx->target = keep;
        TFX_TRY(rc);
        ctx->selmat().nnz = nnz;
        return 0;
    }
    if (!strcmp(key, "force_collectives")) {    // issue the collectives of the multi-rank path even on one rank
        ctx->force_collectives = value != 0;
        return 0;
    }
    if (!strcmp(key, "comm_init_timeout_s")) {  // how long tfx_comm_init_rccl waits for the rendezvous (<= 0: for ever)
        ctx->comm_init_timeout_s = (double)value;
        return 0;
    }
    if (!strcmp(key, "fwd_run")) {              // chunks per run of the forward kernel (>= 1); the sums stay reproducible for a fixed value
        ctx->fwd_run = std::min(16, std::max(1, value));
        return 0;
    }
    if (!strcmp(key, "lsqr_merge_tail")) {      // 1 (default): k_update_xw_next, 0: k_update_xw + k_scale + k_cons_forward as separate launches; same bits
        ctx->lsqr_merge_tail = value != 0;
        return 0;
    }
    if (!strcmp(key, "fwd_group")) {            // row blocks per forward super block for matrices finished from now on (0 = automatic)
        ctx->fwd_group_override = value;
        return 0;
    }
    if (!strcmp(key, "band_batches")) return (int)std::min<int64_t>(ctx->band_batches, INT32_MAX);       // queries
    if (!strcmp(key, "band_fallbacks")) return (int)std::min<int64_t>(ctx->band_fallbacks, INT32_MAX);
    return fail(TFX_E_ARG, "unknown debug key %s", key);
}

// Joint inversion (two problems on one grid): slot 0 / 1 selects which sensitivity matrix the build / upload / download / info /
// free / product / calc_data entry points act on; LSQR solves with S = blockdiag(slot 0, slot 1) when slot 1 holds a matrix.
int tfx_select_problem(tfx_ctx *ctx, int slot)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    if (slot != 0 && slot != 1) return fail(TFX_E_ARG, "tfx_select_problem: slot must be 0 or 1");
    ctx->slot = slot;
    ctx->target = &ctx->selmat();
    return 0;
}

// ---- matrix upload / download ------------------------------------------------------------------------------
static int upload_csr_into(tfx_ctx *ctx, TiledMatrix &dst, int64_t nrows, int64_t ncols, const int64_t *rowptr,
                           const int32_t *cols, const float *vals, bool require_ascending);

int tfx_matrix_upload_csr(tfx_ctx *ctx, int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols,
                          const float *vals)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !rowptr) return fail(TFX_E_ARG, "tfx_matrix_upload_csr: null argument");
    return upload_csr_into(ctx, ctx->selmat(), nrows, ncols, rowptr, cols, vals, true);
}

// General constraint rows (matrix_cons of joint_inverse_problem.F90:332,544): same storage and kernels as S.
int tfx_cons_upload_csr(tfx_ctx *ctx, int64_t nrows, const int64_t *rowptr, const int32_t *cols, const float *vals,
                        const double *rhs)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !rowptr || !rhs) return fail(TFX_E_ARG, "tfx_cons_upload_csr: null argument");
    if (!ctx->mat.valid) return fail(TFX_E_STATE, "tfx_cons_upload_csr: upload / build S first (it defines ncolumns)");
    TFX_TRY(upload_csr_into(ctx, ctx->cons, nrows, ctx->total_cols(), rowptr, cols, vals, true));
    TFX_TRY(ctx->cons_rhs.ensure((size_t)nrows));
    TFX_TRY(copy_any(ctx->cons_rhs.p, rhs, (size_t)nrows * sizeof(double), ctx->stream));
    return 0;
}

int tfx_cons_clear(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    (void)hipStreamSynchronize(ctx->stream);
    ctx->cons.release_storage();
    return 0;
}

static int upload_csr_into(tfx_ctx *ctx, TiledMatrix &dst, int64_t nrows, int64_t ncols, const int64_t *rowptr,
                           const int32_t *cols, const float *vals, bool require_ascending)
{
    (void)require_ascending;
    struct Retarget {
        tfx_ctx *c;
        Retarget(tfx_ctx *cc, TiledMatrix *t) : c(cc) { c->target = t; }
        ~Retarget() { c->target = &c->selmat(); }
    } retarget(ctx, &dst);
    if (ncols > 0x7fffffffLL || nrows > 0x7fffffffLL) return fail(TFX_E_ARG, "matrix dimension exceeds int32");
    TFX_HIP(hipSetDevice(ctx->device));
    int64_t nnz = rowptr[nrows];
    if (nnz < 0) return fail(TFX_E_ARG, "negative nnz");
    if (nnz > 0 && (!cols || !vals)) return fail(TFX_E_ARG, "null cols/vals");
    // validation like t_sparse_matrix%validate (sparse_matrix.f90:188-207) + ascending columns within a row
    for (int64_t r = 0; r < nrows; ++r) {
        if (rowptr[r + 1] < rowptr[r]) return fail(TFX_E_ARG, "rowptr not monotone at row %lld", (long long)r);
        for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            if (cols[k] < 1 || cols[k] > ncols)
                return fail(TFX_E_ARG, "Sparse matrix column-index validation failed! row %lld col %d", (long long)r, cols[k]);
            if (k > rowptr[r] && cols[k] <= cols[k - 1])
                return fail(TFX_E_ARG, "columns must ascend within a row (row %lld)", (long long)r);
        }
    }
    TFX_TRY(matrix_begin(ctx, nrows, ncols, nnz));
    TiledMatrix &m = dst;
    hipStream_t s = ctx->stream;
    for (int rb = 0; rb < m.nrb; ++rb) {
        int64_t r0 = (int64_t)rb * m.RB, r1 = std::min<int64_t>(nrows, r0 + m.RB);
        int nr = (int)(r1 - r0);
        int64_t e0 = rowptr[r0], e1 = rowptr[r1];
        int64_t ne = e1 - e0;
        std::vector<int32_t> hc((size_t)ne), hn(nr);
        std::vector<int64_t> ho(nr);
        int64_t maxlen = 0;
        for (int r = 0; r < nr; ++r) {
            ho[r] = rowptr[r0 + r] - e0;
            hn[r] = (int32_t)(rowptr[r0 + r + 1] - rowptr[r0 + r]);
            maxlen = std::max<int64_t>(maxlen, hn[r]);
        }
        for (int64_t k = 0; k < ne; ++k) hc[(size_t)k] = cols[e0 + k] - 1;     // 0-based local
        DBuf<int32_t> dc, dn;
        DBuf<float> dv;
        DBuf<int64_t> dof;
        TFX_TRY(dc.alloc((size_t)std::max<int64_t>(1, ne)));
        TFX_TRY(dv.alloc((size_t)std::max<int64_t>(1, ne)));
        TFX_TRY(dn.alloc(nr));
        TFX_TRY(dof.alloc(nr));
        if (ne > 0) {
            TFX_HIP(hipMemcpyAsync(dc.p, hc.data(), (size_t)ne * sizeof(int32_t), hipMemcpyHostToDevice, s));
            TFX_HIP(hipMemcpyAsync(dv.p, vals + e0, (size_t)ne * sizeof(float), hipMemcpyHostToDevice, s));
        }
        TFX_HIP(hipMemcpyAsync(dn.p, hn.data(), nr * sizeof(int32_t), hipMemcpyHostToDevice, s));
        TFX_HIP(hipMemcpyAsync(dof.p, ho.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, s));
        TFX_TRY(matrix_append_rows(ctx, r0, nr, dc.p, dv.p, dn.p, dof.p, maxlen));
    }
    TFX_TRY(matrix_finish(ctx));
    m.nnz = nnz;
    return 0;
}

// How the selected matrix is stored: bytes the entry streams (values, column stream, row-start masks) hold per stored entry - what a
// product streams per entry and launch -, the number of stored entries (non-zeros + empty-row markers; the pad entries that fill a
// tile's last chunk are part of `stream_bytes` only), the bytes of those streams, and whether the adjoint product runs on a transposed copy of the tiles (then it streams that copy instead).
int tfx_matrix_format(tfx_ctx *ctx, double *bytes_per_entry, int64_t *stored_entries, int64_t *stream_bytes, int *adjoint_copy)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    const TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "no matrix");
    int64_t stored = 0;
    for (const TileMeta &t : m.h_tiles) stored += t.cnt;
    const int64_t bytes = m.is_dense ? (int64_t)m.dense.bytes()
                                     : (m.n_entries / CHUNK) * (int64_t)REC_BYTES;
    if (m.is_dense) stored = m.nrows * m.ncols;
    if (bytes_per_entry) *bytes_per_entry = m.is_dense ? 4.0 : 4.0 + 4.0 * SLOT_WORDS / CHUNK + 8.0 * MASK_WORDS / CHUNK;
    if (stored_entries) *stored_entries = stored;
    if (stream_bytes) *stream_bytes = bytes;
    if (adjoint_copy) *adjoint_copy = (m.T && m.T->valid) ? 1 : 0;
    return 0;
}

int tfx_matrix_info(tfx_ctx *ctx, int64_t *nrows, int64_t *ncols, int64_t *nnz, int64_t *device_bytes)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    const TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "no matrix");
    if (nrows) *nrows = m.nrows;
    if (ncols) *ncols = m.ncols;
    if (nnz) *nnz = m.nnz;
    if (device_bytes) *device_bytes = (int64_t)m.device_bytes();
    return 0;
}

// Host-side decode of the tiled layout back to CSR (tests, SENSIT-format writers; not on the hot path).
int tfx_matrix_download_csr(tfx_ctx *ctx, int64_t *rowptr, int32_t *cols, float *vals)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !rowptr) return fail(TFX_E_ARG, "null argument");
    TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "no matrix");
    TFX_HIP(hipSetDevice(ctx->device));
    if (m.is_dense) {                       // every entry is a stored entry (sensitivity_gravmag.F90:287-295)
        std::vector<float> row((size_t)m.ld);
        for (int64_t r = 0; r < m.nrows; ++r) {
            rowptr[r] = r * m.ncols;
            TFX_HIP(hipMemcpy(row.data(), m.dense.p + r * m.ld, (size_t)m.ncols * sizeof(float), hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < m.ncols; ++c) {
                if (cols) cols[r * m.ncols + c] = (int32_t)(c + 1);
                if (vals) vals[r * m.ncols + c] = row[(size_t)c];
            }
        }
        rowptr[m.nrows] = m.nrows * m.ncols;
        return 0;
    }
    const size_t nch = (size_t)(m.n_entries / CHUNK);
    std::vector<char> hrec(nch * REC_BYTES);            // the chunk records: values, 12-bit slots, row-start masks (common.h)
    if (m.n_entries > 0) TFX_HIP(hipMemcpy(hrec.data(), m.rec.p, hrec.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> hrow0(nch);
    if (!hrow0.empty())
        TFX_HIP(hipMemcpy(hrow0.data(), m.chunk_row0.p, hrow0.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    // entry e of the streams: its 12-bit slot and its row-start flag (matrix.hip: put_entry)
    auto slot_at = [&](int64_t e) -> uint32_t {
        const int64_t chunk = e >> 9;
        const int i = (int)(e & (CHUNK - 1)), lane = i >> 3, k = i & 7, bit = 12 * k, wi = bit >> 5, sh = bit & 31;
        const uint32_t *w = chunk_slots((const char *)hrec.data(), chunk) + lane * 3;
        uint32_t v = w[wi] >> sh;
        if (sh > 20) v |= w[wi + 1] << (32 - sh);
        return v & 0xfffu;
    };
    auto flag_at = [&](int64_t e) -> bool {
        const int64_t chunk = e >> 9;
        const int i = (int)(e & (CHUNK - 1)), lane = i >> 3, k = i & 7;
        return (chunk_masks((const char *)hrec.data(), chunk)[k] >> lane) & 1ull;
    };
    // tiles sorted by (rb, t)
    std::vector<TileMeta> tl = m.h_tiles;
    std::sort(tl.begin(), tl.end(), [](const TileMeta &a, const TileMeta &b) { return a.rb != b.rb ? a.rb < b.rb : a.t < b.t; });
    // pass 0: counts, pass 1: fill
    std::vector<int64_t> cnt((size_t)m.nrows + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int64_t> fill;
        if (pass == 1) {
            int64_t run = 0;
            for (int64_t r = 0; r < m.nrows; ++r) { int64_t c = cnt[(size_t)r]; rowptr[r] = run; run += c; }
            rowptr[m.nrows] = run;
            fill.assign(rowptr, rowptr + m.nrows);
        }
        for (const TileMeta &tm : tl) {
            int cur = hrow0[(size_t)(tm.off / CHUNK)];
            for (int32_t e = 0; e < tm.cnt; ++e) {
                const bool flag = flag_at(tm.off + e);
                const uint32_t slot = slot_at(tm.off + e);
                if (flag) cur += 1;
                float v = chunk_vals((const char *)hrec.data(), (tm.off + e) >> 9)[val_pos(tm.off + e) & (CHUNK - 1)];
                bool marker = flag && v == 0.0f && slot == 0 && (e + 1 == tm.cnt || flag_at(tm.off + e + 1));
                // a marker is indistinguishable from a stored exact zero in column 0 of the tile that is alone in
                // its row segment; the reference never stores zeros (sparse_matrix.f90:219, threshold >= 1e-30)
                if (marker) continue;
                int64_t row = (int64_t)tm.rb * m.RB + cur;
                if (row < 0 || row >= m.nrows) return fail(TFX_E_STATE, "corrupt tile: row %lld", (long long)row);
                if (pass == 0) cnt[(size_t)row] += 1;
                else {
                    int64_t p = fill[(size_t)row]++;
                    if (cols) cols[p] = (int32_t)((int64_t)tm.t * m.TC + col_slot((int)slot) + 1);
                    if (vals) vals[p] = v;
                }
            }
        }
    }
    return 0;
}

int tfx_matrix_free(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    (void)hipStreamSynchronize(ctx->stream);
    ctx->selmat().release_storage();
    return 0;
}

// Rows of the selected matrix times a per-row factor: entry (r, c) becomes value * (float)scale[r] in fp32 - the scaling
// read_sensitivity_kernel applies on reload (sensitivity_gravmag.F90:834-843) to a kernel stored unscaled.
int tfx_matrix_scale_rows(tfx_ctx *ctx, const double *scale)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !scale) return fail(TFX_E_ARG, "tfx_matrix_scale_rows: null argument");
    TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "tfx_matrix_scale_rows: no matrix");
    TFX_HIP(hipSetDevice(ctx->device));
    std::vector<double> hs((size_t)m.nrows);
    TFX_TRY(copy_any(hs.data(), scale, (size_t)m.nrows * sizeof(double), ctx->stream));
    std::vector<float> hf((size_t)m.nrows);
    for (int64_t r = 0; r < m.nrows; ++r) hf[(size_t)r] = (float)hs[(size_t)r];
    DBuf<float> df;
    TFX_TRY(df.alloc((size_t)m.nrows));
    TFX_TRY(copy_any(df.p, hf.data(), (size_t)m.nrows * sizeof(float), ctx->stream));
    TFX_TRY(scale_rows_dev(ctx, m, df.p));
    TFX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int tfx_matrix_reserve(tfx_ctx *ctx, int64_t nnz_upper)
{
    if (!ctx || nnz_upper < 0) return fail(TFX_E_ARG, "tfx_matrix_reserve: bad arguments");
    ctx->reserve_nnz = nnz_upper;
    ctx->reserve_slot = ctx->slot;
    return 0;
}

// t_sparse_matrix%normalize_columns, src/inversion/sparse_matrix.f90:414-443
int tfx_matrix_normalize_columns(tfx_ctx *ctx, double *column_norm_out)
{
    tfx::AllocScope alloc_scope_(ctx);
    if (!ctx || !column_norm_out) return fail(TFX_E_ARG, "tfx_matrix_normalize_columns: null argument");
    TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "tfx_matrix_normalize_columns: no matrix");
    TFX_HIP(hipSetDevice(ctx->device));
    DBuf<double> dn;
    TFX_TRY(dn.alloc((size_t)m.ncols));
    TFX_TRY(normalize_columns_dev(ctx, m, dn.p));
    TFX_TRY(copy_any(column_norm_out, dn.p, (size_t)m.ncols * sizeof(double), ctx->stream));
    return 0;
}

// get_load_balancing_nelements, src/forward/gravmag/sensitivity_gravmag.F90:470-524 (exact integer rule)
int tfx_partition_columns(const int32_t *nnz, int64_t N, int P, int32_t *nel_at, int64_t *nnz_at)
{
    if (!nnz || !nel_at || !nnz_at || P < 1 || N < P) return fail(TFX_E_ARG, "tfx_partition_columns: bad arguments");
    int64_t total = 0;
    for (int64_t p = 0; p < N; ++p) total += nnz[p];
    std::vector<int64_t> cum(P);
    int64_t run = 0;
    for (int c = 0; c < P; ++c) {
        run += total / P + (c == P - 1 ? total % P : 0);
        cum[c] = run;
        nel_at[c] = 0;
        nnz_at[c] = 0;
    }
    int cpu = 0;
    int64_t nnz_new = 0, sum = 0;
    int32_t nel_new = 0;
    for (int64_t p = 0; p < N; ++p) {
        nnz_new += nnz[p];
        sum += nnz[p];
        nel_new += 1;
        if ((cpu < P - 1 && sum >= cum[cpu]) || p == N - 1) {
            if (cpu >= P) return fail(TFX_E_NUMERIC, "Wrong cpu in get_load_balancing_nelements!");
            nnz_at[cpu] = nnz_new;
            nel_at[cpu] = nel_new;
            nnz_new = 0;
            nel_new = 0;
            ++cpu;
        }
    }
    if (cpu != P) return fail(TFX_E_NUMERIC, "Wrong cpu in get_load_balancing_nelements! (%d of %d parts filled)", cpu, P);
    return 0;
}

// ---- products with host-or-device vectors ------------------------------------------------------------------
static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

int tfx_spmv(tfx_ctx *ctx, const double *x, double *b, int add)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !x || !b) return fail(TFX_E_ARG, "tfx_spmv: null argument");
    TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "tfx_spmv: no matrix");
    TFX_HIP(hipSetDevice(ctx->device));
    const double *dx = x;
    double *db = b;
    if (!is_device_ptr(x)) {
        TFX_TRY(ctx->vx.ensure((size_t)m.ncols));
        TFX_TRY(copy_any(ctx->vx.p, x, (size_t)m.ncols * sizeof(double), ctx->stream));
        dx = ctx->vx.p;
    }
    bool hostb = !is_device_ptr(b);
    if (hostb) {
        TFX_TRY(ctx->vb.ensure((size_t)m.nrows));
        if (add) TFX_TRY(copy_any(ctx->vb.p, b, (size_t)m.nrows * sizeof(double), ctx->stream));
        db = ctx->vb.p;
    }
    TFX_TRY(spmv_dev(ctx, dx, db, add));
    if (hostb) TFX_TRY(copy_any(b, db, (size_t)m.nrows * sizeof(double), ctx->stream));
    return 0;
}

int tfx_spmtv(tfx_ctx *ctx, const double *x, double *b, int add)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !x || !b) return fail(TFX_E_ARG, "tfx_spmtv: null argument");
    TiledMatrix &m = ctx->selmat();
    if (!m.valid) return fail(TFX_E_STATE, "tfx_spmtv: no matrix");
    TFX_HIP(hipSetDevice(ctx->device));
    const double *dx = x;
    double *db = b;
    if (!is_device_ptr(x)) {
        TFX_TRY(ctx->vb.ensure((size_t)m.nrows));
        TFX_TRY(copy_any(ctx->vb.p, x, (size_t)m.nrows * sizeof(double), ctx->stream));
        dx = ctx->vb.p;
    }
    bool hostb = !is_device_ptr(b);
    if (hostb) {
        TFX_TRY(ctx->vx.ensure((size_t)m.ncols));
        if (add) TFX_TRY(copy_any(ctx->vx.p, b, (size_t)m.ncols * sizeof(double), ctx->stream));
        db = ctx->vx.p;
    }
    TFX_TRY(spmtv_dev(ctx, dx, db, add));
    if (hostb) TFX_TRY(copy_any(b, db, (size_t)m.ncols * sizeof(double), ctx->stream));
    return 0;
}

// ---- timers ------------------------------------------------------------------------------------------------
int tfx_timer_start(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    TFX_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

int tfx_timer_stop_ms(tfx_ctx *ctx, double *ms_out)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !ms_out) return fail(TFX_E_ARG, "null argument");
    TFX_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    TFX_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    TFX_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_out = ms;
    return 0;
}

int tfx_profile_enable(tfx_ctx *ctx, int on)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    tfx::prof_drain(ctx);
    ctx->profile = on != 0;
    ctx->prof_ms[0] = ctx->prof_ms[1] = ctx->prof_ms[2] = 0;
    ctx->prof_n[0] = ctx->prof_n[1] = ctx->prof_n[2] = 0;
    return 0;
}

int tfx_profile_get(tfx_ctx *ctx, int which, double *total_ms, int64_t *launches)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || which < 0 || which > 2) return fail(TFX_E_ARG, "bad argument");
    tfx::prof_drain(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[which];
    if (launches) *launches = ctx->prof_n[which];
    return 0;
}

}  // extern "C"
