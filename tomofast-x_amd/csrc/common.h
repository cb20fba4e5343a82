// common.h - context, error handling and device-buffer helpers of libtfx.so (gfx950 only).
#pragma once
// The device code of this library is written for CDNA (gfx9 instruction encoding: `s_waitcnt lgkmcnt` / `s_barrier` in lds_barrier, DPP
// row_shl reductions, mbcnt on 64-bit masks, 64-wide wavefronts).  ARCH in the Makefile can be overridden: anything that is not gfx9
// stops here instead of assembling instructions that mean something else there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "libtfx.so: device code is written for CDNA / gfx9 (MI355X: --offload-arch=gfx950)"
#endif
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <future>
#include <cstdarg>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/tfx.h"

struct tfx_ctx;
namespace tfx {

extern thread_local std::string g_last_error;

inline int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define TFX_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return tfx::fail(TFX_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

#define TFX_TRY(expr)                 \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != 0) return rc_;     \
    } while (0)

// An allocation that fails for lack of memory first gives up the transposed copies the matrices of the calling thread's context got
// in automatic mode (TiledMatrix::T: an optimisation, the adjoint then runs on the tiles of S) and tries once more - e.g. the second
// kernel of a joint inversion that does not fit beside the first one's copy.  g_alloc_ctx is the context of the extern "C" entry point
// the thread is inside (AllocScope, first statement of every entry point that takes a context; null outside), so an eviction can only
// touch the caller's own context and never a destroyed one.  NoEvict switches the retry off for allocations that are themselves
// optional and owned by an object an eviction would free (the storage matrix_begin sets aside for a copy).
extern thread_local tfx_ctx *g_alloc_ctx;
extern thread_local int g_no_evict;
bool evict_adjoint_copies(tfx_ctx *ctx);      // matrix.hip; true when something was freed
struct AllocScope {
    tfx_ctx *prev;
    explicit AllocScope(tfx_ctx *c) : prev(g_alloc_ctx) { g_alloc_ctx = c; }
    ~AllocScope() { g_alloc_ctx = prev; }
    AllocScope(const AllocScope &) = delete;
    AllocScope &operator=(const AllocScope &) = delete;
};
struct NoEvict {
    NoEvict() { ++g_no_evict; }
    ~NoEvict() { --g_no_evict; }
    NoEvict(const NoEvict &) = delete;
    NoEvict &operator=(const NoEvict &) = delete;
};

// Owning device allocation.
template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    ~DBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count)
    {
        release();
        if (count == 0) return 0;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipErrorOutOfMemory && g_alloc_ctx && g_no_evict == 0 && evict_adjoint_copies(g_alloc_ctx)) {
            (void)hipGetLastError();
            e = hipMalloc((void **)&p, count * sizeof(T));
        }
        if (e != hipSuccess)
            return fail(TFX_E_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        n = count;
        static const bool log_big = getenv("TFX_ALLOC_LOG") != nullptr;       // (memory planning: every allocation of 256 MB and more)
        if (log_big && count * sizeof(T) >= ((size_t)1 << 28)) fprintf(stderr, "[tfx alloc] %.2f GB\n", (double)(count * sizeof(T)) / 1e9);
        return 0;
    }
    int ensure(size_t count) { return count <= n ? 0 : alloc(count); }
    size_t bytes() const { return n * sizeof(T); }
};

// ---- tiled sensitivity matrix (DESIGN.md "Data layout in HBM") ------------------------------------------
constexpr int CHUNK = 512;            // entries per chunk = 64 lanes x 8 entries
constexpr int SLOT_WORDS = 192;       // dwords of packed 12-bit column slots per chunk (lane L: words 3L..3L+2 = its 8 slots)
constexpr int MASK_WORDS = 8;         // uint64 row-start masks per chunk: word k, bit L = "entry k of lane L starts a new row"
// The three per-entry streams of a chunk are stored as ONE contiguous record [512 values | 192 slot words | 8 mask words] = 2880 bytes:
// a product reads a single sequential stream (tools/read_bw_probe.hip on this part: three separate arrays 6.71-6.78 TB/s, the same
// loads from interleaved records 6.89-6.93 TB/s).
constexpr int REC_SLOT_OFF = CHUNK * 4;                       // byte offset of the slot words inside a record
constexpr int REC_MASK_OFF = REC_SLOT_OFF + SLOT_WORDS * 4;   // byte offset of the row-start masks
constexpr int REC_BYTES = REC_MASK_OFF + MASK_WORDS * 8;      // 2880
static_assert(REC_BYTES % 64 == 0, "records keep the 16-byte value loads and the 64-byte mask block aligned");
__host__ __device__ inline float *chunk_vals(char *rec, int64_t chunk) { return reinterpret_cast<float *>(rec + chunk * REC_BYTES); }
__host__ __device__ inline uint32_t *chunk_slots(char *rec, int64_t chunk) { return reinterpret_cast<uint32_t *>(rec + chunk * REC_BYTES + REC_SLOT_OFF); }
__host__ __device__ inline unsigned long long *chunk_masks(char *rec, int64_t chunk) { return reinterpret_cast<unsigned long long *>(rec + chunk * REC_BYTES + REC_MASK_OFF); }
__host__ __device__ inline const float *chunk_vals(const char *rec, int64_t chunk) { return reinterpret_cast<const float *>(rec + chunk * REC_BYTES); }
__host__ __device__ inline const uint32_t *chunk_slots(const char *rec, int64_t chunk) { return reinterpret_cast<const uint32_t *>(rec + chunk * REC_BYTES + REC_SLOT_OFF); }
__host__ __device__ inline const uint64_t *chunk_masks(const char *rec, int64_t chunk) { return reinterpret_cast<const uint64_t *>(rec + chunk * REC_BYTES + REC_MASK_OFF); }
constexpr int TC_MAX = 4096;          // columns per column tile (12-bit local column; the x tile is 32 KB of LDS)
constexpr int RB_MAX = 2048;          // rows per (storage) row block
constexpr int FWD_GROUP_MAX = 2;      // the forward product walks up to this many row blocks per staged x tile (row sums: 32 KB of LDS; groups of 4 measured slower in rounds 2-3 and cost scalar registers and compares per chunk)
// The column stored for an entry is the LDS slot of that column inside the tile, not the column itself: slot = col ^ f(col >> 4), a
// bijection inside aligned groups of 16 (an involution) that folds the higher index bits into the four bank bits.  Wavelet
// coefficients sit on index lattices (multiples of 2^l per axis); without the fold the columns of one LDS instruction are
// often congruent modulo 16 and pile onto one bank pair.  Stored pre-swizzled, it costs the product kernels nothing.
__host__ __device__ inline int col_slot(int i) { return i ^ (((i >> 4) ^ (i >> 8)) & 15); }
// Position of entry e (lane L = (e & 511) >> 3 of its chunk, k = e & 7) among the values (val_pos(e) & 511 inside its chunk's record): the chunk's 2 KB hold k = 0..3 of
// all lanes first (lane L at byte 16 L), then k = 4..7 - a wave reads its chunk with two loads of 16 bytes per lane at a lane
// stride of 16 bytes.  (With the lane's 8 values in one 32-byte record both loads touch every cache line of the chunk and the
// stream tops out at 6.3 TB/s; this order reaches 6.7 in tools/read_bw_probe.hip.)
__host__ __device__ inline int64_t val_pos(int64_t e)
{
    return (e & ~(int64_t)511) | (((e >> 2) & 1) << 8) | (((e >> 3) & 63) << 2) | (e & 3);
}

struct TileMeta {
    int64_t off;      // first entry (multiple of CHUNK) in the entry streams
    int32_t nchunks;  // chunks of the tile (padded length / CHUNK)
    int32_t cnt;      // real entries (incl. empty-row markers)
    int32_t t;        // column tile
    int32_t rb;       // row block
    int32_t kind;     // reserved (0)
    int32_t aux;      // reserved (0)
};

struct WorkItem {     // one workgroup's work: a run of tiles sharing rb (forward) or t (adjoint)
    int32_t begin, end;   // range in the order[] array of tile ids
    int32_t key;          // rb (forward) / t (adjoint)
    int32_t slot;         // position inside the run of items with the same key
    int32_t pidx;         // partial-sum tile owned by this item (-1: adjoint slot 0 adds straight into y)
    int32_t cb, ce;       // a heavy tile is shared by several items: this one takes its chunks [cb, ce) (single-tile items; ce < 0: all)
};

struct TiledMatrix {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int TC = TC_MAX, RB = RB_MAX;
    int ntc = 0, nrb = 0;
    int fwd_group = 1;            // row blocks per forward super block (shared x tile)
    DBuf<char> rec;               // REC_BYTES per chunk: values, packed 12-bit slots, row-start masks
    int64_t n_entries = 0;        // used (padded) entries
    int64_t cap_entries = 0;      // allocated entries
    DBuf<int32_t> chunk_row0;     // per chunk: local row of the entry preceding the chunk
    std::vector<TileMeta> h_tiles;
    DBuf<TileMeta> tiles;
    // work lists
    std::vector<WorkItem> h_fwd, h_adj;
    DBuf<WorkItem> fwd, adj;
    DBuf<int32_t> fwd_order, adj_order;
    DBuf<double> fwd_partial;     // fwd_group * RB row sums per forward item
    DBuf<double> adj_partial;     // one TC-tile of column sums per adjoint item with slot >= 1
    DBuf<int32_t> fwd_nslots, fwd_pbase;   // per row block
    DBuf<int32_t> adj_nslots, adj_pbase;   // per column tile
    bool adj_has_partials = false;
    double fwd_avg_nslots = 0.0;  // partial tiles per forward super block on average (picks the reduction kernel)
    DBuf<double> tile_bound;      // largest column sum of |value| per tile: the bound the adjoint kernel on the tiles of S scales by (matrix.hip k_spmv_adj);
    bool vmax_stale = true;       //   computed when that kernel is first used and again after the values changed (scale_rows)
    // dense storage (compression off): fp32 [nrows][ld], no index stream (4 B per entry)
    bool is_dense = false;
    int64_t ld = 0;
    DBuf<float> dense;
    DBuf<double> dense_partial;   // forward: [nchunks][nrows] partial row sums
    bool valid = false;
    // Transposed copy for the adjoint product ("adjoint copy"): S^T in the same tiled layout, so b += S^T x is the
    // FORWARD kernel on it - u rows gathered from LDS, column sums merged in registers and by the segmented wave reduction - instead
    // of one LDS fp64 atomic per non-zero.  Optional (it doubles the matrix memory): ctx->adj_copy.
    TiledMatrix *T = nullptr;
    bool is_transpose_copy = false;
    // Storage of the transposed copy, set aside by matrix_begin together with the storage of S for large matrices whose copy will fit
    // (one early allocation of 100+ GB takes 0.5 s; the same allocation after the build, next to 130 GB in use, was measured at 1.6 s);
    // matrix_build_transpose takes it over.
    struct Prealloc {
        DBuf<char> rec;
        DBuf<int32_t> row0;
        // the two allocations (100+ GB: seconds) run on a helper thread while the build's first row blocks are computed; whoever looks at
        // rec / row0 - or destroys the object - waits for it first
        std::future<void> pending;
        // (get() rethrows what the helper threw; this runs in a destructor, so nothing may leave it: the helper's lambda catches
        // everything itself, and the catch here is the belt to those braces - a failed set-aside just leaves rec / row0 empty)
        void wait() noexcept
        {
            if (!pending.valid()) return;
            try { pending.get(); } catch (...) { rec.release(); row0.release(); }
        }
        ~Prealloc() { wait(); }
    };
    std::unique_ptr<Prealloc> pre;
    void drop_prealloc() { pre.reset(); }
    double copy_build_s = 0.0;    // (of the original) wall clock matrix_build_transpose took: reported beside the kernel build time
    bool evictable = false;       // (of a copy) made in automatic mode: given up when another allocation needs the memory
    ~TiledMatrix() { delete T; }
    TiledMatrix() = default;
    TiledMatrix(const TiledMatrix &) = delete;
    TiledMatrix &operator=(const TiledMatrix &) = delete;
    size_t device_bytes() const;
    void release_storage();       // frees every device buffer and clears the host-side lists
};

// compressed rows kept row-major on the device (all columns): the row-parallel half of the multi-GPU build
struct RowStore {
    DBuf<int32_t> cols;    // [nrows][stride], global 0-based, ascending
    DBuf<float> vals;
    DBuf<int32_t> nel;     // [nrows]
    int64_t nrows = 0, stride = 0;
    int ncm = 1;           // model components: a row holds component k in columns k*N + cell (N cells), ascending
    int64_t N = 0;
};

struct LsqrState;

}  // namespace tfx

struct tfx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cu = 256;
    // grid
    int nx = 0, ny = 0, nz = 0;
    int64_t N = 0;
    tfx::DBuf<double> grid[6];
    tfx::DBuf<double> edges[3];       // xe[nx+1], ye[ny+1], ze[nz+1] when the grid is a tensor product
    bool tensor_grid = false;
    bool force_general_prism = false; // tests: always use the six-array kernel
    int64_t band_min_n = 1 << 20;      // rows of at least this many cells find their threshold by the band select (build.hip)
    int64_t band_batches = 0, band_fallbacks = 0;   // batches that used it / that fell back to the full select
    // matrix S and (optional) general constraint matrix C
    tfx::TiledMatrix mat;
    tfx::TiledMatrix mat2;             // second problem of a joint inversion: S = blockdiag(mat, mat2)  (joint_inverse_problem.F90:712-739)
    int slot = 0;                      // which of the two the matrix-level entry points act on (tfx_select_problem)
    tfx::TiledMatrix &selmat() { return slot ? mat2 : mat; }
    int64_t total_rows() const { return mat.nrows + (mat2.valid ? mat2.nrows : 0); }
    int64_t total_cols() const { return mat.ncols + (mat2.valid ? mat2.ncols : 0); }
    tfx::TiledMatrix cons;
    tfx::DBuf<double> cons_rhs;        // right-hand side of the C rows (replicated)
    tfx::TiledMatrix *target = &mat;   // which matrix matrix_begin / append / finish assemble
    tfx::TiledMatrix::Prealloc *pre_take = nullptr;   // (matrix_build_transpose -> matrix_begin) storage to take over instead of allocating
    tfx::RowStore rowstores[2];        // one per problem slot (a joint run partitions on the counts of both kernels before the relayout)
    tfx::RowStore &rowstore() { return rowstores[slot]; }
    // scratch of matrix_append_rows (grown on demand, reused by every row block)
    struct AppendScratch {
        tfx::DBuf<int32_t> pos, segoff, first_ne, last_ne, tile_nch, tile_total;
        tfx::DBuf<int64_t> tile_off;
        std::vector<int32_t> h_segoff_last, h_nch;
        std::vector<int64_t> h_off;
        tfx::DBuf<uint16_t> tmp16;
    } append;
    // scratch vectors for spmv / spmtv with host pointers
    tfx::DBuf<double> vx, vb, vw;
    // comm
    // comm.hip.  comm_mu guards `comm` and `comm_pending` (check-and-install of tfx_comm_init_rccl against tfx_comm_abort from another
    // thread); comm_pending is the rendezvous that is in flight, so that an abort can cancel exactly that attempt.
    std::mutex comm_mu;
    std::shared_ptr<void> comm_pending;
    double comm_init_timeout_s = 120.0; // tfx_comm_init_rccl gives up after this long (TFX_COMM_INIT_TIMEOUT / debug key "comm_init_timeout_s"; <= 0: waits for ever)
    void *comm = nullptr;              // ncclComm_t (comm.hip): when set, every collective of the path is RCCL on the ctx stream
    tfx_allreduce_fn allreduce = nullptr;
    tfx_allgatherv_fn allgatherv = nullptr;
    void *allreduce_user = nullptr;
    int rank = 0, nranks = 1;
    bool force_collectives = false;    // debug: issue the collectives even on one rank (a world-size-1 communicator exercises RCCL)
    bool multi() const { return nranks > 1 || force_collectives; }
    // lsqr
    tfx::LsqrState *lsqr = nullptr;
    // WAVELET_DOMAIN = F (joint_inverse_problem.F90:189-198): LSQR unknowns are spatial, S acts on Wav(v)
    bool spatial_unknowns = false;
    int wd_n1 = 0, wd_n2 = 0, wd_n3 = 0, wd_type = 0;
    int64_t wd_nvec = 1;      // model components transformed one after the other
    int64_t wd_col_begin = -1; // multi-rank WAVELET_DOMAIN = F: first cell of this rank's column range (tfx_lsqr_set_partition)
    int wd_ncomp = 0;          //   and the number of model components (of all problems) in the local unknown vector
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int gen_after_wavelet = 3;        // debug key: the overlapped generator of the next batch starts behind this many axis passes of the current batch's wavelet transform (0: at the batch start, 3: behind all of them)
    int wave_pipe = 0;                // debug key "wave_pipe" / TFX_WAVE_PIPE: workgroups per CU of the software-pipelined wavelet pass (0: one workgroup per tile)
    int gen_wgs_per_cu = 0;           // debug key "gen_wgs_per_cu": resident generator workgroups per CU in overlap mode (0: one per tile)
    int gen_grid_limit = 0;           // (set around the generator launches of the overlapped build)
    int chain_under_wavelet = 1;      // debug key "chain_under_wavelet": overlapped build - the threshold / compaction chain of batch b runs on a third stream beside the wavelet passes of batch b + 1
    int build_overlap = 1;            // debug key "build_overlap": row generator on its own stream, one batch ahead of the wavelet / compaction
    int items_per_cu = 16;            // debug key "items_per_cu": work items per CU the tile list is cut into
    // Adjoint on a transposed copy of the tiles: 0 never, 1 always (a copy that does not fit is an error), 2 (default) automatic = for
    // matrices of at least adj_copy_min_nnz stored entries whenever the device has room for the second copy (debug key "adj_copy" /
    // TFX_ADJ_COPY).  With the copy b += S^T x is the forward kernel on S^T: fp64 sums in a fixed order, no LDS atomic per non-zero
    // (headline size: 18.05 ms per launch against 18.39, for 6.5 s of build and twice the matrix memory).  Without it the adjoint
    // runs on the tiles of S with exact integer accumulation (matrix.hip k_spmv_adj).  Both are reproducible run to run.
    int adj_copy = 2;
    int64_t adj_copy_min_nnz = 0;
    struct TransposeScratch {
        tfx::DBuf<int32_t> cnt, nel, tcols, tids, bmax;
        tfx::DBuf<int64_t> rowoff, totals, bsum;
        tfx::DBuf<float> tvals;
    } trs;
    double tr_panel_entries = 9.0e8, tr_pos_budget = 1.5e8;   // panel size of the transposition (debug keys "tr_panel_entries" / "tr_pos_budget": tests force many small panels of either shape)
    int64_t reserve_nnz = 0;          // tfx_matrix_reserve: entry bound of the next kernel build into slot `reserve_slot` (0 = rows x K)
    int reserve_slot = 0;             // the slot that was selected when the reservation was made: a build into the other slot ignores it
    int lsqr_merge_tail = 1;          // debug key "lsqr_merge_tail" / TFX_LSQR_MERGE_TAIL: the x / w update of an LSQR iteration also does the next iteration's u = -alpha u and constraint forward step (one launch instead of three; same bits)
    int fwd_run = 2;                  // debug key "fwd_run": consecutive chunks a wave of the forward kernel takes at a time (matrix.hip k_spmv_fwd)
    int fwd_group_override = 0;       // debug key "fwd_group": row blocks per forward super block (0 = automatic)
    size_t wave_lds_attr[4] = {0, 0, 0, 0};   // the same for the four wavelet axis kernels (Haar / D4 x forward / inverse)
    bool profile = false;
    double prof_ms[3] = {0, 0, 0};     // 0 forward product, 1 adjoint product, 2 in-stream all-reduce (comm.hip)
    int64_t prof_n[3] = {0, 0, 0};
    hipEvent_t pev0 = nullptr, pev1 = nullptr;
    // per-launch event pairs of the profiling mode: recorded without a host round trip, resolved when the totals are read
    struct ProfPair { hipEvent_t a, b; int which; };
    std::vector<ProfPair> prof_pending, prof_free;
};

namespace tfx {
// matrix.hip
int matrix_append_rows(tfx_ctx *ctx, int64_t row_begin, int64_t nr, const int32_t *d_cols, const float *d_vals,
                       const int32_t *d_nel, const int64_t *d_rowoff, int64_t maxlen);
int matrix_begin(tfx_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz_upper);
int matrix_finish(tfx_ctx *ctx);
int spmv_dev(tfx_ctx *ctx, const double *d_x, double *d_b, int add);     // b (+)= S x   (device pointers)
int spmtv_dev(tfx_ctx *ctx, const double *d_x, double *d_b, int add);    // b (+)= S^T x
int spmv_dev(tfx_ctx *ctx, TiledMatrix &m, const double *d_x, double *d_b, int add);
int spmtv_dev(tfx_ctx *ctx, TiledMatrix &m, const double *d_x, double *d_b, int add);
int matrix_begin_dense(tfx_ctx *ctx, int64_t nrows, int64_t ncols);
int scale_rows_dev(tfx_ctx *ctx, TiledMatrix &m, const float *d_scale);
int normalize_columns_dev(tfx_ctx *ctx, TiledMatrix &m, double *d_norm);      // matrix.hip; synchronises the ctx stream
int matrix_build_transpose(tfx_ctx *ctx, TiledMatrix &m);    // (no-op unless ctx->adj_copy asks for it and the copy fits)
int chunk_exponent_stats(tfx_ctx *ctx, TiledMatrix &m, int span, int64_t *fit, int64_t *total, unsigned int *hist34);
int copy_any(void *dst, const void *src, size_t bytes, hipStream_t s);
void prof_drain(tfx_ctx *ctx);
void prof_begin(tfx_ctx *ctx);
void prof_end(tfx_ctx *ctx, int which);
// comm.hip
int comm_allreduce_f64(tfx_ctx *ctx, double *buf, int64_t n);
int comm_allgatherv_f64(tfx_ctx *ctx, const double *send, double *recv, const int64_t *counts, const int64_t *displs);   // 1: unavailable
// build.hip
int detect_tensor_grid(tfx_ctx *ctx);
int wavelet_dev(tfx_ctx *ctx, double *d, int n1, int n2, int n3, int64_t nvec, int type, int dir, int axis_from = 0, int axis_to = 3);
}  // namespace tfx
