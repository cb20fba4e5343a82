// comm.hip - RCCL inside libtfx.so: the collectives of the multi-GPU path run on the ctx stream, between the kernels that
// produce and consume their buffers - no host callback, no stream synchronisation, the queued LSQR iterations stay queued.
//
// What the reference does with MPI on the host (one rank per model-column range):
//   lsqr_solver2.F90:214        MPI_Allreduce of u = [S_loc v ; ...]            -> all-reduce #1 (rows + 1 doubles)
//   lsqr_solver2.F90:511-515    MPI_Allreduce of |v_loc|^2                       -> all-reduce #2 (1 double)
//   model.F90:288-293           MPI_Allreduce of the predicted data              -> tfx_calc_data
//   wavelet_utils.F90:37-72     gather to rank 0 / transform / scatter           -> all-gather of the column slices, every GPU transforms
//   sensitivity_gravmag.F90:322 MPI_Allreduce of sensit_nnz                      -> tfx_comm_allreduce (int32)
//   sensitivity_gravmag.F90:795-830  rank-0 MPI_Scatterv of every row            -> point-to-point pieces (tfx_comm_send / _recv)
// One process per GPU; the host language only carries the 128-byte unique id from rank 0 to the others (MPI_Bcast, a file, a
// torch store - anything).  Without a communicator the ctx falls back to the host-supplied hooks (tfx_set_allreduce): that is
// how the multi-rank logic is tested on boxes without several GPUs (gloo / MPI staging), never the production path.
#include "common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <thread>

namespace tfx {

#define TFX_NCCL(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess)                                                                                  \
            return tfx::fail(TFX_E_COMM, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); \
    } while (0)

static inline ncclComm_t comm_of(tfx_ctx *ctx) { return (ncclComm_t)ctx->comm; }

// sum over ranks, fp64, in place on a device buffer, on the ctx stream
int comm_allreduce_f64(tfx_ctx *ctx, double *buf, int64_t n)
{
    if (!ctx->multi() || n <= 0) return 0;
    // tfx_profile_enable: HIP events around the reduction on the ctx stream (slot 2) - what bench.py reports as the all-reduce time
    if (ctx->comm) {
        prof_begin(ctx);
        const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, comm_of(ctx), ctx->stream);
        prof_end(ctx, 2);
        if (r != ncclSuccess) return fail(TFX_E_COMM, "ncclAllReduce: %s", ncclGetErrorString(r));
        return 0;
    }
    if (!ctx->allreduce) return fail(TFX_E_COMM, "several ranks but neither a communicator nor an all-reduce hook");
    prof_begin(ctx);
    int rc = ctx->allreduce(ctx->allreduce_user, buf, n, (void *)ctx->stream);
    prof_end(ctx, 2);
    if (rc != 0) return fail(TFX_E_COMM, "all-reduce hook failed (%d)", rc);
    return 0;
}

// Gathers the slices of all ranks: rank r contributes counts[r] doubles, they land at recv + displs[r] on every rank.
// RCCL: one grouped broadcast per rank (the slices are nnz-balanced column ranges, so their lengths differ: ncclAllGather
// wants equal counts).  Hook fallback: tfx_allgatherv_fn when the host supplied one.  Returns 1 when neither is available
// (the caller then falls back to the zero-padded all-reduce).
int comm_allgatherv_f64(tfx_ctx *ctx, const double *send, double *recv, const int64_t *counts, const int64_t *displs)
{
    if (ctx->comm) {
        TFX_NCCL(ncclGroupStart());
        for (int r = 0; r < ctx->nranks; ++r) {
            if (counts[r] == 0) continue;
            ncclResult_t rr = ncclBroadcast(r == ctx->rank ? (const void *)send : (const void *)(recv + displs[r]), recv + displs[r],
                                            (size_t)counts[r], ncclDouble, r, comm_of(ctx), ctx->stream);
            if (rr != ncclSuccess) {
                (void)ncclGroupEnd();
                return fail(TFX_E_COMM, "ncclBroadcast: %s", ncclGetErrorString(rr));
            }
        }
        TFX_NCCL(ncclGroupEnd());
        return 0;
    }
    if (ctx->allgatherv) {
        int rc = ctx->allgatherv(ctx->allreduce_user, send, counts[ctx->rank], recv, counts, displs, (void *)ctx->stream);
        if (rc != 0) return fail(TFX_E_COMM, "all-gather hook failed (%d)", rc);
        return 0;
    }
    return 1;
}

}  // namespace tfx

using namespace tfx;

extern "C" {

int tfx_comm_unique_id(char *id_out)
{
    if (!id_out) return fail(TFX_E_ARG, "tfx_comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == TFX_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    TFX_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

// One rendezvous attempt.  ncclCommInitRank blocks until every rank has arrived; a peer that died before the rendezvous would hold
// the caller for ever, so the call runs on a helper thread and the caller waits for it with a timeout.  The state is shared between
// the two and owned by neither (the helper may outlive the ctx): an abandoned or cancelled attempt that still completes aborts its
// own communicator.
struct InitAttempt {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;        // the helper has a result
    bool orphaned = false;    // the caller stopped waiting (timeout) or the attempt was cancelled (tfx_comm_abort): nobody will take the result
    ncclResult_t res = ncclSuccess;
    ncclComm_t comm = nullptr;
};

int tfx_comm_init_rccl(tfx_ctx *ctx, const char *unique_id, int rank, int nranks)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !unique_id) return fail(TFX_E_ARG, "tfx_comm_init_rccl: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(TFX_E_ARG, "bad rank %d / %d", rank, nranks);
    auto at = std::make_shared<InitAttempt>();
    {
        std::lock_guard<std::mutex> g(ctx->comm_mu);
        if (ctx->comm) return fail(TFX_E_STATE, "tfx_comm_init_rccl: the context already has a communicator");
        if (ctx->comm_pending) return fail(TFX_E_STATE, "tfx_comm_init_rccl: another rendezvous of this context is still in flight");
        ctx->comm_pending = at;                 // from here on tfx_comm_abort cancels THIS attempt
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    const int device = ctx->device;
    std::thread([at, id, rank, nranks, device]() {
        ncclComm_t c = nullptr;
        ncclResult_t r = hipSetDevice(device) == hipSuccess ? ncclCommInitRank(&c, nranks, id, rank) : ncclUnhandledCudaError;
        std::unique_lock<std::mutex> lk(at->mu);
        if (at->orphaned) {
            lk.unlock();
            if (r == ncclSuccess && c) (void)ncclCommAbort(c);
            return;
        }
        at->res = r;
        at->comm = c;
        at->done = true;
        at->cv.notify_all();
    }).detach();
    bool timed_out = false, cancelled = false;
    {
        std::unique_lock<std::mutex> lk(at->mu);
        auto ready = [&] { return at->done || at->orphaned; };
        if (ctx->comm_init_timeout_s > 0) {
            if (!at->cv.wait_for(lk, std::chrono::duration<double>(ctx->comm_init_timeout_s), ready)) {
                at->orphaned = true;
                timed_out = true;
            }
        } else {
            at->cv.wait(lk, ready);
        }
        cancelled = at->orphaned && !timed_out;
    }
    std::lock_guard<std::mutex> g(ctx->comm_mu);
    if (ctx->comm_pending == std::static_pointer_cast<void>(at)) ctx->comm_pending.reset();
    if (timed_out)
        return fail(TFX_E_COMM, "tfx_comm_init_rccl: the rendezvous did not complete within %.0f s (a peer that never arrived?)", ctx->comm_init_timeout_s);
    {
        // (a cancel that arrived between the helper's result and this point: the flag is re-read under both locks)
        std::lock_guard<std::mutex> la(at->mu);
        if (at->orphaned) cancelled = true;
        if (cancelled) {
            if (at->done && at->res == ncclSuccess && at->comm) (void)ncclCommAbort(at->comm);
            return fail(TFX_E_COMM, "tfx_comm_init_rccl: cancelled by tfx_comm_abort while the rendezvous was in flight");
        }
    }
    if (at->res != ncclSuccess) return fail(TFX_E_COMM, "ncclCommInitRank: %s", ncclGetErrorString(at->res));
    ctx->comm = (void *)at->comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return 0;
}

int tfx_comm_destroy(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    std::lock_guard<std::mutex> g(ctx->comm_mu);
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ncclResult_t r = ncclCommDestroy(comm_of(ctx));
        ctx->comm = nullptr;
        if (!ctx->allreduce) { ctx->rank = 0; ctx->nranks = 1; }
        if (r != ncclSuccess) return fail(TFX_E_COMM, "ncclCommDestroy: %s", ncclGetErrorString(r));
    }
    return 0;
}

// Aborts the communicator without the orderly hand-shake of ncclCommDestroy: what a rank does when the other ranks reported a failed
// start-up (the fallback ladder of the hosts: every rank agrees on the outcome of tfx_comm_init_rccl over its control channel, and
// the ranks whose own call succeeded drop their half-open communicator with this).
int tfx_comm_abort(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    std::lock_guard<std::mutex> g(ctx->comm_mu);
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        ncclResult_t r = ncclCommAbort(comm_of(ctx));
        ctx->comm = nullptr;
        if (!ctx->allreduce) { ctx->rank = 0; ctx->nranks = 1; }
        if (r != ncclSuccess) return fail(TFX_E_COMM, "ncclCommAbort: %s", ncclGetErrorString(r));
    } else if (ctx->comm_pending) {
        // a tfx_comm_init_rccl in flight on another thread: that attempt must not install its communicator (whoever finishes last -
        // the waiting caller or the helper thread - aborts it)
        InitAttempt *at = static_cast<InitAttempt *>(ctx->comm_pending.get());
        std::lock_guard<std::mutex> la(at->mu);
        at->orphaned = true;
        at->cv.notify_all();
    }
    return 0;
}

// Facts about the communicator and the RCCL build that serves it: ranks the communicator itself counts (ncclCommCount - the
// "ranks seen by RCCL" of the bench line), this rank's index and device in it, the library version (ncclGetVersion) and the file
// the nccl* symbols were resolved from (libtfx.so names librccl.so.1 as a dependency; a process that has already mapped an RCCL of
// that soname - PyTorch bundles one - keeps using that one copy, otherwise the loader takes /opt/rocm/lib's).  Without a
// communicator the counts are 0 and only version / path are filled.
int tfx_comm_info(tfx_ctx *ctx, int *nranks_seen, int *rank_seen, int *device_seen, int *version, char *path, int path_len)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    int n = 0, r = -1, d = -1, v = 0;
    if (ctx->comm) {
        TFX_NCCL(ncclCommCount(comm_of(ctx), &n));
        TFX_NCCL(ncclCommUserRank(comm_of(ctx), &r));
        TFX_NCCL(ncclCommCuDevice(comm_of(ctx), &d));
    }
    (void)ncclGetVersion(&v);
    if (nranks_seen) *nranks_seen = n;
    if (rank_seen) *rank_seen = r;
    if (device_seen) *device_seen = d;
    if (version) *version = v;
    if (path && path_len > 0) {
        path[0] = 0;
        Dl_info di;
        if (dladdr((void *)&ncclGetVersion, &di) && di.dli_fname) snprintf(path, (size_t)path_len, "%s", di.dli_fname);
    }
    return 0;
}

// ---- collectives on device buffers for the host's own exchange steps (build histogram, relayout), on the ctx stream
static int need_comm(tfx_ctx *ctx, const char *who)
{
    if (!ctx) return fail(TFX_E_ARG, "%s: null ctx", who);
    if (!ctx->comm) return fail(TFX_E_STATE, "%s: no communicator (tfx_comm_init_rccl)", who);
    return 0;
}

int tfx_comm_allreduce(tfx_ctx *ctx, void *dev_buf, int64_t n, int dtype)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(need_comm(ctx, "tfx_comm_allreduce"));
    if (n < 0 || (n > 0 && !dev_buf)) return fail(TFX_E_ARG, "tfx_comm_allreduce: bad buffer");
    ncclDataType_t t;
    switch (dtype) {
    case TFX_F64: t = ncclDouble; break;
    case TFX_I32: t = ncclInt32; break;
    case TFX_I64: t = ncclInt64; break;
    default: return fail(TFX_E_ARG, "tfx_comm_allreduce: unknown dtype %d", dtype);
    }
    if (n == 0) return 0;
    TFX_HIP(hipSetDevice(ctx->device));
    TFX_NCCL(ncclAllReduce(dev_buf, dev_buf, (size_t)n, t, ncclSum, comm_of(ctx), ctx->stream));
    return 0;
}

int tfx_comm_allgatherv(tfx_ctx *ctx, const double *dev_send, double *dev_recv, const int64_t *counts, const int64_t *displs)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx || !dev_recv || !counts || !displs) return fail(TFX_E_ARG, "tfx_comm_allgatherv: null argument");
    const int nr = std::max(1, ctx->nranks);
    for (int r = 0; r < nr; ++r)
        if (counts[r] < 0 || displs[r] < 0) return fail(TFX_E_ARG, "tfx_comm_allgatherv: negative count / displacement for rank %d", r);
    if (counts[std::min(ctx->rank, nr - 1)] > 0 && !dev_send) return fail(TFX_E_ARG, "tfx_comm_allgatherv: null send buffer");
    TFX_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm && !ctx->allgatherv) {
        if (nr > 1) return fail(TFX_E_COMM, "tfx_comm_allgatherv: several ranks but neither a communicator nor an all-gather hook");
        if (counts[0] > 0)
            TFX_HIP(hipMemcpyAsync(dev_recv + displs[0], dev_send, (size_t)counts[0] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    const int rc = comm_allgatherv_f64(ctx, dev_send, dev_recv, counts, displs);
    if (rc == 1) return fail(TFX_E_COMM, "tfx_comm_allgatherv: neither a communicator nor an all-gather hook");
    return rc;
}

int tfx_comm_group_begin(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(need_comm(ctx, "tfx_comm_group_begin"));
    TFX_HIP(hipSetDevice(ctx->device));
    TFX_NCCL(ncclGroupStart());
    return 0;
}

int tfx_comm_group_end(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(need_comm(ctx, "tfx_comm_group_end"));
    TFX_NCCL(ncclGroupEnd());
    return 0;
}

int tfx_comm_send(tfx_ctx *ctx, const void *dev_buf, int64_t bytes, int peer)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(need_comm(ctx, "tfx_comm_send"));
    if (bytes < 0 || peer < 0 || peer >= ctx->nranks || peer == ctx->rank) return fail(TFX_E_ARG, "tfx_comm_send: bad arguments");
    if (bytes == 0) return 0;
    TFX_NCCL(ncclSend(dev_buf, (size_t)bytes, ncclUint8, peer, comm_of(ctx), ctx->stream));
    return 0;
}

int tfx_comm_recv(tfx_ctx *ctx, void *dev_buf, int64_t bytes, int peer)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(need_comm(ctx, "tfx_comm_recv"));
    if (bytes < 0 || peer < 0 || peer >= ctx->nranks || peer == ctx->rank) return fail(TFX_E_ARG, "tfx_comm_recv: bad arguments");
    if (bytes == 0) return 0;
    TFX_NCCL(ncclRecv(dev_buf, (size_t)bytes, ncclUint8, peer, comm_of(ctx), ctx->stream));
    return 0;
}

int tfx_comm_barrier(tfx_ctx *ctx)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    TFX_TRY(need_comm(ctx, "tfx_comm_barrier"));
    TFX_HIP(hipSetDevice(ctx->device));
    TFX_TRY(ctx->vb.ensure(1));
    TFX_HIP(hipMemsetAsync(ctx->vb.p, 0, sizeof(double), ctx->stream));
    TFX_NCCL(ncclAllReduce(ctx->vb.p, ctx->vb.p, 1, ncclDouble, ncclSum, comm_of(ctx), ctx->stream));
    TFX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int tfx_set_allgatherv(tfx_ctx *ctx, tfx_allgatherv_fn fn)
{
    tfx::AllocScope alloc_scope_(ctx);      // (an allocation that runs out of memory may give up THIS context's automatic adjoint copies)
    if (!ctx) return fail(TFX_E_ARG, "null ctx");
    ctx->allgatherv = fn;
    return 0;
}

}  // extern "C"
