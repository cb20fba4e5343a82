"""Multi-GPU glue: one process per GPU; the data path's collectives are RCCL inside libtfx.so, torch.distributed (gloo) is the control channel.

The reference's model-cell domain decomposition (MPI, src/forward/gravmag/sensitivity_gravmag.F90:470-524 and
src/inversion/lsqr_solver2.F90:194-241) maps to: rank r owns a contiguous column range of S (nnz-balanced with the
reference's own greedy rule), x / v / w and the damping rows are rank-local, u_data is replicated, and LSQR needs two
small all-reduces per iteration, which libtfx.so requests through its all-reduce hook (tfx_set_allreduce).  This module
holds the partition logic (pure host code, tested on CPU with gloo, world_size 2) and sets the collectives up:
`setup_comm` joins the ranks in an RCCL communicator INSIDE libtfx.so (torch.distributed only carries the 128-byte id),
after which every collective of the path is queued by the library on its own stream; the torch.distributed hooks remain
for boxes where RCCL cannot connect the ranks (gloo tests, several processes on one GPU)."""
import ctypes as C

import numpy as np


def calculate_nelements_at_cpu(n, rank, nranks):
    """src/utils/parallel_tools.f90:46-63: 1-D block partition (the remainder goes to the last rank)."""
    base = n // nranks
    return base + (n % nranks if rank == nranks - 1 else 0)


def row_range(n, rank, nranks):
    base = n // nranks
    return rank * base, rank * base + calculate_nelements_at_cpu(n, rank, nranks)


def column_ranges(nelements_at_cpu):
    ends = np.cumsum(np.asarray(nelements_at_cpu, np.int64))
    return [(int(e - n), int(e)) for e, n in zip(ends, nelements_at_cpu)]


class _DevArray:
    """Exposes a raw device pointer through the CUDA array interface so torch can alias it (zero copy)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class TorchAllreduce:
    """Hook fallback for boxes without RCCL between the ranks (gloo CPU tests, several processes sharing one GPU): the fp64 sum
    all-reduce and the all-gather of a device buffer through torch.distributed, issued on the stream libtfx.so passes in."""

    def __init__(self, device_index):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device = torch.device("cuda", device_index)
        self.zero_copy = self._probe_zero_copy()
        self._hip = None
        self._staging = None
        self._alias = {}          # (ptr, n) -> tensor aliasing the library's buffer (LSQR reduces the same two buffers every iteration)
        self._ext = {}            # raw stream handle -> torch.cuda.ExternalStream

    def _probe_zero_copy(self):
        torch = self.torch
        try:
            t = torch.arange(4, dtype=torch.float64, device=self.device)
            a = torch.as_tensor(_DevArray(t.data_ptr(), 4), device=self.device)
            a[1] = 41.0
            torch.cuda.synchronize(self.device)
            return bool(t[1].item() == 41.0 and a.data_ptr() == t.data_ptr())
        except Exception:
            return False

    def _on(self, stream):
        """Context manager that makes `stream` (the hipStream_t the library launches on) torch's current stream, so that the
        collective and the staging copies are ordered with the kernels queued before and after them (tfx.h: the hook is
        'enqueued on `stream`').  The null stream is torch's default stream: nothing to switch."""
        import contextlib
        if not stream:
            return contextlib.nullcontext()
        ext = self._ext.get(stream)
        if ext is None:
            ext = self._ext[stream] = self.torch.cuda.ExternalStream(stream, device=self.device)
        return self.torch.cuda.stream(ext)

    def _copy(self, dst, src, nbytes, stream):
        if self._hip is None:
            self._hip = C.CDLL("libamdhip64.so")
        rc = self._hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), C.c_int(3), C.c_void_p(stream))   # 3 = D2D
        if rc != 0:
            raise RuntimeError("hipMemcpyAsync failed: %d" % rc)

    def _tensor(self, ptr, n):
        t = self._alias.get((ptr, n))
        if t is None:
            if len(self._alias) > 64:
                self._alias.clear()
            t = self._alias[(ptr, n)] = self.torch.as_tensor(_DevArray(ptr, n), device=self.device)
        return t

    def __call__(self, ptr, n, stream):
        torch, dist = self.torch, self.dist
        with self._on(stream):
            if self.zero_copy:
                dist.all_reduce(self._tensor(ptr, n), op=dist.ReduceOp.SUM)
                return
            if self._staging is None or self._staging.numel() < n:
                self._staging = torch.empty(max(n, 1 << 16), dtype=torch.float64, device=self.device)
            st = self._staging[:n]
            self._copy(st.data_ptr(), ptr, 8 * n, stream)
            dist.all_reduce(st, op=dist.ReduceOp.SUM)
            self._copy(ptr, st.data_ptr(), 8 * n, stream)

    def allgatherv(self, send_ptr, nsend, recv_ptr, counts, displs, stream):
        """MPI_Allgatherv on device buffers: rank r's counts[r] doubles land at recv + displs[r] on every rank."""
        torch, dist = self.torch, self.dist
        with self._on(stream):
            mine = torch.empty(max(nsend, 1), dtype=torch.float64, device=self.device)
            if nsend:
                self._copy(mine.data_ptr(), send_ptr, 8 * nsend, stream)
            pieces = [torch.empty(max(c, 1), dtype=torch.float64, device=self.device) for c in counts]
            # ragged all-gather as one broadcast per rank (gloo and nccl both take it)
            for r, c in enumerate(counts):
                buf = mine if r == dist.get_rank() else pieces[r]
                dist.broadcast(buf[:max(c, 1)], src=r)
                if c:
                    self._copy(recv_ptr + 8 * displs[r], buf.data_ptr(), 8 * c, stream)
            torch.cuda.current_stream(self.device).synchronize()       # the temporaries die with this frame


class HostComm:
    """The exchange steps the HOST drives around the library (nnz histogram, counts, matrix pieces, barrier, timing):
    RCCL inside libtfx.so when the ctx has a communicator (`rccl=True`), torch.distributed otherwise."""

    def __init__(self, ctx, rank, nranks, device_index, rccl):
        import torch
        self.ctx, self.rank, self.nranks, self.rccl = ctx, rank, nranks, rccl
        self.torch = torch
        self.dev = torch.device("cuda", device_index)

    def allreduce_host(self, arr):
        """Sum over the ranks of a numpy array (int32 / int64 / float64); returns the reduced array."""
        if self.nranks == 1:
            return arr
        torch = self.torch
        a = np.ascontiguousarray(arr)
        if not self.rccl:
            return allreduce_numpy(a)
        kind = {np.dtype(np.float64): "f64", np.dtype(np.int32): "i32", np.dtype(np.int64): "i64"}[a.dtype]
        t = torch.from_numpy(a).to(self.dev)
        torch.cuda.synchronize(self.dev)
        self.ctx.comm_allreduce(t.data_ptr(), a.size, kind)
        torch.cuda.synchronize(self.dev)
        return t.cpu().numpy()

    def allgather_host(self, arr):
        """Equal-shape numpy arrays of all ranks, as a list."""
        if self.nranks == 1:
            return [arr]
        a = np.ascontiguousarray(arr)
        if self.rccl:                                   # gather as a sum of disjoint supports (small host-side tables)
            full = np.zeros((self.nranks,) + a.shape, a.dtype)
            full[self.rank] = a
            full = self.allreduce_host(full.reshape(-1)).reshape(full.shape)
            return [full[r] for r in range(self.nranks)]
        import torch.distributed as dist
        torch = self.torch
        out = [torch.zeros(a.shape, dtype=torch.from_numpy(a).dtype) for _ in range(self.nranks)]
        dist.all_gather(out, torch.from_numpy(a), group=control_group())
        return [o.numpy() for o in out]

    def exchange(self, sends, recvs):
        """sends: [(dst_rank, device tensor)], recvs: [(src_rank, device tensor)] - matched point-to-point transfers."""
        torch = self.torch
        if self.rccl:
            torch.cuda.synchronize(self.dev)
            self.ctx.comm_group_begin()
            for dst, t in sends:
                self.ctx.comm_send(t.data_ptr(), t.numel() * t.element_size(), dst)
            for src, t in recvs:
                self.ctx.comm_recv(t.data_ptr(), t.numel() * t.element_size(), src)
            self.ctx.comm_group_end()
            torch.cuda.synchronize(self.dev)
            return
        import torch.distributed as dist
        torch.cuda.synchronize(self.dev)
        _p2p(sends, recvs, dist.get_backend())
        torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.nranks == 1:
            self.torch.cuda.synchronize(self.dev)
            return
        if self.rccl:
            self.ctx.comm_barrier()
        else:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, value):
        v = np.zeros(self.nranks)
        v[self.rank] = value
        return float(self.allreduce_host(v).max())


_CONTROL = {"group": None}


def control_group():
    """The control channel of the ranks: a gloo group (CPU tensors over TCP on 127.0.0.1) that exists independently of every GPU
    communicator.  It carries the 128-byte RCCL id, the agreement flags of the start-up ladder and the host-side tables; the data
    path never touches it when the library's own communicator is up.  The default group itself when that is gloo."""
    import torch.distributed as dist
    if _CONTROL["group"] is None:
        _CONTROL["group"] = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _CONTROL["group"]


def agree(ok):
    """True only when EVERY rank reports ok (min over the control channel): the ranks take the same branch of a fallback."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group())
    return bool(int(t[0]) == 1)


def _call_with_timeout(fn, seconds):
    """Runs fn() in a helper thread (the ctypes calls release the GIL); returns (finished, exception or None).  A collective that
    never returns - a peer that died inside the rendezvous - then costs `seconds`, not the whole run."""
    import threading
    box = {}

    def run():
        try:
            fn()
        except BaseException as e:      # noqa
            box["exc"] = e
        box["done"] = True
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        _stragglers.append(th)          # still inside a collective: fail_all aborts the communicator and then waits for it
    return bool(box.get("done")), box.get("exc")


_stragglers = []


def setup_comm(ctx, rank, nranks, device_index=0, want_rccl=None, log=None, init_timeout=None, force=False):
    """Gives the ctx its collectives and returns the HostComm for the host-driven exchange steps.

    The start-up is a ladder that every rank climbs in lock-step - after each rung the ranks AGREE on the outcome over the gloo
    control channel, so a failure on one rank moves all of them to the next rung instead of stranding the others in a collective:
      1. RCCL inside libtfx.so (the production path): pre-flight (one GPU per rank, a unique id) -> agree -> tfx_comm_init_rccl
         under a timeout -> agree -> a one-element all-reduce and a ring send / recv through the new communicator -> agree;
         the communicator must count `nranks` members itself (ncclCommCount).  From then on LSQR's reductions, calc_data, the
         slices of WAVELET_DOMAIN = F and the relayout are RCCL calls on the ctx stream; torch.distributed only carried 128 bytes.
      2. hooks over torch.distributed (gloo stages through the host): slower, but it completes - ranks sharing one GPU in the
         tests, or a node on which RCCL can not connect the ranks.
    `comm.report` says which rung ran and why (bench.py prints it).  TFX_COMM=hooks skips rung 1, TFX_COMM=rccl insists on it.
    force=True takes a single rank through the same start-up (a world-size-1 communicator with the collectives forced on): what a
    one-GPU box can prove about the production path."""
    import os
    import torch
    import torch.distributed as dist
    if log is None:
        log = lambda msg: None      # noqa
    if nranks == 1 and not force:
        comm = HostComm(ctx, 0, 1, device_index, False)
        comm.report = {"path": "single rank", "ladder": []}
        return comm
    mode = os.environ.get("TFX_COMM", "")
    if want_rccl is None:
        want_rccl = mode != "hooks" and (mode == "rccl" or dist.get_backend() in ("nccl", "gloo") and torch.cuda.device_count() >= 1
                                         and os.environ.get("TFX_BENCH_SHARE_GPU") != "1")
    if init_timeout is None:
        init_timeout = float(os.environ.get("TFX_COMM_INIT_TIMEOUT", "240"))
    report = {"path": None, "ladder": []}
    if want_rccl:
        why = _try_rccl(ctx, rank, nranks, device_index, init_timeout, report)
        if why is None:
            info = ctx.comm_info()
            report.update(path="RCCL inside libtfx.so (tfx_comm_init_rccl)", **info)
            if nranks == 1:
                ctx.debug_set("force_collectives", 1)
            comm = HostComm(ctx, rank, nranks, device_index, True)
            comm.report = report
            return comm
        log("RCCL start-up did not complete on every rank (%s): all ranks fall back to the torch.distributed hooks" % why)
        if mode == "rccl":
            raise RuntimeError("TFX_COMM=rccl but the communicator could not be set up: %s" % why)
    hook = TorchAllreduce(device_index)
    ctx.set_allreduce(hook, rank, nranks)
    ctx.set_allgatherv(hook.allgatherv)
    report["path"] = "torch.distributed hooks (%s)" % dist.get_backend()
    try:
        report.update({k: v for k, v in ctx.comm_info().items() if k in ("rccl_version", "librccl")}, rccl_ranks=0)
    except Exception:      # noqa
        pass
    comm = HostComm(ctx, rank, nranks, device_index, False)
    comm.report = report
    return comm


def _try_rccl(ctx, rank, nranks, device_index, timeout, report):
    """Rung 1 of setup_comm.  Returns None when every rank holds a working communicator, else the reason (the same decision on all
    ranks); on failure no rank is left with a communicator."""
    import os
    import torch
    import torch.distributed as dist
    grp = control_group()
    step = report["ladder"]

    def fail_all(stage, local_reason):
        # collect the first reason any rank has (for the log), then make sure nobody keeps a half-open communicator
        reasons = [None] * nranks
        dist.all_gather_object(reasons, local_reason, group=grp)
        try:
            ctx.comm_abort()
        except Exception:      # noqa
            pass
        # a helper thread that timed out inside a collective returns once its communicator is aborted: wait for it, so that nothing
        # still drives the ctx when the hooks are installed (a thread that does not come back leaves the ctx unusable for RCCL)
        while _stragglers:
            th = _stragglers.pop()
            th.join(5.0)
            if th.is_alive():
                ctx._rccl_poisoned = True
        why = "%s: %s" % (stage, next((("rank %d: %s" % (r, x)) for r, x in enumerate(reasons) if x), "unknown"))
        step.append({"stage": stage, "ok": False, "why": why})
        return why

    # -- pre-flight: every rank on a GPU of its own (bus ids over the control channel), rank 0 can draw an id
    mine, err, uid = None, None, b"\0" * 128
    try:
        import socket
        props = torch.cuda.get_device_properties(device_index)
        bus = tuple(getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        # two ranks share a GPU when they sit on the same host AND drive the same device index (the PCI address rides along for the
        # log; it is not trusted on its own - a virtualised box may report the same address for every device)
        vis = "|".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        mine = (socket.gethostname(), int(device_index), vis, "pci %s:%s:%s" % bus if None not in bus else "uuid %s" % (getattr(props, "uuid", None),))
        if rank == 0:
            uid = ctx.comm_unique_id()
    except Exception as e:      # noqa
        err = repr(e)
    ids = [None] * nranks
    dist.all_gather_object(ids, (mine, err), group=grp)
    # (every field must coincide: when in doubt the rendezvous is tried - RCCL itself refuses two ranks on one GPU, and the ladder
    # then moves all ranks to the hooks)
    dup = len({i[0] for i in ids}) != nranks
    local = err or ("ranks share a GPU (%s)" % ", ".join("%s gpu %s [%s] %s" % i[0] if i[0] else "?" for i in ids) if dup else None)
    if not agree(local is None):
        return fail_all("pre-flight", local)
    step.append({"stage": "pre-flight", "ok": True})
    # -- the id travels as 128 bytes over gloo; the rendezvous itself under a timeout
    box = [uid]
    dist.broadcast_object_list(box, src=0, group=grp)
    uid = box[0]
    # the rendezvous gives up by itself after `timeout` (a helper thread inside libtfx.so, tfx_comm_init_rccl); the wrapper here is
    # the belt to those braces
    ctx.debug_set("comm_init_timeout_s", max(1, int(timeout)))
    done, exc = _call_with_timeout(lambda: ctx.comm_init_rccl(uid, rank, nranks), timeout + min(15.0, 0.5 * timeout + 2.0))
    local = None if (done and exc is None) else ("tfx_comm_init_rccl %s" % ("timed out after %.0f s" % timeout if not done else repr(exc)))
    if not agree(local is None):
        return fail_all("tfx_comm_init_rccl", local)
    step.append({"stage": "tfx_comm_init_rccl", "ok": True})
    # -- first contact: the communicator counts its members, a sum travels through it, every rank reaches both ring neighbours
    def probe():
        dev = torch.device("cuda", device_index)
        info = ctx.comm_info()
        if info["rccl_ranks"] != nranks or info["rccl_rank"] != rank:
            raise RuntimeError("the communicator reports rank %d of %d, expected %d of %d" % (info["rccl_rank"], info["rccl_ranks"], rank, nranks))
        one = torch.full((3,), float(rank + 1), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        ctx.comm_allreduce(one.data_ptr(), 3, "f64")
        ctx.comm_barrier()
        if abs(float(one[0].item()) - nranks * (nranks + 1) / 2.0) > 0:
            raise RuntimeError("all-reduce through the communicator gave %r" % (one.tolist(),))
        if nranks == 1:
            return
        sendb = torch.full((1024,), float(rank), dtype=torch.float32, device=dev)
        recvb = torch.full((1024,), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        ctx.comm_group_begin()
        ctx.comm_send(sendb.data_ptr(), 4096, (rank + 1) % nranks)
        ctx.comm_recv(recvb.data_ptr(), 4096, (rank - 1) % nranks)
        ctx.comm_group_end()
        ctx.comm_barrier()
        if not bool((recvb == float((rank - 1) % nranks)).all().item()):
            raise RuntimeError("ring send / recv delivered the wrong data")
    done, exc = _call_with_timeout(probe, timeout)
    local = None if (done and exc is None) else ("first collectives %s" % ("timed out after %.0f s" % timeout if not done else repr(exc)))
    if not agree(local is None):
        return fail_all("first collectives", local)
    step.append({"stage": "first collectives (all-reduce, ring send / recv, ncclCommCount)", "ok": True})
    return None


class AgreedSteps:
    """Steps that every rank runs in lock-step, each under a timeout and with the outcome AGREED over the gloo control channel (min over
    the ranks): a step that raises or hangs on ONE rank is a failed step on all of them, its record names the rank and the reason, and
    the steps after it are recorded as not run - no rank goes on into a collective the others have given up on.  `multi` = a process group
    exists; fn(info) fills `info` with what it measured."""

    def __init__(self, multi, log=None):
        self.multi, self.ok, self.steps = bool(multi), True, []
        self.log = log or (lambda msg: None)

    def run(self, name, fn, timeout, applicable=True, why_not=""):
        import time
        if not self.ok:
            self.steps.append({"step": name, "ok": None, "why": "not run: an earlier step failed"})
            return None
        if not applicable:
            self.steps.append({"step": name, "ok": None, "why": why_not or "not applicable"})
            return None
        info = {}
        t0 = time.time()
        done, exc = _call_with_timeout(lambda: fn(info), timeout)
        local = None if (done and exc is None) else ("timed out after %.0f s" % timeout if not done else repr(exc))
        ok = agree(local is None) if self.multi else (local is None)
        rec = {"step": name, "ok": bool(ok), "s": round(time.time() - t0, 3)}
        rec.update(info)
        if not ok:
            reasons = [local]
            if self.multi:
                import torch.distributed as dist
                reasons = [None] * dist.get_world_size()
                dist.all_gather_object(reasons, local, group=control_group())
            rec["why"] = next((("rank %d: %s" % (r, x)) for r, x in enumerate(reasons) if x), "unknown")
            self.ok = False
        self.steps.append(rec)
        self.log("self-test %s: %s" % (name, "ok (%.2f s)" % rec["s"] if ok else "FAILED - " + rec.get("why", "")))
        return ok


def comm_selftest(ctx, comm, rank, nranks, device_index=0, step_timeout=90.0, log=None, lsqr_iterations=5):
    """First contact with N GPUs, step by step (bench.py --selftest, and the first thing a `--gpus N` run does): every collective SHAPE
    the path uses is executed once on a small, known-answer input - under a per-step timeout, with the ranks agreeing on each outcome
    over the gloo control channel - so that a call that has never run with N > 1 real peers fails EARLY and says which one it was
    instead of hanging the timed region.  Steps (reference call sites in brackets):
      1. ladder                 which rung of setup_comm ran (its own report)
      2. allreduce_rows_plus_1  three in-stream ncclAllReduce of 99 857 doubles back to back, known integer answer; then the same
                                reduction BETWEEN two kernels of the library on the ctx stream (forward wavelet -> all-reduce ->
                                inverse wavelet on 47 x 47 x 45 doubles)                       [lsqr_solver2.F90:214, model.F90:288-293]
      3. allgatherv_unequal     tfx_comm_allgatherv: one group of ncclBroadcast calls, unequal counts, one of them zero
                                                                                                  [wavelet_utils.F90:37-72]
      4. relayout_group         ONE ncclGroup of P - 1 sends + P - 1 receives of different sizes (multiples of a 2048-row block)
                                                                                                  [sensitivity_gravmag.F90:795-830]
      5. lsqr                   `small` workload: partitioned build (row-parallel + relayout on RCCL), predicted data, 5 LSQR
                                iterations; x gathered over gloo and compared with a single-rank solve of the same problem on rank 0
                                                                                                  [lsqr_solver2.F90:163-290]
    Steps 2-4 need the library's own communicator; on the hooks rung they are recorded as not applicable.  Returns
    {"ok": bool, "steps": [...]}; after a failed or timed-out step the remaining ones are not run (the communicator may be wedged)."""
    import time
    import torch
    import torch.distributed as dist
    from . import synthetic
    from .sensitivity import Context
    if log is None:
        log = lambda msg: None      # noqa
    dev = torch.device("cuda", device_index)
    multi = dist.is_available() and dist.is_initialized()
    runner = AgreedSteps(multi, log)
    runner.steps.append({"step": "ladder", "ok": True, "path": comm.report.get("path"), "rungs": [r.get("stage") for r in comm.report.get("ladder", [])]})
    steps, state = runner.steps, runner

    def run_step(name, fn, timeout, applicable=True):
        runner.run(name, fn, timeout, applicable, "not applicable: no RCCL communicator on this rung (%s)" % comm.report.get("path"))

    T = nranks * (nranks + 1) // 2

    def s_allreduce(info):
        n = 99857                                                       # rows + 1 of the headline problem
        pat = (torch.arange(n, dtype=torch.float64, device=dev) % 1024) + 1.0
        buf = pat * float(rank + 1)
        torch.cuda.synchronize(dev)
        for _ in range(3):                                              # queued back to back on the ctx stream, no host sync between
            ctx.comm_allreduce(buf.data_ptr(), n, "f64")
        ctx.comm_barrier()
        if not torch.equal(buf, pat * float(T * nranks * nranks)):
            raise RuntimeError("three chained all-reduces of %d doubles gave a wrong sum (first element %r, expected %r)" %
                               (n, float(buf[0].item()), float(T * nranks * nranks)))
        # between two kernels of the library: W x_r -> sum over ranks -> W^-1 = sum of x_r (Haar lifting is linear)
        n1, n2, n3 = 47, 47, 45
        g = torch.Generator(device="cpu").manual_seed(7)
        base = torch.randn(n1 * n2 * n3, dtype=torch.float64, generator=g)
        mine = (base * float(rank + 1)).to(dev)
        torch.cuda.synchronize(dev)
        ctx.wavelet_device(mine.data_ptr(), n1, n2, n3, 1, 1, 1)
        ctx.comm_allreduce(mine.data_ptr(), mine.numel(), "f64")
        ctx.wavelet_device(mine.data_ptr(), n1, n2, n3, 1, 1, 2)
        ctx.comm_barrier()
        err = float((mine.cpu() - base * float(T)).abs().max() / (base.abs().max() * T))
        info.update(doubles=n, chained=3, between_kernels_rel_err=err)
        if not err <= 1e-13:
            raise RuntimeError("wavelet -> all-reduce -> inverse wavelet: relative error %.2e" % err)

    def s_allgatherv(info):
        counts = np.array([1000 + 37 * r for r in range(nranks)], np.int64)
        if nranks >= 3:
            counts[1] = 0
        displs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        send = torch.full((max(int(counts[rank]), 1),), rank + 0.5, dtype=torch.float64, device=dev)
        recv = torch.full((int(counts.sum()),), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        ctx.comm_allgatherv(send.data_ptr(), recv.data_ptr(), counts, displs)
        ctx.comm_barrier()
        got = recv.cpu().numpy()
        want = np.concatenate([np.full(int(c), r + 0.5) for r, c in enumerate(counts)])
        info.update(counts=[int(c) for c in counts])
        if not np.array_equal(got, want):
            raise RuntimeError("all-gather with unequal counts delivered wrong data")

    def piece(src, dst):
        return 2048 * (1 + (5 * src + 3 * dst) % 7)                     # int32 elements; different for every ordered pair

    def s_relayout(info):
        sends, recvs, keep = [], [], []
        for d in range(nranks):
            if d != rank:
                t = torch.full((piece(rank, d),), rank * 64 + d, dtype=torch.int32, device=dev)
                keep.append(t)
                sends.append((d, t))
        for o in range(nranks):
            if o != rank:
                t = torch.full((piece(o, rank),), -1, dtype=torch.int32, device=dev)
                recvs.append((o, t))
        comm.exchange(sends, recvs)
        for o, t in recvs:
            if not bool((t == o * 64 + rank).all().item()):
                raise RuntimeError("the piece of rank %d arrived damaged" % o)
        info.update(sends=len(sends), receives=len(recvs), bytes_out=int(sum(4 * t.numel() for _, t in sends)))

    def s_lsqr(info):
        nx, ny, nz, ox, oy, ctype, rate = 64, 64, 32, 32, 32, 1, 0.1
        N = nx * ny * nz
        grid = synthetic.grid(nx, ny, nz)
        xs, ys, zs = synthetic.observations(nx, ny, ox, oy)
        ctx.set_grid(nx, ny, nz, *grid)
        cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
        if nranks > 1 and comm.rccl:
            part = build_partitioned_exchange(ctx, rank, nranks, xs, ys, zs, cw, ctype, rate, device_index=device_index, comm=comm)
        else:
            part = build_partitioned(ctx, rank, nranks, xs, ys, zs, cw, ctype, rate, comm=comm)
        c0, c1 = part["col_range"]
        xw = ctx.forward_wavelet(synthetic.true_model(nx, ny, nz) / cw, nx, ny, nz, ctype)
        d_obs = ctx.calc_data(xw[c0:c1], 1.0, None)
        alpha = np.float32(1e-7)
        x, it, r = ctx.lsqr_solve_sensit(d_obs, lsqr_iterations, 1e-300, 0.0, 0.0, [np.full(c1 - c0, alpha, np.float32)], [np.zeros(c1 - c0)])
        ctx.matrix_free()
        pieces = [(c0, c1, x)]
        if multi and nranks > 1:
            pieces = [None] * nranks
            dist.all_gather_object(pieces, (c0, c1, x), group=control_group())
        info.update(iterations=int(it), r=float(r), nnz=int(part["nnz_total"]), columns_per_rank=[int(p[1] - p[0]) for p in pieces])
        if rank == 0:
            full = np.zeros(N)
            for a, b, xv in pieces:
                full[a:b] = xv
            one = Context(device_index)
            try:
                one.set_grid(nx, ny, nz, *grid)
                res = one.calculate_sensit(xs, ys, zs, cw, ctype, rate)
                d1 = one.calc_data(xw, 1.0, None)
                x1, it1, r1 = one.lsqr_solve_sensit(d1, lsqr_iterations, 1e-300, 0.0, 0.0, [np.full(N, alpha, np.float32)], [np.zeros(N)])
            finally:
                one.close()
            rel = float(np.linalg.norm(full - x1) / np.linalg.norm(x1))
            info.update(single_rank_r=float(r1), x_rel_l2_vs_single_rank=rel, x_bits_identical=bool(np.array_equal(full, x1)),
                        data_rel_l2_vs_single_rank=float(np.linalg.norm(d_obs - d1) / np.linalg.norm(d1)), nnz_single_rank=int(res["nnz"]))
            if int(res["nnz"]) != int(part["nnz_total"]) or not rel <= 1e-9 or abs(r - r1) > 1e-9 * abs(r1):
                raise RuntimeError("x after %d iterations is %.2e from the single-rank solve (r %.12e against %.12e, nnz %d against %d)" %
                                   (lsqr_iterations, rel, r, r1, part["nnz_total"], res["nnz"]))

    rccl = bool(comm.rccl)
    run_step("allreduce_rows_plus_1", s_allreduce, step_timeout, applicable=rccl)
    run_step("allgatherv_unequal", s_allgatherv, step_timeout, applicable=rccl)
    run_step("relayout_group", s_relayout, step_timeout, applicable=rccl and nranks > 1)
    run_step("lsqr", s_lsqr, 3.0 * step_timeout)
    return {"ok": bool(state.ok), "steps": steps, "step_timeout_s": step_timeout}


def fall_back_to_hooks(ctx, comm, rank, nranks, device_index, why):
    """After a failed self-test on the RCCL rung: every rank drops the communicator and installs the torch.distributed hooks (the ranks
    have agreed on the failure inside comm_selftest, so all of them get here)."""
    import torch.distributed as dist
    try:
        ctx.comm_abort()
    except Exception:      # noqa
        pass
    while _stragglers:
        th = _stragglers.pop()
        th.join(5.0)
    ctx.debug_set("force_collectives", 0)
    hook = TorchAllreduce(device_index)
    ctx.set_allreduce(hook, rank, nranks)
    ctx.set_allgatherv(hook.allgatherv)
    new = HostComm(ctx, rank, nranks, device_index, False)
    new.report = dict(comm.report)
    new.report["path"] = "torch.distributed hooks (%s) after the RCCL self-test failed: %s" % (dist.get_backend(), why)
    new.report["rccl_ranks"] = 0
    return new


def allreduce_numpy(arr, op="sum"):
    """Host-side all-reduce of a numpy array (histograms, scalars) through torch.distributed (any backend)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=control_group())
    return t.numpy()


def data_row_partition(ndata, nranks):
    """Data dealt out for the row-parallel build by the reference's own rule (calculate_nelements_at_cpu,
    src/utils/parallel_tools.f90:46-63; used for the data in sensitivity_gravmag.F90:179-189): contiguous ranges of ndata / nranks
    data, the remainder to the last rank - the ranks build within (nranks - 1) rows of each other (round 2 dealt whole 2048-row
    blocks: 7 vs 6 blocks on 8 ranks at the headline size, a 14 % imbalance).  Returns the nranks + 1 range starts."""
    return np.array([row_range(ndata, r, nranks)[0] for r in range(nranks)] + [ndata], np.int64)


def _p2p(tensors_to_send, recv_specs, backend):
    """tensors_to_send: list of (dst_rank, tensor); recv_specs: list of (src_rank, tensor) filled in place.
    NCCL moves device tensors; gloo (CPU tests / single-GPU rehearsal) stages through host memory."""
    import torch
    import torch.distributed as dist
    ops, staged = [], []
    for dst, t in tensors_to_send:
        tt = t if backend == "nccl" else t.cpu()
        staged.append(tt)
        ops.append(dist.P2POp(dist.isend, tt, dst))
    for src, t in recv_specs:
        tt = t if backend == "nccl" else torch.empty(t.shape, dtype=t.dtype)
        staged.append((t, tt))
        ops.append(dist.P2POp(dist.irecv, tt, src))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for item in staged:
        if isinstance(item, tuple) and item[0] is not item[1]:
            item[0].copy_(item[1])


HBM_BYTES_MI355X = 288 * 10 ** 9        # what a plan is checked against when no device is at hand (the tests on CPU)


def memory_plan(ncells, ndata, compression_rate, nranks, nkernels=1, ndata_components=1, nmodel_components=1, dense=False,
                adjoint_copy=True, hbm_bytes=HBM_BYTES_MI355X, exchange=True):
    """Device memory ONE rank of an `nranks`-GPU run needs, phase by phase, BEFORE anything is allocated (bench.py prints it and refuses
    to start a run whose plan does not fit; tests/test_gpu_multirank.py builds one rank's share at full size and checks the measured peak
    against it).  Upper bounds in bytes, from the sizes the library allocates (csrc/build.hip, csrc/matrix.hip):
      grid          six coordinate arrays + the column weight + the per-column histogram                         60 B per cell
      build_work    six line buffers of <= 32 lines / 2 GiB (three batches in flight: generator, wavelet passes, statistics; the select's
                    candidates and the compaction's slots), one row block in ELL form (2048 rows x K x 8 B)
      row_store     (row-parallel build) this rank's ndata / nranks data x all columns: K entries x 8 B per line, until the relayout is done
      share         the tiles of this rank's column range: 5.625 B per stored entry + 3 % for padding, row markers and the chunk table
      relayout      per row block the received pieces and the packed pieces to send: <= 2 x 2048 rows x K x 8 B
      copy          the transposed copy of the share (the adjoint as the forward kernel) + the scratch of one panel of its build (<= 15 GB)
      lsqr          u (replicated), v, w, x, damping rows and right-hand side of the local columns
    `peak` is the largest phase; joint inversions (nkernels = 2) keep the first kernel's share + copy while the second is built."""
    N, D, P = int(ncells), int(ndata), int(nranks)
    ncd, ncm = int(ndata_components), int(nmodel_components)
    nlines = D * ncd
    K = N if dense else max(1, int(compression_rate * N))
    nnz_kernel = float(nlines) * K * ncm
    grid = 60 * N
    lines = max(1, min(32, (1 << 31) // (8 * N)))
    build_work = (1 if dense else 6.25) * lines * N * 8 + (0 if dense else (2048 + ncd) * K * ncm * 8) + (64 << 20)
    rows_loc = -(-D // P) * ncd
    row_store = rows_loc * K * ncm * 8 if (exchange and P > 1 and not dense) else 0
    share = (4.0 * nnz_kernel / P) if dense else (5.625 * 1.03 * nnz_kernel / P + (64 << 20))
    relayout = 2 * 2048 * K * ncm * 8 if row_store else 0
    ncl = -(-N // P) * ncm
    lsqr = 8 * (3 * (nlines * nkernels + 1) + nkernels * ncl * 8) + (32 << 20)
    runtime = 3.0e9                     # HIP / RCCL context, kernel code, the allocators' slack

    def phases_with(copy):
        # panels of ~9e8 entries by the mean density (a dense band of columns holds up to half as many again): (column, value) pairs 8 B +
        # 2 B of slot scratch per entry, the per-row tile index, the counts
        scratch = 0 if not copy else min(nnz_kernel / P, 9.0e8) * 15.0 + 1.5e9
        resident, ph = 0.0, {}          # resident: kernels finished earlier (joint inversion)
        for k in range(nkernels):
            tag = "" if nkernels == 1 else "_kernel%d" % (k + 1)
            ph["build" + tag] = grid + runtime + resident + build_work + row_store + (0 if row_store else share + copy + scratch)
            if row_store:
                ph["relayout" + tag] = grid + runtime + resident + row_store + share + copy + relayout
            if copy:
                ph["adjoint_copy" + tag] = grid + runtime + resident + row_store + share + copy + scratch
            resident += share + copy
        ph["solve"] = grid + runtime + resident + lsqr
        return ph

    copy = 0 if (dense or not adjoint_copy) else 1.08 * share
    with_copy, without = phases_with(copy), phases_with(0)
    limit = 0.97 * hbm_bytes
    copy_fits = bool(copy) and max(with_copy.values()) <= limit
    chosen = with_copy if copy_fits else without
    return {"ranks": P, "cells": N, "data": D, "kernels": nkernels, "entries_per_line": K, "nnz_per_kernel": nnz_kernel,
            "bytes": {"grid": grid, "build_work": build_work, "row_store": row_store, "share": share, "relayout": relayout, "copy": copy, "copy_scratch": (min(nnz_kernel / P, 9.0e8) * 15.0 + 1.5e9) if copy else 0,
                      "lsqr": lsqr, "runtime": runtime},
            # the transposed copy is optional (automatic mode gives it up when it does not fit: the adjoint then runs on the tiles of S)
            "adjoint_copy_fits": copy_fits,
            "phases_GB": {k: round(v / 1e9, 2) for k, v in chosen.items()}, "peak_GB": round(max(chosen.values()) / 1e9, 2),
            "peak_with_adjoint_copy_GB": round(max(with_copy.values()) / 1e9, 2) if copy else None,
            "hbm_GB": round(hbm_bytes / 1e9, 1), "fits": bool(max(without.values()) <= limit)}


def build_partitioned_exchange(ctx, rank, nranks, Xdata, Ydata, Zdata, column_weight, compression_type, compression_rate,
                               problem_weight=1.0, data_weight=None, mag_field=None, get_partition=None, device_index=0,
                               nmodel_components=1, comm=None, data_type=1, ndata_components=1):
    """Row-parallel build + relayout: every rank compresses only ITS blocks of data (all columns, kept row-major on
    the device), the per-column histogram is all-reduced, the reference's greedy rule gives the column ranges, and each
    row block is then cut into column ranges and sent to the owners, who lay their pieces out as tiles.  Every row is
    computed once; the matrix crosses the links once (the reference does this through SENSIT files and a rank-0
    MPI_Scatterv per row: sensitivity_gravmag.F90:179-189, :306-309, :795-830).
    Several data components (full gradient tensor, three-component magnetic data): the matrix has ndata_components rows per
    datum (row = i*ncd + d); data are dealt out in equal contiguous ranges (data_row_partition), a row block of the matrix may have
    several contributors.
    nmodel_components = 3 (magnetisation vector): a rank owns its cell range of every component; the pieces carry component k
    at k*(cells of the range) + cell."""
    import torch
    from .sensitivity import get_load_balancing_nelements
    if comm is None:                    # hosts that set the hooks themselves (tests): host-driven steps through torch.distributed
        comm = HostComm(ctx, rank, nranks, device_index, False)
    dev = torch.device("cuda", device_index)
    N = ctx.nelements_total
    nd = len(Xdata)
    ncd = int(ndata_components)
    nrows = nd * ncd                                                # matrix rows
    RB = ctx.ROW_BLOCK
    dstart = data_row_partition(nd, nranks)                         # my data: [dstart[rank], dstart[rank + 1])
    r0, r1 = int(dstart[rank]), int(dstart[rank + 1])
    nloc = max(0, r1 - r0)
    m0 = [int(d) * ncd for d in dstart]                             # first matrix row of every rank
    # 1. my rows, all columns
    if nloc > 0:
        dw = None if data_weight is None else np.asarray(data_weight).reshape(-1)[r0 * ncd:r1 * ncd]
        res = ctx.rowstore_build(Xdata[r0:r1], Ydata[r0:r1], Zdata[r0:r1], column_weight, compression_type, compression_rate,
                                 problem_weight, dw, mag_field, data_type=data_type, ndata_components=ncd,
                                 nmodel_components=nmodel_components)
        hist, err = res["nnz_hist"].astype(np.int64), res["error_sum"]
    else:
        hist, err = np.zeros(N, np.int64), 0.0
    # 2. partition (sensitivity_gravmag.F90:322, :470-524)
    hist = comm.allreduce_host(hist)
    err = float(comm.allreduce_host(np.array([err]))[0])
    nel, nnz = (get_partition or get_load_balancing_nelements)(hist.astype(np.int32), nranks)
    bounds = np.concatenate([[0], np.cumsum(np.asarray(nel, np.int64))])
    c0, c1 = int(bounds[rank]), int(bounds[rank + 1])
    # 3. who sends how much of which row to whom
    counts_loc = ctx.rowstore_counts(nloc * ncd, bounds) if nloc > 0 else np.zeros((0, nranks), np.int32)
    maxloc = int(max(np.diff(dstart))) * ncd
    pad = np.zeros((maxloc, nranks), np.int32)
    pad[:nloc * ncd] = counts_loc
    gathered = comm.allgather_host(pad)
    counts = np.zeros((nrows, nranks), np.int32)                    # counts[matrix row, dest]
    for r in range(nranks):
        a, b = m0[r], m0[r + 1]
        if b > a:
            counts[a:b] = gathered[r][:b - a]
    assert int(counts[:, rank].sum()) == int(nnz[rank]), (counts[:, rank].sum(), nnz[rank])
    # 4. relayout, matrix row block by row block.  The rows of a block come from the rank(s) that built them - a block that
    # straddles a boundary of the data partition has two (rarely more) contributors, whose pieces are contiguous row sub-ranges in
    # rank order, so they land one behind the other in the receive buffer and form the packed row block.
    ctx.matrix_begin(nrows, nmodel_components * (c1 - c0), int(nnz[rank]))
    m0a = np.asarray(m0, np.int64)
    for b in range((nrows + RB - 1) // RB):
        ga, gb = b * RB, min((b + 1) * RB, nrows)
        n_in = int(counts[ga:gb, rank].sum())
        rc = torch.empty(max(n_in, 1), dtype=torch.int32, device=dev)
        rv = torch.empty(max(n_in, 1), dtype=torch.float32, device=dev)
        sends, recvs, keep = [], [], []
        off = 0
        first = int(np.searchsorted(m0a[1:], ga, side="right"))
        for o in range(first, nranks):
            sa, sb = max(ga, m0[o]), min(gb, m0[o + 1])
            if sb <= sa:
                if m0[o] >= gb:
                    break
                continue
            n_piece = int(counts[sa:sb, rank].sum())                 # what I receive of owner o's rows
            if o == rank:
                for d in range(nranks):
                    n_out = int(counts[sa:sb, d].sum())
                    if n_out == 0:
                        continue
                    if d == rank:
                        got = ctx.rowstore_pack(sa - m0[rank], sb - sa, int(bounds[d]), int(bounds[d + 1]), rc[off:off + n_out],
                                                rv[off:off + n_out], n_out)
                        assert got == n_out
                        continue
                    sc = torch.empty(n_out, dtype=torch.int32, device=dev)
                    sv = torch.empty(n_out, dtype=torch.float32, device=dev)
                    got = ctx.rowstore_pack(sa - m0[rank], sb - sa, int(bounds[d]), int(bounds[d + 1]), sc, sv, n_out)
                    assert got == n_out
                    keep += [sc, sv]
                    sends += [(d, sc), (d, sv)]
            elif n_piece > 0:
                recvs += [(o, rc[off:off + n_piece]), (o, rv[off:off + n_piece])]
            off += n_piece
        assert off == n_in, (off, n_in)
        comm.exchange(sends, recvs)                                  # synchronises before (pack kernels) and after
        ctx.matrix_append_rows(ga, rc, rv, counts[ga:gb, rank])
    ctx.matrix_finish()
    ctx.rowstore_free()
    return dict(col_range=(c0, c1), nelements_at_cpu=nel, nnz_at_cpu=nnz, nnz_total=int(hist.sum()),
                comp_error=err / (nd * ncd * nmodel_components))


def build_partitioned(ctx, rank, nranks, Xdata, Ydata, Zdata, column_weight, compression_type, compression_rate,
                      problem_weight=1.0, data_weight=None, get_partition=None, comm=None):
    """Column-partitioned sensitivity build for `nranks` GPUs.

    Phase 1 (row-parallel, like the reference's P1 decomposition, sensitivity_gravmag.F90:179-189): every rank compresses
    its share of the observation rows only to count the per-column non-zeros; the histogram is all-reduced
    (:322) and the reference's greedy rule gives the column ranges (:470-524).
    Phase 2: every rank builds the rows again keeping only its own column range, straight into its tiled matrix - no
    disk round trip and no rank-0 scatter (:648-883).  (A row-parallel build with an all-to-all relayout would avoid
    recomputing rows; see DESIGN.md "Multi-GPU".)
    Returns dict(col_range, nelements_at_cpu, nnz_at_cpu, nnz_total, comp_error)."""
    from .sensitivity import get_load_balancing_nelements
    N = ctx.nelements_total
    nd = len(Xdata)
    if nranks == 1:
        res = ctx.calculate_sensit(Xdata, Ydata, Zdata, column_weight, compression_type, compression_rate, problem_weight, data_weight)
        return dict(col_range=(0, N), nelements_at_cpu=np.array([N]), nnz_at_cpu=np.array([res["nnz"]]), nnz_total=res["nnz"],
                    comp_error=res["comp_error"])
    r0, r1 = row_range(nd, rank, nranks)
    if r1 > r0:
        dw = None if data_weight is None else data_weight[r0:r1]
        res1 = ctx.calculate_sensit(Xdata[r0:r1], Ydata[r0:r1], Zdata[r0:r1], column_weight, compression_type, compression_rate,
                                    problem_weight, dw, col_range=(0, 0), want_hist=True)
        hist, err = res1["nnz_hist"].astype(np.int64), res1["error_sum"]
    else:
        hist, err = np.zeros(N, np.int64), 0.0
    if comm is None:
        comm = HostComm(ctx, rank, nranks, getattr(ctx, "device", 0), False)
    hist = comm.allreduce_host(hist)
    err = float(comm.allreduce_host(np.array([err]))[0])
    nel, nnz = (get_partition or get_load_balancing_nelements)(hist.astype(np.int32), nranks)
    c0, c1 = column_ranges(nel)[rank]
    ctx.matrix_reserve(int(nnz[rank]))          # (the histogram says what the range holds: not the rows x K of a whole kernel)
    res2 = ctx.calculate_sensit(Xdata, Ydata, Zdata, column_weight, compression_type, compression_rate, problem_weight, data_weight,
                                col_range=(c0, c1))
    assert res2["nnz"] == int(nnz[rank]), (res2["nnz"], nnz[rank])
    return dict(col_range=(c0, c1), nelements_at_cpu=nel, nnz_at_cpu=nnz, nnz_total=int(hist.sum()), comp_error=err / nd)
