"""world_size-2 tests of the multi-rank path on CPU (gloo): partition + column-partitioned build orchestration
(tomofast-x_amd/distributed.py, with an oracle-backed stand-in for the GPU context) and the column-partitioned LSQR
scheme of lsqr.hip (restated in numpy in multirank_model.py) against the single-rank oracle."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


class OracleCtx:
    """Stand-in with the Context.calculate_sensit interface, computing on the CPU oracle (tests only)."""

    def __init__(self, grid, dims):
        self.grid, self.dims = grid, dims
        self.nelements_total = int(np.prod(dims))
        self.S = None

    def matrix_reserve(self, nnz_upper):
        self.reserved = int(nnz_upper)          # (what the next column-range build may hold: checked below)

    def calculate_sensit(self, X, Y, Z, cw, ctype, rate, pw=1.0, dw=None, col_range=None, want_hist=False):
        import oracle_lib as orc
        import multirank_model as mm
        obs = np.stack([X, Y, Z], 1)
        rp, cols, vals, hist, err = orc.build_matrix_grav(self.grid, self.dims, cw, obs, ctype, rate)
        c0, c1 = (0, self.nelements_total) if col_range is None else col_range
        self.S = mm.column_slice((rp, cols, vals), c0, c1)
        return dict(nnz=int(self.S[0][-1]), error_sum=err * len(X), comp_error=err, nnz_hist=hist if want_hist else None)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        import oracle_lib as orc
        import multirank_model as mm
        g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))
        grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
        dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
        N = int(np.prod(dims))
        obs = g["obs"]
        cw = g["np1_column_weight"]
        ctx = OracleCtx(grid, dims)
        part = tfx.distributed.build_partitioned(ctx, rank, world, obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, float(g["rate"]))
        # the reference's own 2-rank partition of the same problem
        assert np.array_equal(part["nelements_at_cpu"], g["np2_nelements_at_cpu"])
        assert np.array_equal(part["nnz_at_cpu"], g["np2_nnz_at_cpu"])
        assert part["nnz_total"] == int(g["np1_nnz_total"])
        assert abs(part["comp_error"] - float(g["np1_comp_error"])) <= 1e-12
        c0, c1 = part["col_range"]
        assert int(ctx.S[0][-1]) == int(g["np2_nnz_at_cpu"][rank])

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a, np.float64).copy())
            dist.all_reduce(t)
            return t.numpy()

        # column-partitioned LSQR vs the single-rank oracle on the same system [S; alpha I]
        S_full = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
        b = g["np1_data_observed"]
        alpha = np.float32(float(g["alpha"]))
        rng = np.random.default_rng(5)
        rhs_full = rng.standard_normal(N) * 1e-9
        Cm = orc.diag_csr(np.full(N, alpha, np.float32))
        # this 30-row system is ill-conditioned: summation-order differences (1e-16) reach 5e-6 by iteration 10 even on
        # one rank, so the mid-convergence comparison is loose and the early one tight
        for niter, tol in ((3, 1e-12), (25, 1e-4)):
            x_loc, it, r = mm.lsqr_column_partitioned(ctx.S, c1 - c0, b.size, b, [np.full(c1 - c0, alpha, np.float32)],
                                                      [rhs_full[c0:c1]], niter, 1e-13, rank, allreduce)
            x_ref, it_ref, r_ref = orc.lsqr(S_full, Cm, N, np.concatenate([b, rhs_full]), niter)
            assert it == it_ref == niter
            err = np.linalg.norm(x_loc - x_ref[c0:c1]) / np.linalg.norm(x_ref)
            assert err <= tol, (niter, err)
            assert abs(r - r_ref) <= (1e-12 if niter == 3 else 1e-3) * r_ref
        # forward data: partial products summed over ranks = full product (model.F90:288-293)
        xw = rng.standard_normal(N)
        d = allreduce(orc.spmv(*ctx.S, xw[c0:c1]))
        assert np.allclose(d, orc.spmv(*S_full, xw), rtol=1e-12, atol=1e-20)
        # host helpers
        assert tfx.distributed.row_range(30, rank, world) == ((0, 15) if rank == 0 else (15, 30))
        h = tfx.distributed.allreduce_numpy(np.arange(4, dtype=np.int64) + rank)
        assert np.array_equal(h, np.arange(4) * 2 + 1)
        q.put((rank, "ok"))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_partition_build_and_lsqr_scheme():
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_block_partition_helpers():
    tfx = importlib.import_module("tomofast-x_amd")
    d = tfx.distributed
    # calculate_nelements_at_cpu, src/utils/parallel_tools.f90:46-63: remainder goes to the last rank
    assert [d.calculate_nelements_at_cpu(10, r, 3) for r in range(3)] == [3, 3, 4]
    assert [d.row_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert d.column_ranges([3, 5, 2]) == [(0, 3), (3, 8), (8, 10)]


# ---- the start-up ladder of distributed.setup_comm, with a scripted stand-in for the GPU context -------------------------------
class LadderCtx:
    """Records what the ladder does to the context; `init` scripts what tfx_comm_init_rccl does on this rank."""

    def __init__(self, init):
        self.init, self.calls, self.has_comm = init, [], False
        self.rank, self.nranks = 0, 1

    def comm_unique_id(self):
        self.calls.append("unique_id")
        return bytes(range(128))

    def comm_init_rccl(self, uid, rank, nranks):
        self.calls.append("init")
        assert uid == bytes(range(128))                      # the 128 bytes of rank 0 arrived over the control channel
        if self.init == "raise":
            raise RuntimeError("scripted ncclCommInitRank failure")
        if self.init == "hang":
            import time
            time.sleep(30.0)
        self.has_comm = True

    def comm_abort(self):
        self.calls.append("abort")
        self.has_comm = False

    def debug_set(self, key, value=0):
        self.calls.append("debug_set %s=%s" % (key, value))
        return 0

    def comm_info(self):
        return dict(rccl_ranks=0, rccl_rank=-1, rccl_device=-1, rccl_version=0, librccl="scripted")

    def set_allreduce(self, hook, rank, nranks):
        self.calls.append("set_allreduce")
        self.rank, self.nranks = rank, nranks

    def set_allgatherv(self, fn):
        self.calls.append("set_allgatherv")


def _ladder_worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    import time
    import types
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        # no GPU here: every rank "drives" its own device (a distinct PCI address per rank), or all the same one
        shared = scenario == "shared_gpu"
        torch.cuda.get_device_properties = lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=7 if shared else 10 + rank, pci_device_id=0, uuid="x")
        init = {"one_rank_fails": "raise" if rank == 1 else "ok", "one_rank_hangs": "hang" if rank == 0 else "ok"}.get(scenario, "ok")
        ctx = LadderCtx(init)
        t0 = time.time()
        comm = tfx.distributed.setup_comm(ctx, rank, world, device_index=0 if shared else rank, want_rccl=True, init_timeout=3.0)
        q.put((rank, comm.rccl, comm.report, ctx.calls, ctx.has_comm, time.time() - t0))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, "error", traceback.format_exc(), [], False, 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["shared_gpu", "one_rank_fails", "one_rank_hangs"])
def test_startup_ladder_moves_all_ranks_to_the_hooks_together(scenario):
    """distributed.setup_comm on 2 ranks over gloo with a scripted context: whatever goes wrong on ONE rank - the ranks share a GPU,
    tfx_comm_init_rccl raises, or it never returns - EVERY rank ends on the hook rung, no rank keeps a communicator (the rank whose
    own rendezvous succeeded aborts it), the reason is in the report, and a hang costs the timeout, not the run."""
    world, port = 2, 29581 + ["shared_gpu", "one_rank_fails", "one_rank_hangs"].index(scenario)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_ladder_worker, args=(r, world, port, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, rccl, report, calls, has_comm, dt in res:
        assert rccl is False, (rank, rccl, report)
        assert report["path"].startswith("torch.distributed hooks"), report
        assert not has_comm and "set_allreduce" in calls and "set_allgatherv" in calls, calls
        assert report["ladder"][-1]["ok"] is False
        assert dt < 25.0, dt
    ladder = [r[2]["ladder"] for r in res]
    assert ladder[0] == ladder[1]                                   # the same story on every rank
    if scenario == "shared_gpu":
        assert ladder[0][0]["stage"] == "pre-flight" and "share a GPU" in ladder[0][0]["why"]
        assert all("init" not in r[3] for r in res)                 # nobody entered the rendezvous
    else:
        assert ladder[0][0] == {"stage": "pre-flight", "ok": True} and ladder[0][1]["stage"] == "tfx_comm_init_rccl"
        assert ("scripted ncclCommInitRank failure" in ladder[0][1]["why"]) == (scenario == "one_rank_fails")
        assert ("timed out" in ladder[0][1]["why"]) == (scenario == "one_rank_hangs")
        assert all("abort" in r[3] for r in res)                    # also the rank whose own call succeeded drops its communicator


def test_memory_plan_of_the_baseline_configurations():
    """tomofast-x_amd/distributed.py::memory_plan (what bench.py prints and checks before it allocates): BASELINE config 5 on 8 GPUs
    and config 4 (two kernels) on 4 GPUs fit a 288 GB MI355X with the transposed copies; the headline on ONE GPU fits with its copy
    (measured: 230 GB resident); config 3 and config 4 on one GPU fit only without the copy (what the automatic mode does there);
    a run that can not fit is refused."""
    tfx = importlib.import_module("tomofast-x_amd")
    mp = tfx.distributed.memory_plan
    c5 = dict(ncells=256 * 256 * 152, ndata=316 * 316, compression_rate=0.02)
    c4 = dict(ncells=512 * 512 * 128, ndata=256 * 256, compression_rate=0.01)
    peaks = []
    for P in (1, 2, 4, 8):
        r = mp(nranks=P, **c5)
        assert r["fits"] and r["adjoint_copy_fits"], r
        peaks.append(r["peak_GB"])
        if P > 1:
            assert set(r["phases_GB"]) == {"build", "relayout", "adjoint_copy", "solve"}
            assert r["bytes"]["row_store"] == -(-99856 // P) * int(0.02 * 9961472) * 8
    assert peaks == sorted(peaks, reverse=True) and peaks[-1] < 70.0             # 8 GPUs: a fifth of the device
    assert 225.0 <= mp(nranks=1, **c5)["phases_GB"]["solve"] <= 260.0             # one GPU, both copies resident (measured 230 GB)
    r = mp(nranks=4, nkernels=2, **c4)
    assert r["fits"] and r["adjoint_copy_fits"] and r["peak_GB"] < 220.0, r
    assert "relayout_kernel2" in r["phases_GB"]
    r = mp(nranks=1, **c4)                                                        # config 3
    assert r["fits"] and not r["adjoint_copy_fits"] and r["peak_with_adjoint_copy_GB"] > 0.97 * 288.0
    r = mp(nranks=1, nkernels=2, **c4)                                            # config 4 on one GPU: ran in round 3 (256 GB resident)
    assert r["fits"] and not r["adjoint_copy_fits"] and 250.0 < r["peak_GB"] < 279.5
    assert not mp(nranks=1, nkernels=2, ncells=512 * 512 * 128, ndata=256 * 256, compression_rate=0.02)["fits"]
    r = mp(nranks=1, dense=True, ncells=256 * 256 * 64, ndata=4096, compression_rate=1.0)     # config 2
    assert r["fits"] and 68.0 < r["phases_GB"]["solve"] < 75.0


# ---- the step runner of bench.py --selftest (distributed.AgreedSteps): failure on ONE rank is a failure on all ----------------------
def _steps_worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    import time
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        runner = tfx.distributed.AgreedSteps(True)

        def first(info):
            info["value"] = 7

        def second(info):
            if scenario == "one_rank_raises" and rank == 1:
                raise RuntimeError("scripted failure on rank 1")
            if scenario == "one_rank_hangs" and rank == 0:
                time.sleep(20.0)
            info["seen"] = rank

        def third(info):
            info["reached"] = True

        t0 = time.time()
        runner.run("first", first, 5.0)
        runner.run("second", second, 2.0)
        runner.run("third", third, 5.0)
        runner.run("rccl only", third, 5.0, applicable=False, why_not="not applicable: hooks")
        q.put((rank, runner.ok, runner.steps, time.time() - t0))
    except Exception:      # noqa
        import traceback
        q.put((rank, "error", traceback.format_exc(), 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["all_fine", "one_rank_raises", "one_rank_hangs"])
def test_selftest_steps_fail_on_every_rank_together(scenario):
    """distributed.AgreedSteps (the step runner of `bench.py --selftest` / `comm.selftest`) on 2 ranks over gloo: a step that raises or
    hangs on ONE rank is recorded as failed on BOTH with the rank and the reason, a hang costs the step's timeout, and the steps behind
    the failure are recorded as not run - nobody walks into the next collective alone."""
    world, port = 2, 29591 + ["all_fine", "one_rank_raises", "one_rank_hangs"].index(scenario)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_steps_worker, args=(r, world, port, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, ok, steps, dt in res:
        assert ok != "error", steps
        names = [s["step"] for s in steps]
        assert names == ["first", "second", "third", "rccl only"], names
        assert steps[0]["ok"] is True and steps[0]["value"] == 7
        assert dt < 15.0, dt
        if scenario == "all_fine":
            assert ok is True and steps[1]["ok"] is True and steps[2]["ok"] is True and steps[2]["reached"] and steps[3]["ok"] is None
            assert steps[3]["why"] == "not applicable: hooks"
        else:
            assert ok is False and steps[1]["ok"] is False and steps[2]["ok"] is None and "not run" in steps[2]["why"] and "not run" in steps[3]["why"]
            if scenario == "one_rank_raises":
                assert steps[1]["why"].startswith("rank 1:") and "scripted failure on rank 1" in steps[1]["why"]
            else:
                assert steps[1]["why"].startswith("rank 0:") and "timed out" in steps[1]["why"]
    assert [s.get("why") for s in res[0][2]] == [s.get("why") for s in res[1][2]]         # the same story on both ranks


def _fallback_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        ctx = LadderCtx("ok")
        ctx.has_comm = True
        comm = tfx.distributed.HostComm(ctx, rank, world, 0, True)
        comm.report = {"path": "RCCL inside libtfx.so (tfx_comm_init_rccl)", "ladder": [{"stage": "pre-flight", "ok": True}], "rccl_ranks": world}
        new = tfx.distributed.fall_back_to_hooks(ctx, comm, rank, world, 0, "rank 1: scripted")
        q.put((rank, new.rccl, new.report, ctx.calls, ctx.has_comm, (ctx.rank, ctx.nranks)))
    except Exception:      # noqa
        import traceback
        q.put((rank, "error", traceback.format_exc(), [], False, None))
    finally:
        dist.destroy_process_group()


def test_failed_selftest_moves_all_ranks_from_rccl_to_the_hooks():
    """distributed.fall_back_to_hooks (what bench.py does when comm.selftest fails on the RCCL rung): the communicator is aborted, the
    forced-collectives switch cleared, the torch.distributed hooks installed with the rank / rank count, and the report says why."""
    world, port = 2, 29597
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, rccl, report, calls, has_comm, who in res:
        assert rccl is False, report
        assert calls[0] == "abort" and "debug_set force_collectives=0" in calls and "set_allreduce" in calls and "set_allgatherv" in calls, calls
        assert not has_comm and who == (rank, world)
        assert report["path"].startswith("torch.distributed hooks (gloo) after the RCCL self-test failed: rank 1: scripted") and report["rccl_ranks"] == 0
        assert report["ladder"] == [{"stage": "pre-flight", "ok": True}]
