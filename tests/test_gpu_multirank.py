"""The real multi-rank path of libtfx.so on the GPU box: 2 processes share GPU 0, torch.distributed with gloo (NCCL/RCCL
needs one GPU per rank; the hook, the column partition, the rank-local constraint rows and both reductions are the same
code that runs over RCCL at N > 1).  Compared with the single-rank oracle on the same problem."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _worker(rank, world, port, q, own_stream=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        import oracle_lib as orc
        g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))
        dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
        N = int(np.prod(dims))
        obs = g["obs"]
        cw = g["np1_column_weight"]
        # own_stream: the library works on a caller-supplied NON-null stream; the hook has to run its reductions (and staging
        # copies) on that stream, not on torch's current one (tfx.h: "enqueued on `stream`")
        side = torch.cuda.Stream(device=0) if own_stream else None
        ctx = tfx.Context(0, stream=side.cuda_stream) if own_stream else tfx.Context(0)
        ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
        hook = tfx.distributed.TorchAllreduce(0)
        ctx.set_allreduce(hook, rank, world)
        part = tfx.distributed.build_partitioned(ctx, rank, world, obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, float(g["rate"]))
        c0, c1 = part["col_range"]
        # the GPU-built histogram may differ from the reference's by a threshold tie or two
        assert np.all(np.abs(part["nelements_at_cpu"] - g["np2_nelements_at_cpu"]) <= 2)
        assert abs(part["nnz_total"] - int(g["np1_nnz_total"])) <= 2 * obs.shape[0]
        # identical matrices on both sides for the solver comparison: upload this rank's slice of the reference's rows
        import multirank_model as mm
        S_full = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
        S_loc = mm.column_slice(S_full, c0, c1)
        ctx.matrix_upload_csr(obs.shape[0], c1 - c0, *S_loc)
        b = g["np1_data_observed"]
        alpha = np.float32(float(g["alpha"]))
        rng = np.random.default_rng(5)
        rhs_full = rng.standard_normal(N) * 1e-9
        Cm = orc.diag_csr(np.full(N, alpha, np.float32))
        # (25 iterations stop mid-convergence: deterministic products, so that the outcome does not hang on the run-dependent order
        # of the LDS atomics - the difference to the single-rank oracle is then a fixed number)
        ctx.debug_set("deterministic", 1)
        for niter, tol in ((3, 1e-11), (25, 1e-3)):
            x_loc, it, r = ctx.lsqr_solve_sensit(b, niter, 1e-13, 0.0, 0.0, [np.full(c1 - c0, alpha, np.float32)], [rhs_full[c0:c1]])
            x_ref, it_ref, r_ref = orc.lsqr(S_full, Cm, N, np.concatenate([b, rhs_full]), niter)
            assert it == it_ref == niter
            err = np.linalg.norm(x_loc - x_ref[c0:c1]) / np.linalg.norm(x_ref)
            assert err <= tol, (niter, err)
        # forward data through the hook (model.F90:288-293)
        xw = rng.standard_normal(N)
        d = ctx.calc_data(xw[c0:c1], 1.0, None)
        assert np.allclose(d, orc.spmv(*S_full, xw), rtol=1e-11, atol=1e-20)
        ctx.close()
        q.put((rank, "ok zero_copy=%s" % hook.zero_copy))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("own_stream", [False, True])
def test_two_ranks_on_one_gpu(own_stream):
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + os.getpid() % 2000 + (7 if own_stream else 0)
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q, own_stream)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg.startswith("ok"), "rank %d: %s" % (rank, msg)
    print(res)


def _worker_exchange(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        nx, ny, nz, ox, oy = 32, 24, 12, 72, 64             # 4608 observations = 3 row blocks of 2048
        grid = tfx.synthetic.grid(nx, ny, nz)
        xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
        ctx = tfx.Context(0)
        ctx.set_grid(nx, ny, nz, *grid)
        cw = ctx.calculate_depth_weight()
        part = tfx.distributed.build_partitioned_exchange(ctx, rank, world, xs, ys, zs, cw, 2, 0.05)
        c0, c1 = part["col_range"]
        A = ctx.matrix_download_csr()
        info = ctx.matrix_info()
        assert info["nnz"] == int(part["nnz_at_cpu"][rank]) == int(A[0][-1])
        # the same column range built the direct way (every row on this rank)
        res = ctx.calculate_sensit(xs, ys, zs, cw, 2, 0.05, col_range=(c0, c1))
        B = ctx.matrix_download_csr()
        assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and A[2].tobytes() == B[2].tobytes()
        assert abs(part["nnz_total"] - xs.size * int(0.05 * nx * ny * nz)) <= 2 * xs.size
        # and it is a partition of all columns
        ranges = [None] * world
        dist.all_gather_object(ranges, (c0, c1))
        assert ranges[0][0] == 0 and ranges[-1][1] == nx * ny * nz and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        # magnetisation-vector kernel (3 model components, TMI data): a rank owns its cell range of every component
        field = (63.0, -11.0, 4.0, 51000.0)
        nsub = 2100                                       # 2 row blocks
        part = tfx.distributed.build_partitioned_exchange(ctx, rank, world, xs[:nsub], ys[:nsub], zs[:nsub], cw, 1, 0.03, mag_field=field,
                                                          nmodel_components=3)
        c0, c1 = part["col_range"]
        A = ctx.matrix_download_csr()
        assert ctx.matrix_info()["ncols"] == 3 * (c1 - c0) and int(A[0][-1]) == int(part["nnz_at_cpu"][rank])
        ctx.calculate_sensit(xs[:nsub], ys[:nsub], zs[:nsub], cw, 1, 0.03, col_range=(c0, c1), mag_field=field, nmodel_components=3)
        B = ctx.matrix_download_csr()
        assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and A[2].tobytes() == B[2].tobytes()
        # several DATA components: the matrix has ncd rows per datum, data are dealt out in blocks of 2048 (= ncd row blocks):
        # full gradient tensor (6 lines per observation) and three-component magnetic data with a magnetisation-vector model,
        # problem weight and per-line data weights fused into the rows on both paths
        for kw in (dict(data_type=2, ndata_components=6), dict(mag_field=field, ndata_components=3, nmodel_components=3)):
            nsub = 2100                                   # 2 blocks of data -> 12600 / 6300 matrix rows
            ncd = kw["ndata_components"]
            dw = 0.5 + np.random.default_rng(4).random((nsub, ncd))
            part = tfx.distributed.build_partitioned_exchange(ctx, rank, world, xs[:nsub], ys[:nsub], zs[:nsub], cw, 2, 0.04,
                                                              problem_weight=1.75, data_weight=dw, **kw)
            c0, c1 = part["col_range"]
            A = ctx.matrix_download_csr()
            assert ctx.matrix_info()["nrows"] == nsub * ncd and int(A[0][-1]) == int(part["nnz_at_cpu"][rank])
            ctx.calculate_sensit(xs[:nsub], ys[:nsub], zs[:nsub], cw, 2, 0.04, 1.75, dw, col_range=(c0, c1), **kw)
            B = ctx.matrix_download_csr()
            assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and A[2].tobytes() == B[2].tobytes(), kw
        ctx.close()
        q.put((rank, "ok"))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])          # 5: three row blocks on five ranks - two ranks build nothing and only receive
def test_row_parallel_build_with_relayout(world):
    """The exchange-based multi-GPU build (row-parallel compression, all-reduce of the histogram, point-to-point relayout)
    gives every rank exactly the matrix a direct build of its column range gives."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29600 + os.getpid() % 2000 + world
    procs = [ctxm.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_joint(rank, world, port, q):
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
        g = np.load(os.path.join(GOLDEN, "e2e_joint.npz"))
        dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
        N = int(np.prod(dims))
        nel = g["np2_nelements_at_cpu"]                       # the reference's joint 2-rank partition (both histograms added)
        c0 = int(nel[:rank].sum())
        c1 = c0 + int(nel[rank])
        nloc = c1 - c0
        ctx = tfx.Context(0)
        ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
        hook = tfx.distributed.TorchAllreduce(0)
        ctx.set_allreduce(hook, rank, world)
        S = []
        for i, tag in enumerate(("grav", "magn")):
            vals = (g["np1_%s_vals" % tag] * np.float32(g["pw"][i])).astype(np.float32)
            Sf = (g["np1_%s_row_ptr" % tag], g["np1_%s_cols" % tag], vals)
            S.append(Sf)
            ctx.select_problem(i)
            ctx.matrix_upload_csr(Sf[0].size - 1, nloc, *mm.column_slice(Sf, c0, c1))
        ctx.select_problem(0)
        nd = [S[0][0].size - 1, S[1][0].size - 1]
        assert ctx.system_dims() == (nd[0] + nd[1], 2 * nloc)
        rng = np.random.default_rng(11)
        b = rng.standard_normal(nd[0] + nd[1])
        alpha = [np.float32(3e-9), np.float32(7e-10)]
        rhs_full = rng.standard_normal(2 * N) * 1e-10
        # local unknowns [m1_loc; m2_loc]; one damping block per problem over its own column block
        diag, rhs = [], []
        for i in range(2):
            dblk = np.zeros(2 * nloc, np.float32)
            dblk[i * nloc:(i + 1) * nloc] = alpha[i]
            r = np.zeros(2 * nloc)
            r[i * nloc:(i + 1) * nloc] = rhs_full[i * N + c0:i * N + c1]
            diag.append(dblk)
            rhs.append(r)
        x_loc, it, r = ctx.lsqr_solve_sensit(b, 12, 1e-13, 0.0, 0.0, diag, rhs)
        # single-rank oracle on blockdiag(S1, S2) with the same two diagonal blocks
        rp = np.concatenate([S[0][0], S[1][0][1:] + S[0][0][-1]])
        Sj = (rp, np.concatenate([S[0][1], S[1][1] + N]).astype(np.int32), np.concatenate([S[0][2], S[1][2]]))
        d1 = np.zeros(2 * N, np.float32); d1[:N] = alpha[0]
        d2 = np.zeros(2 * N, np.float32); d2[N:] = alpha[1]
        c1m, c2m = orc.diag_csr(d1), orc.diag_csr(d2)
        Cm = (np.concatenate([c1m[0], c2m[0][1:] + c1m[0][-1]]), np.concatenate([c1m[1], c2m[1]]), np.concatenate([c1m[2], c2m[2]]))
        z = np.zeros(N)                                                # block b has 2N rows, non-empty only in its own column block
        rhs_rows = np.concatenate([rhs_full[:N], z, z, rhs_full[N:]])
        x_ref, it_ref, r_ref = orc.lsqr(Sj, Cm, 2 * N, np.concatenate([b, rhs_rows]), 12)
        want = np.concatenate([x_ref[c0:c1], x_ref[N + c0:N + c1]])
        assert it == it_ref == 12
        err = np.linalg.norm(x_loc - want) / np.linalg.norm(x_ref)
        assert err <= 1e-6, err
        ctx.close()
        q.put((rank, "ok"))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_joint_system_two_ranks():
    """Joint system blockdiag(S_grav, S_magn) column-partitioned over 2 ranks (each rank holds its cell range of BOTH kernels,
    like the reference's nelements_at_cpu for joint inversion) vs the single-rank oracle."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctxm.Process(target=_worker_joint, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_inversion(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))
        dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
        obs = g["obs"]
        ctx = tfx.Context(0)
        ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
        tfx.distributed.setup_comm(ctx, rank, world, 0)          # gloo here: all-reduce + all-gather hooks through torch.distributed
        cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
        part = tfx.distributed.build_partitioned(ctx, rank, world, obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, float(g["rate"]))
        m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, 1, g["np1_data_observed"], int(g["nmajor"]), int(g["nminor"]),
                                                         alpha=float(g["alpha"]), col_range=part["col_range"])
        ref = g["np2_model_final"]                       # the reference's own 2-rank run
        err = np.linalg.norm(m - ref) / np.linalg.norm(ref)
        assert err <= 1e-6, err
        cost_ref = np.linalg.norm(g["np1_data_final"] - g["np1_data_observed"]) / np.linalg.norm(g["np1_data_observed"])
        assert abs(hist[-1]["cost"] - cost_ref) <= 1e-5 * cost_ref + 1e-16
        ctx.close()
        q.put((rank, "ok"))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_whole_inversion_two_ranks():
    """Major loop on 2 ranks: kernel built per column range, LSQR over the hook, model-update slices gathered and
    inverse-transformed on every rank, forward data all-reduced - vs the reference's final model of its own 2-rank run."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29800 + os.getpid() % 2000
    procs = [ctxm.Process(target=_worker_inversion, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_inversion_spatial(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        import multirank_model as mm
        g = np.load(os.path.join(GOLDEN, "e2e_dgrad.npz"))
        dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
        N = int(np.prod(dims))
        ctx = tfx.Context(0)
        ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
        tfx.distributed.setup_comm(ctx, rank, world, 0)          # gloo here: all-reduce + all-gather hooks through torch.distributed
        nel = g["np2_nelements_at_cpu"]                  # the reference's own 2-rank partition
        c0 = int(nel[:rank].sum())
        c1 = c0 + int(nel[rank])
        S_full = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
        ctx.matrix_upload_csr(S_full[0].size - 1, c1 - c0, *mm.column_slice(S_full, c0, c1))
        m, d, hist = tfx.inversion.solve_problem_gravity(ctx, g["np1_column_weight"], int(g["ctype"]), g["np1_data_observed"],
                                                         int(g["nmajor"]), int(g["nminor"]), alpha=float(g["alpha"]), beta=float(g["beta"]),
                                                         col_range=(c0, c1))
        ref = g["np2_model_final"]                       # the reference's own 2-rank run
        err = np.linalg.norm(m - ref) / np.linalg.norm(ref)
        assert err <= 1e-5, err
        assert np.allclose([h["r"] for h in hist], g["np2_lsqr_r"], rtol=1e-4)
        ctx.close()
        q.put((rank, "ok"))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_whole_inversion_with_gradient_damping_two_ranks():
    """WAVELET_DOMAIN = F on 2 ranks through the host: spatial unknowns per column range (tfx_lsqr_set_partition), the
    gradient-damping rows replicated with each rank's own columns (rows that straddle the cut are summed by the LSQR
    all-reduce), model-update slices gathered - vs the reference's own 2-rank run."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29800 + (os.getpid() + 7) % 2000
    procs = [ctxm.Process(target=_worker_inversion_spatial, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_joint_coupled(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tfx = importlib.import_module("tomofast-x_amd")
        import multirank_model as mm
        for name in ("e2e_joint", "e2e_xgrad", "e2e_clust"):
            g = np.load(os.path.join(GOLDEN, name + ".npz"))
            dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
            N = int(np.prod(dims))
            ctx = tfx.Context(0)
            ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
            tfx.distributed.setup_comm(ctx, rank, world, 0)          # gloo here: all-reduce + all-gather hooks through torch.distributed
            c0, c1 = (0, N // 2 + 3) if rank == 0 else (N // 2 + 3, N)
            probs = []
            for i, tag in enumerate(("grav", "magn")):
                ctx.select_problem(i)
                S = (g["np1_%s_row_ptr" % tag], g["np1_%s_cols" % tag], g["np1_%s_vals" % tag])
                ctx.matrix_upload_csr(S[0].size - 1, c1 - c0, *mm.column_slice(S, c0, c1))
                probs.append(dict(column_weight=g["np1_%s_column_weight" % tag], data_obs=g["np1_%s_data_observed" % tag],
                                  problem_weight=1.0, alpha=float(g["alpha"][i])))
            ctx.select_problem(0)
            kw = {}
            if name == "e2e_xgrad":
                kw["cross_gradient"] = dict(weight=float(g["xgrad_weight"]), der_type=int(g["der_type"]))
            if name == "e2e_clust":
                kw["clustering"] = dict(weight=g["clust_weight"], mixtures=g["mixtures"], opt_type=int(g["opt_type"]),
                                        cell_weights=None if int(g["cons_type"]) == 1 else g["cell_weights"])
            m, d, hist = tfx.inversion.solve_problem_joint(ctx, probs, int(g["ctype"]), int(g["nmajor"]), int(g["nminor"]),
                                                           col_range=(c0, c1), **kw)
            for i, tag in enumerate(("grav", "magn")):
                ref = g["np1_%s_model_final" % tag]
                self_diff = np.linalg.norm(g["np2_%s_model_final" % tag] - ref) / np.linalg.norm(ref)
                err = np.linalg.norm(m[i] - ref) / np.linalg.norm(ref)
                assert err <= max(1e-6, 100.0 * self_diff), (name, tag, err, self_diff)
            ctx.close()
        q.put((rank, "ok"))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_joint_inversions_with_coupling_two_ranks():
    """solve_problem_joint on 2 ranks (each holds a cell range of both kernels): the plain joint system in the wavelet domain,
    and the cross-gradient / clustering couplings with spatial unknowns (constraint rows replicated, each rank's own columns of
    both column blocks) - vs the reference's final models."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29800 + (os.getpid() + 11) % 2000
    procs = [ctxm.Process(target=_worker_joint_coupled, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_spatial(rank, world, port, q):
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
        g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))       # Haar lifting is orthonormal: S W with spatial unknowns == S with W x
        dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
        N = int(np.prod(dims))
        nel = g["np2_nelements_at_cpu"]
        c0 = int(nel[:rank].sum())
        c1 = c0 + int(nel[rank])
        S_full = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
        nd = S_full[0].size - 1
        ctx = tfx.Context(0)
        ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
        tfx.distributed.setup_comm(ctx, rank, world, 0)          # gloo here: all-reduce + all-gather hooks through torch.distributed
        ctx.matrix_upload_csr(nd, c1 - c0, *mm.column_slice(S_full, c0, c1))
        b = g["np1_data_observed"]
        alpha = np.float32(1e-7)
        rng = np.random.default_rng(3)
        rhs_full = rng.standard_normal(N) * 1e-8
        # spatial unknowns on 2 ranks ...
        ctx.lsqr_set_wavelet_domain(False, 1)
        ctx.lsqr_set_partition(c0, 1)
        ctx.debug_set("deterministic", 1)     # fixed accumulation order in the products, so that two solves can be compared tightly
        x_loc, it, r = ctx.lsqr_solve_sensit(b, 15, 1e-13, 0.0, 0.0, [np.full(c1 - c0, alpha, np.float32)], [rhs_full[c0:c1]])
        # the same solve with the slices travelling as a zero-padded sum (a host that supplies only the all-reduce hook):
        # same numbers moved differently
        ctx.set_allgatherv(None)
        x_sum, it_sum, r_sum = ctx.lsqr_solve_sensit(b, 15, 1e-13, 0.0, 0.0, [np.full(c1 - c0, alpha, np.float32)], [rhs_full[c0:c1]])
        ctx.debug_set("deterministic", 0)
        assert it_sum == it and np.linalg.norm(x_sum - x_loc) <= 1e-12 * np.linalg.norm(x_loc) and abs(r_sum - r) <= 1e-12 * r
        ctx.lsqr_set_wavelet_domain(True)
        # ... against the single-rank oracle: x_spatial solves min |S W x - b|^2 + |alpha x - rhs|^2; W orthonormal, so
        # x_wavelet = W x solves the wavelet-domain system with the transformed right-hand side, iteration by iteration
        Cm = orc.diag_csr(np.full(N, alpha, np.float32))
        rhs_w = orc.wavelet(rhs_full, dims[0], dims[1], dims[2], 1)
        xw_ref, it_ref, r_ref = orc.lsqr(S_full, Cm, N, np.concatenate([b, rhs_w]), 15)
        x_ref = orc.wavelet(xw_ref, dims[0], dims[1], dims[2], 1, inverse=True)
        assert it == it_ref == 15
        err = np.linalg.norm(x_loc - x_ref[c0:c1]) / np.linalg.norm(x_ref)
        assert err <= 1e-6 and abs(r - r_ref) <= 1e-6 * r_ref, (err, r, r_ref)
        ctx.close()
        q.put((rank, "ok"))
    except Exception:      # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_spatial_unknowns_two_ranks():
    """WAVELET_DOMAIN = F on 2 ranks: every product with S gathers the slices, transforms on every rank and keeps its slice."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29900 + os.getpid() % 2000
    procs = [ctxm.Process(target=_worker_spatial, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_one_rank_of_2_4_8_at_the_headline_size_stays_inside_its_memory_plan():
    """What ONE rank of a P-GPU run of BASELINE config 5 (9.96e6 cells x 99 856 data, D4 r = 0.02) holds, built at FULL size on the one
    GPU of this box: its row store (its ndata / P data x all columns, the row-parallel half of the build), then - the row store still
    resident, as during the relayout - the tiles of its column range with their transposed copy, then the solve state.  A sampler
    thread reads the device's used memory every 20 ms; every phase must stay inside tomofast-x_amd/distributed.py::memory_plan (what
    bench.py --gpus P checks before it allocates) and the plan must not be loose by more than a third."""
    import threading
    import time
    import torch
    tfx = importlib.import_module("tomofast-x_amd")
    nx, ny, nz, ox, oy, ctype, rate = 256, 256, 152, 316, 316, 2, 0.02
    N = nx * ny * nz
    xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
    D = xs.size
    ctx = tfx.Context(0)
    hbm = ctx.device_info()["hbm_bytes"]

    def used():
        free_b, total_b = torch.cuda.mem_get_info(0)
        return total_b - free_b

    class Peak:
        def __enter__(self):
            self.peak, self.stop = used(), False
            self.t = threading.Thread(target=self.run, daemon=True)
            self.t.start()
            return self

        def run(self):
            while not self.stop:
                self.peak = max(self.peak, used())
                time.sleep(0.02)

        def __exit__(self, *a):
            self.stop = True
            self.t.join()
            self.peak = max(self.peak, used())

    try:
        base = used()
        ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
        cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
        for P in (8, 4, 2):
            plan = tfx.distributed.memory_plan(N, D, rate, P, hbm_bytes=hbm)
            assert plan["fits"] and plan["adjoint_copy_fits"]
            ph = {k: v * 1e9 for k, v in plan["phases_GB"].items()}
            dstart = tfx.distributed.data_row_partition(D, P)
            r0, r1 = int(dstart[0]), int(dstart[1])
            with Peak() as pk:
                res = ctx.rowstore_build(xs[r0:r1], ys[r0:r1], zs[r0:r1], cw, ctype, rate, 1.0, None, None)
            peak_build = pk.peak - base
            hist = res["nnz_hist"].astype(np.int32)               # this rank's rows: a fair stand-in for the all-rows histogram's balance
            nel, nnz = tfx.sensitivity.get_load_balancing_nelements(hist, P)
            c0, c1 = 0, int(nel[0])
            # (the rank knows from the all-reduced histogram what its columns hold; here: its own rows' count scaled to all rows)
            ctx.matrix_reserve(int(1.03 * P * int(hist[c0:c1].astype(np.int64).sum())) + D)
            with Peak() as pk:
                got = ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, col_range=(c0, c1))
            peak_share = pk.peak - base
            assert ctx.matrix_format()["adjoint_copy"]
            resident = used() - base                              # row store + share + copy
            ctx.rowstore_free()
            ncl = c1 - c0
            ctx.lsqr_begin(np.ones(D), 1e-300, 0.0, 0.0, [np.full(ncl, np.float32(1e-7), np.float32)], [np.zeros(ncl)])
            ctx.lsqr_iterate(2)
            solve = used() - base
            ctx.lsqr_end()
            ctx.matrix_free()
            share_frac = got["nnz"] / (float(D) * int(rate * N))
            print("P = %d, one rank at full size: row store phase %.1f GB (plan %.1f), share + copy built beside it %.1f GB (plan: relayout %.1f, "
                  "adjoint copy %.1f), resident with the row store %.1f GB, solve %.1f GB (plan %.1f); share of the entries %.4f (1 / P = %.4f)" %
                  (P, peak_build / 1e9, ph["build"] / 1e9, peak_share / 1e9, ph["relayout"] / 1e9, ph["adjoint_copy"] / 1e9, resident / 1e9,
                   solve / 1e9, ph["solve"] / 1e9, share_frac, 1.0 / P))
            assert abs(share_frac - 1.0 / P) <= 0.02 / P
            assert peak_build <= ph["build"], (peak_build, ph["build"])
            # this emulation builds the share with the direct kernel build beside the row store, so its work buffers and the scratch of the
            # transposition are alive together (in a P-rank run the relayout's buffers stand in the work buffers' place): the same terms
            b = plan["bytes"]
            together = b["grid"] + b["runtime"] + b["row_store"] + b["share"] + b["copy"] + b["build_work"] + b["copy_scratch"]
            # (+ the 3 % by which this emulation over-reserves its share: it only has its own rows' histogram to scale)
            assert peak_share <= together + 0.03 * (b["share"] + b["copy"]), (peak_share, together)
            assert resident <= ph["adjoint_copy"] and solve <= ph["solve"], (resident, solve, ph)
            assert solve >= 0.66 * ph["solve"] and peak_build >= 0.5 * ph["build"] and peak_share >= 0.66 * together, (solve, peak_build, peak_share, ph)
    finally:
        ctx.close()
