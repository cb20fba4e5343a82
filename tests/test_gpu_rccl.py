"""RCCL inside libtfx.so on the one GPU of the test box: a world-size-1 communicator with the collectives forced on
(debug key "force_collectives") runs the real ncclAllReduce / ncclBroadcast calls on the ctx stream.  What this proves: the
library links RCCL, joins a communicator from a unique id, queues its reductions in-stream between the LSQR kernels (no host
hook is set), and the multi-rank code path (separate alpha / rotation launches, gathered slices) gives the bits of the
single-rank path.  What it cannot prove here: several GPUs (the driver's multi-GPU bench does)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


@pytest.fixture()
def ctx():
    c = tfx.Context(0)
    yield c
    c.close()


def _problem():
    g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    return g, dims, S


def test_world_size_1_communicator_runs_lsqr_through_rccl(ctx):
    g, dims, S = _problem()
    N = int(np.prod(dims))
    nd = S[0].size - 1
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    ctx.matrix_upload_csr(nd, N, *S)
    b = g["np1_data_observed"]
    diag, rhs = [np.full(N, np.float32(1e-7), np.float32)], [np.random.default_rng(1).standard_normal(N) * 1e-8]
    ctx.debug_set("deterministic", 1)          # fixed accumulation order in the products: the two runs can be compared bit for bit
    x0, it0, r0 = ctx.lsqr_solve_sensit(b, 25, 1e-13, 0.0, 0.0, diag, rhs)
    d0 = ctx.calc_data(x0, 2.5, np.linspace(0.5, 2.0, nd))
    # the same with a communicator and the multi-rank path forced: reduction 1 (rows + 1 doubles) and reduction 2 (1 double) of
    # every iteration are ncclAllReduce calls on the ctx stream, calc_data's too
    ctx.comm_init_rccl(ctx.comm_unique_id(), 0, 1)
    ctx.debug_set("force_collectives", 1)
    x1, it1, r1 = ctx.lsqr_solve_sensit(b, 25, 1e-13, 0.0, 0.0, diag, rhs)
    d1 = ctx.calc_data(x1, 2.5, np.linspace(0.5, 2.0, nd))
    assert it0 == it1 == 25
    assert bits_equal(x0, x1) and r0 == r1 and bits_equal(d0, d1)
    # spatial unknowns: the slices are gathered with grouped ncclBroadcast (one rank: its own slice), transformed, sliced
    ctx.lsqr_set_wavelet_domain(False, 1)
    ctx.lsqr_set_partition(0, 1)
    xs1, its1, rs1 = ctx.lsqr_solve_sensit(b, 10, 1e-13, 0.0, 0.0, diag, rhs)
    ctx.debug_set("force_collectives", 0)
    xs0, its0, rs0 = ctx.lsqr_solve_sensit(b, 10, 1e-13, 0.0, 0.0, diag, rhs)
    ctx.lsqr_set_wavelet_domain(True)
    assert its0 == its1 == 10 and bits_equal(xs0, xs1) and rs0 == rs1
    ctx.debug_set("deterministic", 0)
    ctx.comm_destroy()
    # and against the oracle, as a sanity anchor of the whole sequence
    xo, ito, ro = orc.lsqr(S, orc.diag_csr(diag[0]), N, np.concatenate([b, rhs[0]]), 25)
    assert np.linalg.norm(x0 - xo) <= 1e-6 * np.linalg.norm(xo)


def test_host_driven_collectives_over_the_communicator(ctx):
    """tfx_comm_allreduce (f64 / i32 / i64), barrier, and the HostComm wrapper the exchange build uses - identity on one rank."""
    import torch
    ctx.comm_init_rccl(ctx.comm_unique_id(), 0, 1)
    comm = tfx.distributed.HostComm(ctx, 0, 1, 0, True)
    comm.nranks = 2                                  # make the wrapper issue the calls (a 1-rank communicator sums to itself)
    for arr in (np.arange(1000, dtype=np.int32), np.arange(7, dtype=np.int64) * 10 ** 12, np.linspace(0, 1, 333)):
        assert np.array_equal(comm.allreduce_host(arr), arr)
    t = torch.arange(64, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.comm_allreduce(t.data_ptr(), 64, "f64")
    ctx.comm_barrier()
    assert torch.equal(t.cpu(), torch.arange(64, dtype=torch.float64))
    with pytest.raises(tfx.TfxError):
        ctx.comm_send(t.data_ptr(), 8, 0)           # no peer on a one-rank communicator
    ctx.comm_destroy()
    with pytest.raises(tfx.TfxError):
        ctx.comm_barrier()                          # no communicator any more


def test_comm_info_and_abort(ctx):
    """tfx_comm_info: what the communicator itself reports (ncclCommCount / UserRank / CuDevice), the RCCL version and the file that
    serves the nccl* symbols; tfx_comm_abort: drops the communicator without a hand-shake (the failure leg of the start-up ladder), and -
    called while no communicator exists - cancels a rendezvous that is still in flight: the next tfx_comm_init_rccl starts clean."""
    info = ctx.comm_info()
    assert info["rccl_ranks"] == 0 and info["rccl_rank"] == -1 and info["rccl_version"] > 20000 and "librccl" in info["librccl"]
    ctx.comm_init_rccl(ctx.comm_unique_id(), 0, 1)
    info = ctx.comm_info()
    assert (info["rccl_ranks"], info["rccl_rank"], info["rccl_device"]) == (1, 0, 0)
    ctx.comm_abort()
    assert ctx.comm_info()["rccl_ranks"] == 0
    with pytest.raises(tfx.TfxError):
        ctx.comm_barrier()
    ctx.comm_abort()                                # nothing to abort: marks any in-flight rendezvous as cancelled ...
    ctx.comm_init_rccl(ctx.comm_unique_id(), 0, 1)  # ... and a fresh one resets the mark
    assert ctx.comm_info()["rccl_ranks"] == 1
    ctx.comm_barrier()
    ctx.comm_destroy()
