"""The tail launch of an LSQR iteration (csrc/lsqr.hip k_update_xw_next: x / w update + the next iteration's u = -alpha u and constraint
forward step, the default) against the same work as three launches (debug key "lsqr_merge_tail" = 0): identical bits of x, r and the
iteration count for every shape of the system lsqr_solve_sensit takes (lsqr_solver2.F90:47-308) - diagonal blocks with soft thresholding,
general constraint rows, spatial unknowns, the stepping API across chunks, the exit tests inside a queued chunk, a target misfit, the
multi-rank order of operations on a world-size-1 communicator."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_cases  # noqa: E402
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


def _both(ctx, solve):
    """solve() with the three separate launches, then with the merged tail launch; returns the first result."""
    ctx.debug_set("lsqr_merge_tail", 0)
    ref = solve()
    ctx.debug_set("lsqr_merge_tail", 1)
    got = solve()
    assert got[1] == ref[1], (got[1], ref[1])
    assert got[2] == ref[2], (got[2], ref[2])
    assert bits_equal(got[0], ref[0]), float(np.max(np.abs(got[0] - ref[0])))
    return ref


def test_diagonal_blocks_and_soft_threshold_same_bits(ctx):
    g = np.load(os.path.join(GOLDEN, "e2e_d4.npz"))
    N = int(g["nx"]) * int(g["ny"]) * int(g["nz"])
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    ctx.matrix_upload_csr(g["obs"].shape[0], N, *S)
    b = g["np1_data_observed"]
    rng = np.random.default_rng(9)
    d1 = np.full(N, np.float32(1e-7), np.float32)
    d2 = (np.float32(3e-7) * (1 + rng.random(N))).astype(np.float32)
    r1, r2 = rng.standard_normal(N) * 1e-10, rng.standard_normal(N) * 1e-10
    for gamma in (0.0, 1e-6):
        for blocks in (([], []), ([d1], [r1]), ([d1, d2], [r1, r2])):
            x, it, r = _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 40, 1e-13, gamma, 0.0, *blocks))
            assert 0 < it <= 40          # (the undamped system reaches |rhobar| < 1e-30 before 40 iterations - in both forms alike)
    # anchor: the default (merged) solve against the oracle
    x, it, r = ctx.lsqr_solve_sensit(b, 6, 1e-13, 0.0, 0.0, [d1], [r1])
    xo, ito, ro = orc.lsqr(S, orc.diag_csr(d1), N, np.concatenate([b, r1]), 6, 1e-13, 0.0)
    assert it == ito and np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo) and abs(r - ro) <= 1e-9 * ro


def test_many_blocks_per_norm_same_bits(ctx):
    """1.2e6 columns, 3000 rows: the norms have 1024 partial sums each (both forms take their two-launch reductions)."""
    rng = np.random.default_rng(4)
    nrows, ncols, per_row = 3000, 1_200_000, 400
    stride = ncols // per_row                        # one column in each of per_row strides: strictly increasing inside a row
    cols = (np.arange(per_row, dtype=np.int64)[None, :] * stride + rng.integers(0, stride, (nrows, per_row)) + 1).astype(np.int32)   # 1-based
    vals = rng.standard_normal((nrows, per_row)).astype(np.float32)
    rowptr = np.arange(nrows + 1, dtype=np.int64) * per_row
    ctx.matrix_upload_csr(nrows, ncols, rowptr, cols.ravel(), vals.ravel())
    b = rng.standard_normal(nrows)
    d = np.full(ncols, np.float32(0.05), np.float32)
    rhs = rng.standard_normal(ncols) * 1e-3
    x, it, r = _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 30, 1e-13, 0.0, 0.0, [d], [rhs]))
    assert it == 30 and np.isfinite(x).all() and r < 1.0


def test_general_constraint_rows_and_spatial_unknowns_same_bits(ctx):
    g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    nd = S[0].size - 1
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    ctx.matrix_upload_csr(nd, N, *S)
    b = g["np1_data_observed"]
    rng = np.random.default_rng(2)
    # general C: a bidiagonal difference operator (two entries per row), with a right-hand side
    cc = np.stack([np.arange(1, N), np.arange(2, N + 1)], 1).astype(np.int32).ravel()      # 1-based
    cv = np.tile(np.array([-1e-7, 1e-7], np.float32), N - 1)
    ctx.cons_upload_csr(np.arange(N, dtype=np.int64) * 2, cc, cv, rng.standard_normal(N - 1) * 1e-9)
    diag, rhs = [np.full(N, np.float32(1e-7), np.float32)], [rng.standard_normal(N) * 1e-9]
    _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 25, 1e-13, 0.0, 0.0, diag, rhs))
    ctx.lsqr_set_wavelet_domain(False, 1)
    _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 12, 1e-13, 0.0, 0.0, diag, rhs))
    ctx.lsqr_set_wavelet_domain(True)
    ctx.cons_clear()


def test_exit_tests_inside_a_chunk_and_stepping_same_bits(ctx):
    """r <= rmin is met in the middle of a queued chunk of 16 iterations: the iterations behind it are void in both forms; the
    stepping API (begin / iterate(k) / end) continues across chunks; the reference's known-answer systems (tests_lsqr.f90)."""
    for name, c in sorted(kat_cases.cases().items()):
        S = kat_cases.dense_to_csr(c["A"])
        ctx.matrix_upload_csr(c["A"].shape[0], c["A"].shape[1], *S)
        x, it, r = _both(ctx, lambda: ctx.lsqr_solve_sensit(c["b"], c["niter"], c["rmin"]))
        kat_cases.check(c, x)
    g = np.load(os.path.join(GOLDEN, "e2e_d4.npz"))
    N = int(g["nx"]) * int(g["ny"]) * int(g["nz"])
    ctx.matrix_upload_csr(g["obs"].shape[0], N, g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    b = g["np1_data_observed"]
    d1 = [np.full(N, np.float32(1e-7), np.float32)]
    z = [np.zeros(N)]
    x, it, r = _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 200, 1e-3, 0.0, 0.0, d1, z))
    assert 0 < it < 200 and r <= 1e-3

    def stepped():
        ctx.lsqr_begin(b, 1e-13, 0.0, 0.0, d1, z)
        done = 0
        r_ = 1.0
        for k in (1, 5, 16, 18):
            dk, r_ = ctx.lsqr_iterate(k)
            done += dk
        return ctx.lsqr_end(), done, r_
    xs, its, rs = _both(ctx, stepped)
    x1, it1, r1 = ctx.lsqr_solve_sensit(b, 40, 1e-13, 0.0, 0.0, d1, z)
    assert its == it1 and bits_equal(xs, x1) and rs == r1


def test_target_misfit_exit_same_bits(ctx):
    g = np.load(os.path.join(GOLDEN, "e2e_d4.npz"))
    N = int(g["nx"]) * int(g["ny"]) * int(g["nz"])
    nd = g["obs"].shape[0]
    ctx.matrix_upload_csr(nd, N, g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    b = g["np1_data_observed"]
    x0, it0, r0 = ctx.lsqr_solve_sensit(b, 30, 1e-13, 0.0, 0.0)
    rms = np.sqrt(np.sum((ctx.mult_vector(x0) - b) ** 2) / nd)
    x, it, r = _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 200, 1e-13, 0.0, 3.0 * rms))
    assert 0 < it < 30


def test_multi_rank_order_of_operations_same_bits(ctx):
    """World-size-1 communicator with the collectives forced on: the multi-rank iteration (..., ncclAllReduce of |v|^2, alpha + rotation,
    tail launch) against its separate launches and against the single-rank solve."""
    g = np.load(os.path.join(GOLDEN, "e2e_haar.npz"))
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    nd = S[0].size - 1
    ctx.matrix_upload_csr(nd, N, *S)
    b = g["np1_data_observed"]
    diag, rhs = [np.full(N, np.float32(1e-7), np.float32)], [np.random.default_rng(1).standard_normal(N) * 1e-8]
    x0, it0, r0 = ctx.lsqr_solve_sensit(b, 25, 1e-13, 1e-7, 0.0, diag, rhs)
    ctx.comm_init_rccl(ctx.comm_unique_id(), 0, 1)
    ctx.debug_set("force_collectives", 1)
    x1, it1, r1 = _both(ctx, lambda: ctx.lsqr_solve_sensit(b, 25, 1e-13, 1e-7, 0.0, diag, rhs))
    ctx.debug_set("force_collectives", 0)
    ctx.comm_destroy()
    assert it0 == it1 == 25 and bits_equal(x0, x1) and r0 == r1
