"""Parity of the HIP path (libtfx.so through its C ABI) with the CPU oracle and the reference's golden vectors.
Runs on a real MI355X only (`pytest -m gpu`).

Stated tolerances
  wavelets, threshold, compaction on identical inputs ......... bit-exact
  prism rows (device libm atan2/log vs glibc, <= 2 ulp/term) ... 8 ulp of G * sum|terms| (the 24 terms cancel heavily)
  column weights (device pow) ................................... 1e-14 relative
  built matrix vs reference SENSIT rows ......................... same nel per row (+-1 on threshold ties), >= 99.9 %
                                                                  identical sparsity, kept values within 2 fp32 ulp
  S x, S^T y (fp64 accumulation, different summation order) ..... 1e-12 of |S| |x|
  LSQR on the same matrix ....................................... 10 x the measured distance (profiles/r06_parity.json): x rel-L2
                                                                  6e-10, the one mid-convergence iterate of the 40 x 60 toy 6e-5
  end-to-end inversion .......................................... final model rel-L2 1e-6, data cost rel 1e-5
"""
import importlib
import os

import numpy as np
import pytest

import kat_cases
import oracle_lib as orc
import oracle_inversion as oinv
from parity_report import report
import ref_binaries

pytestmark = pytest.mark.gpu

tfx = importlib.import_module("tomofast-x_amd")


@pytest.fixture(scope="module")
def ctx():
    c = tfx.Context(0)
    yield c
    c.close()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def nonuniform(nx, ny, nz, rng, x0=100.0, y0=-50.0, z0=0.0):
    """A tensor-product grid with irregular spacings (cells share their faces bit for bit), i fastest."""
    xe = x0 + np.concatenate([[0], np.cumsum(rng.uniform(20, 90, nx))])
    ye = y0 + np.concatenate([[0], np.cumsum(rng.uniform(20, 90, ny))])
    ze = z0 + np.concatenate([[0], np.cumsum(rng.uniform(10, 60, nz))])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    return [xe[i], xe[i + 1], ye[j], ye[j + 1], ze[k], ze[k + 1]]


# ---------------------------------------------------------------------------------------------------------------
def test_wavelets_bit_exact_vs_reference(ctx, golden_dir):
    g = load(golden_dir, "wavelet")
    keys = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    for k in keys:
        dims, t = k.split("_t")
        n1, n2, n3 = [int(v) for v in dims.split("x")]
        a = g[k + "_in"]
        assert bits_equal(ctx.forward_wavelet(a, n1, n2, n3, int(t)), g[k + "_fwd"]), k
        assert bits_equal(ctx.inverse_wavelet(a, n1, n2, n3, int(t)), g[k + "_inv"]), k


@pytest.mark.parametrize("dims", [(64, 48, 40), (100, 3, 17), (512, 4, 2), (5, 600, 3)])
@pytest.mark.parametrize("wtype", [1, 2])
def test_wavelets_bit_exact_vs_oracle_larger(ctx, dims, wtype):
    rng = np.random.default_rng(11)
    n1, n2, n3 = dims
    a = rng.standard_normal((3, n1 * n2 * n3))
    fw = ctx.forward_wavelet(a, n1, n2, n3, wtype)          # batched: 3 vectors at once
    for v in range(3):
        ref = orc.wavelet(a[v], n1, n2, n3, wtype)
        assert bits_equal(fw[v], ref)
        assert bits_equal(ctx.inverse_wavelet(ref, n1, n2, n3, wtype), orc.wavelet(ref, n1, n2, n3, wtype, inverse=True))


@pytest.mark.parametrize("wtype", [1, 2])
def test_pipelined_wavelet_pass_bit_identical(ctx, wtype):
    """The software-pipelined persistent form of the axis pass (k_wavelet_axis_pipe, debug key "wave_pipe": measured in round 6 and not
    the default - profiles/README.md) runs the same lifting code on the same LDS tile: same bits, forward and inverse, on every axis
    (x: contiguous tiles; y / z: strided 16-byte transfers; a z axis whose length leaves transfers idle)."""
    n1, n2, n3 = 64, 32, 40
    rng = np.random.default_rng(23)
    a = rng.standard_normal((48, n1 * n2 * n3))
    want_f, want_i = ctx.forward_wavelet(a, n1, n2, n3, wtype), ctx.inverse_wavelet(a, n1, n2, n3, wtype)
    assert bits_equal(want_f[0], orc.wavelet(a[0], n1, n2, n3, wtype))
    try:
        for wgs in (1, 3):
            assert ctx.debug_set("wave_pipe", wgs) == wgs
            assert bits_equal(ctx.forward_wavelet(a, n1, n2, n3, wtype), want_f), wgs
            assert bits_equal(ctx.inverse_wavelet(a, n1, n2, n3, wtype), want_i), wgs
    finally:
        ctx.debug_set("wave_pipe", 0)


def test_wavelet_unknown_type(ctx):
    with pytest.raises(ValueError):
        ctx.forward_wavelet(np.zeros(8), 2, 2, 2, 3)


def test_prism_rows_vs_reference(ctx, golden_dir):
    g = load(golden_dir, "prism")
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    rows = ctx.graviprism_z(g["obs"][:, 0], g["obs"][:, 1], g["obs"][:, 2])
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    for o, r, ref in zip(g["obs"], rows, g["rows"]):
        # gz is a sum of 24 terms that cancel heavily; device atan2/log differ from glibc by <= 2 ulp per term, so the
        # error bound is a few ulp of the TERM magnitude, not of the result
        assert np.all(np.abs(r - ref) <= 8 * 2.3e-16 * prism_term_scale(grid, o))


def test_fastmath_device(ctx):
    """The device build of csrc/fastmath.h (log / atan2 of the prism kernels) against the host libm: <= 1 ulp class + the host's own."""
    rng = np.random.default_rng(5)
    n = 400000
    XX = 3e4 * rng.uniform(-1, 1, n)
    YY = np.where(rng.uniform(size=n) < 0.3, 10.0 ** rng.uniform(-6, 6, n), 3e4 * rng.uniform(-1, 1, n))
    ZZ = 1e4 * rng.uniform(0, 1, n) + 0.1
    R = np.sqrt(XX * XX + YY * YY + ZZ * ZZ)
    spacing = lambda v: np.spacing(np.maximum(np.abs(v), 2.3e-308))
    # log(R + XX), atan2(XX YY, ZZ R): the arguments of gravity_field.f90:165-181
    a = R + XX
    keep = a > 0
    lg, _ = ctx.fastmath_eval(a[keep], a[keep])
    ref = np.log(a[keep])
    assert np.all(np.abs(lg - ref) <= 1.6 * np.maximum(spacing(ref), 2.3e-16))
    y, x = XX * YY, ZZ * R * np.where(rng.uniform(size=n) < 0.2, -1.0, 1.0)
    _, at = ctx.fastmath_eval(y, x)
    ref = np.arctan2(y, x)
    assert np.all(np.abs(at - ref) <= 2.7 * spacing(ref))
    assert np.all(np.abs(at - ref) <= 2.0 * np.maximum(spacing(ref), 1.2e-16))
    # special values take the library path
    sa = np.array([0.0, -0.0, 1.0, -1.0, 1.0, 0.0, np.inf, 1e-320, 5.0, 1e300, -3.0])
    sb = np.array([1.0, -1.0, 0.0, -0.0, 1.0, 0.0, 1.0, 1.0, -np.inf, 1e-300, -3.0])
    lg, at = ctx.fastmath_eval(sa, sb)
    with np.errstate(all="ignore"):
        rl, ra = np.log(sa), np.arctan2(sa, sb)
    assert np.all((lg == rl) | (np.isnan(lg) & np.isnan(rl)) | (np.abs(lg - rl) <= 2 * spacing(rl)))
    assert np.all((np.abs(at - ra) <= 2 * spacing(ra)) & (np.signbit(at) == np.signbit(ra)))


def prism_term_scale(grid, o):
    X1, X2, Y1, Y2, Z1, Z2 = grid
    s = np.zeros(X1.size)
    for xx in (o[0] - X1, o[0] - X2):
        for yy in (o[1] - Y1, o[1] - Y2):
            for zz in (o[2] - Z1, o[2] - Z2):
                R = np.sqrt(xx * xx + yy * yy + zz * zz)
                s += np.abs(zz) * 2 * np.pi + np.abs(xx * np.log(R + yy)) + np.abs(yy * np.log(R + xx))
    return 6.674e-11 * s



def test_prism_tensor_path_bit_identical_to_general(ctx, golden_dir):
    """Tensor-product grids take the shared-node kernel; it must give the same bits as the six-array kernel."""
    g = load(golden_dir, "prism")
    cases = [((int(g["nx"]), int(g["ny"]), int(g["nz"])), [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")], g["obs"])]
    xs, ys, zs = tfx.synthetic.observations(70, 37, 3, 2)
    cases.append(((70, 37, 19), list(tfx.synthetic.grid(70, 37, 19)), np.stack([xs, ys, zs], 1)))
    for dims, grid, obs in cases:
        ctx.set_grid(*dims, *grid)
        assert ctx.debug_set("tensor_grid") == 1
        rows_t = ctx.graviprism_z(obs[:, 0], obs[:, 1], obs[:, 2])
        ctx.debug_set("force_general_prism", 1)
        assert ctx.debug_set("tensor_grid") == 0
        rows_g = ctx.graviprism_z(obs[:, 0], obs[:, 1], obs[:, 2])
        ctx.debug_set("force_general_prism", 0)
        assert bits_equal(rows_t, rows_g)
    # a grid whose cells do not share faces is not a tensor grid: general kernel
    dims, grid, obs = cases[0]
    bent = [a.copy() for a in grid]
    bent[1][7] += 1e-9
    ctx.set_grid(*dims, *bent)
    assert ctx.debug_set("tensor_grid") == 0
    ierr, ref = orc.graviprism_z(bent, *obs[0])
    r = ctx.graviprism_z(obs[:1, 0], obs[:1, 1], obs[:1, 2])[0]
    assert np.all(np.abs(r - ref) <= 8 * 2.3e-16 * prism_term_scale(bent, obs[0]))


def test_magprism_tensor_path_bit_identical_to_general(ctx, golden_dir):
    """Magnetic rows on a tensor-product grid (atan2 terms and corner distances shared through LDS) vs the six-array kernel:
    same bits for every component combination, observations above, beside and INSIDE cells."""
    g = load(golden_dir, "magprism")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    cases = [((int(g["nx"]), int(g["ny"]), int(g["nz"])), grid, g["obs"])]
    xs, ys, zs = tfx.synthetic.observations(37, 21, 3, 2)
    obs2 = np.stack([xs, ys, zs], 1)
    obs2 = np.concatenate([obs2, [[1234.5, 987.6, 433.3]]])                    # inside a cell of the uniform 100 m grid
    cases.append(((37, 21, 19), list(tfx.synthetic.grid(37, 21, 19)), obs2))
    field = (-62.0, 11.0, 20.0, 57000.0)
    for dims, gr, obs in cases:
        ctx.set_grid(*dims, *gr)
        assert ctx.debug_set("tensor_grid") == 1
        for ncm, ncd in ((1, 1), (1, 3), (3, 1), (3, 3)):
            rows_t = ctx.sensit_lines(2, obs[:, 0], obs[:, 1], obs[:, 2], ndata_components=ncd, nmodel_components=ncm, mag_field=field)
            ctx.debug_set("force_general_prism", 1)
            rows_g = ctx.sensit_lines(2, obs[:, 0], obs[:, 1], obs[:, 2], ndata_components=ncd, nmodel_components=ncm, mag_field=field)
            ctx.debug_set("force_general_prism", 0)
            assert bits_equal(rows_t, rows_g), (dims, ncm, ncd)
        # and the whole build (cost_full partials differ in order only: same kept entries)
        cw = orc.column_weight_type1(gr, 3.0, 0.0, 1.0)
        res_t = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, 0.2, mag_field=field)
        A = ctx.matrix_download_csr()
        ctx.debug_set("force_general_prism", 1)
        res_g = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, 0.2, mag_field=field)
        B = ctx.matrix_download_csr()
        ctx.debug_set("force_general_prism", 0)
        assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and A[2].tobytes() == B[2].tobytes()
        assert abs(res_t["comp_error"] - res_g["comp_error"]) <= 1e-12 * res_g["comp_error"]


def test_gradiprism_tensor_path_bit_identical_to_general(ctx, golden_dir):
    """Gzz / full gradient tensor on a tensor-product grid (corner terms shared through LDS) vs the six-array kernel."""
    g = load(golden_dir, "gradprism")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    cases = [((int(g["nx"]), int(g["ny"]), int(g["nz"])), grid, g["obs"])]
    xs, ys, zs = tfx.synthetic.observations(70, 37, 3, 2)
    cases.append(((70, 37, 19), list(tfx.synthetic.grid(70, 37, 19)), np.stack([xs, ys, zs], 1)))
    for dims, gr, obs in cases:
        ctx.set_grid(*dims, *gr)
        assert ctx.debug_set("tensor_grid") == 1
        for ncd in (1, 6):
            rows_t = ctx.sensit_lines(1, obs[:, 0], obs[:, 1], obs[:, 2], data_type=2, ndata_components=ncd)
            ctx.debug_set("force_general_prism", 1)
            rows_g = ctx.sensit_lines(1, obs[:, 0], obs[:, 1], obs[:, 2], data_type=2, ndata_components=ncd)
            ctx.debug_set("force_general_prism", 0)
            assert bits_equal(rows_t, rows_g), (dims, ncd)


def mag_term_scale(grid, o, inten):
    """|intensity| / 4 pi times the sum of |atan2| and |log| terms (each O(1..pi)): the tensor entries cancel to O((h/R)^3)."""
    return abs(inten) / (4 * np.pi) * 60.0 * np.ones(grid[0].size)


def test_magprism_rows_vs_reference(ctx, golden_dir):
    """magprism (TMI, scalar model): observations above, beside and INSIDE cells (6-sub-box split), three field directions."""
    g = load(golden_dir, "magprism")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *grid)
    for fi, field in enumerate(g["fields"]):
        rows = ctx.magprism(g["obs"][:, 0], g["obs"][:, 1], g["obs"][:, 2], field)
        for o, r, ref in zip(g["obs"], rows, g["rows_%d" % fi]):
            # 36 transcendental terms of magnitude <= pi cancel; device libm differs by <= 2 ulp per term
            assert np.all(np.abs(r - ref) <= 8 * 2.3e-16 * mag_term_scale(grid, o, field[3])), (fi, o)
            assert np.max(np.abs(r - ref)) <= 1e-9 * np.max(np.abs(ref))


def test_magprism_boundary_error(ctx):
    one = [np.array([v], np.float64) for v in (0.0, 1.0, 0.0, 1.0, 0.0, 1.0)]
    ctx.set_grid(1, 1, 1, *one)
    with pytest.raises(tfx.TfxError) as e:
        ctx.magprism([1.0], [0.5], [-1.0], (90.0, 0.0, 0.0, 5e4))
    assert e.value.code == -3 and "X-boundary" in str(e.value)


def test_build_mag_kernel_vs_oracle(ctx):
    """Magnetic sensitivity kernel through the same compress / tile pipeline (problem_type 2) vs the oracle's rows."""
    nx, ny, nz, ox, oy = 24, 20, 10, 4, 3
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
    field = (-62.0, 11.0, 0.0, 57000.0)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = orc.column_weight_type1(grid, 3.0, 0.0, 1.0)
    res = ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.2, mag_field=field)
    rp, cols, vals = ctx.matrix_download_csr()
    N = nx * ny * nz
    K = int(0.2 * N)
    magv = orc.dircos(*field[:3])
    tot = 0
    for r in range(xs.size):
        ierr, row = orc.magprism_tmi(grid, xs[r], ys[r], zs[r], magv, field[3])
        assert ierr == 0
        w = orc.wavelet(row * cw, nx, ny, nz, 1)
        c_ref, v_ref, _, _ = orc.compress_row(w, K)
        cb, vb = cols[rp[r]:rp[r + 1]], vals[rp[r]:rp[r + 1]]
        common, ib, ir = np.intersect1d(cb, c_ref, return_indices=True)
        assert common.size >= 0.995 * c_ref.size and abs(cb.size - c_ref.size) <= 4
        # values: relative to the row scale (small coefficients carry the cancellation error of the tensor entries)
        assert np.max(np.abs(vb[ib].astype(np.float64) - v_ref[ir].astype(np.float64))) <= 1e-9 * np.max(np.abs(v_ref))
        tot += cb.size
    assert tot == res["nnz"]


def test_magnetic_end_to_end_vs_reference(ctx, golden_dir):
    """Magnetic inversion (problem 2) built and solved on the GPU vs the reference's SENSIT rows and final model."""
    g = load(golden_dir, "e2e_mag")
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = ctx.calculate_depth_weight(3.0, 0.0, 1.0)
    assert np.max(np.abs(cw - g["column_weight"]) / g["column_weight"]) <= 1e-14
    obs = g["obs"]
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, float(g["rate"]), mag_field=g["field"])
    built = ctx.matrix_download_csr()
    nd = obs.shape[0]
    same = tot = 0
    for r in range(nd):
        cb = built[1][built[0][r]:built[0][r + 1]]
        cr = g["cols"][g["row_ptr"][r]:g["row_ptr"][r + 1]]
        same += np.intersect1d(cb, cr).size
        tot += max(cb.size, cr.size)
    assert same >= 0.995 * tot
    assert abs(res["comp_error"] - float(g["comp_error"])) <= 1e-6 * float(g["comp_error"])
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, 1, g["data_observed"], int(g["nmajor"]), int(g["nminor"]),
                                                     alpha=float(g["alpha"]))
    ref = g["model_final"]
    assert np.linalg.norm(m - ref) <= 1e-5 * np.linalg.norm(ref)


def test_prism_geometry_error(ctx):
    one = [np.array([v], np.float64) for v in (0.0, 1.0, 0.0, 1.0, 0.0, 1.0)]
    ctx.set_grid(1, 1, 1, *one)
    with pytest.raises(tfx.TfxError) as e:
        ctx.graviprism_z([-1.0], [0.0], [0.0])
    assert e.value.code == -3 and "coincides with model grid boundary" in str(e.value)


def test_column_weight_vs_reference(ctx, golden_dir):
    g = load(golden_dir, "e2e_haar")
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
    assert np.max(np.abs(cw - g["np1_column_weight"]) / g["np1_column_weight"]) <= 1e-14


def test_mindist_weight_type3_vs_reference(ctx, golden_dir):
    """forward.depthWeighting.type = 3 (weights_gravmag.f90:140-162) against the compiled reference's weight file, the oracle, and the
    inversion that uses it end to end."""
    g = load(golden_dir, "e2e_dw3")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *grid)
    obs = g["obs"]
    cw = ctx.calculate_mindist_weight(obs[:, 0], obs[:, 1], obs[:, 2], 2.0, 4.0e3)
    assert np.max(np.abs(cw - g["np1_column_weight"]) / g["np1_column_weight"]) <= 1e-14
    for power, mult in ((3.0, 1.0), (2.5, 7.0)):              # x*x*x and the general pow path
        a, b = ctx.calculate_mindist_weight(obs[:, 0], obs[:, 1], obs[:, 2], power, mult), orc.column_weight_type3(grid, obs, power, mult)
        assert np.allclose(a, b, rtol=1e-13, atol=0)
    ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]))
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]))
    ref = g["np1_model_final"]
    assert np.linalg.norm(m - ref) <= 1e-6 * np.linalg.norm(ref)


def test_distance_weight_type2_vs_reference(ctx, golden_dir):
    g = load(golden_dir, "e2e_dw2")
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    obs = g["obs"]
    cw = ctx.calculate_distance_weight(obs[:, 0], obs[:, 1], obs[:, 2], 2.0, 1.0, 4.0e3)
    assert np.max(np.abs(cw - g["np1_column_weight"]) / g["np1_column_weight"]) <= 1e-13
    # and the inversion that uses it, end to end against the reference's files
    ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]))
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]))
    ref = g["np1_model_final"]
    assert np.linalg.norm(m - ref) <= 1e-6 * np.linalg.norm(ref)


@pytest.mark.parametrize("rate", [0.0, 0.01, 0.1, 0.15, 0.5, 1.0])
def test_threshold_and_compaction_bit_exact(ctx, golden_dir, rate):
    g = load(golden_dir, "wavelet")
    for key in ("16x16x8_t1_fwd", "13x7x33_t2_fwd", "10x11x12_t1_fwd"):
        row = g[key].copy()
        row[5] = 0.0                      # exact zeros are always dropped (threshold floor 1e-30)
        row[7] = row[9]                   # a tie
        N = row.size
        K = int(rate * N)
        c_ref, v_ref, thr_ref, cd_ref = orc.compress_row(row, K)
        c, v, thr, cd = ctx.compress_row(row, K)
        assert thr == thr_ref
        assert np.array_equal(c, c_ref) and bits_equal(v, v_ref)
        assert abs(cd - cd_ref) <= 1e-13 * max(cd_ref, 1e-300)


def test_threshold_degenerate_rows(ctx):
    N = 5000
    for row in (np.zeros(N), np.full(N, 3.5), np.concatenate([np.zeros(N - 3), [1e-31, -2.0, 7.0]])):
        for K in (0, 1, 2, N // 2, N):
            c_ref, v_ref, thr_ref, _ = orc.compress_row(row, K)
            c, v, thr, _ = ctx.compress_row(row, K)
            assert thr == thr_ref and np.array_equal(c, c_ref) and bits_equal(v, v_ref), (K,)


# ---------------------------------------------------------------------------------------------------------------
def random_csr(rng, nrows, ncols, mean_nnz, empty_frac=0.1, clustered=False):
    rp = [0]
    cols, vals = [], []
    for r in range(nrows):
        if rng.random() < empty_frac:
            rp.append(rp[-1])
            continue
        n = max(1, int(rng.poisson(mean_nnz)))
        n = min(n, ncols)
        if clustered:
            centre = rng.integers(0, ncols)
            c = np.unique(np.clip((centre + rng.standard_normal(n) * mean_nnz * 2).astype(np.int64), 0, ncols - 1))
        else:
            c = np.unique(rng.integers(0, ncols, n))
        cols.append((c + 1).astype(np.int32))
        vals.append((rng.standard_normal(c.size) * np.exp(rng.normal(0, 3, c.size))).astype(np.float32))
        rp.append(rp[-1] + c.size)
    return np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals)


CSR_SHAPES = [
    (7, 9, 4, False),              # tiny, single tile
    (300, 20000, 50, False),       # 2 column tiles
    (2500, 40000, 30, False),      # 2 row blocks x 3 column tiles, short segments
    (2500, 40000, 400, True),      # clustered: long dense segments + many empty (row, tile) segments
    (64, 70000, 6000, False),      # long rows
    (4100, 1000, 3, False),        # 3 row blocks, narrow
]


@pytest.mark.parametrize("shape", CSR_SHAPES)
def test_matrix_roundtrip_and_products(ctx, shape):
    nrows, ncols, mean_nnz, clustered = shape
    rng = np.random.default_rng(nrows * 7 + ncols)
    S = random_csr(rng, nrows, ncols, mean_nnz, clustered=clustered)
    ctx.matrix_upload_csr(nrows, ncols, *S)
    info = ctx.matrix_info()
    assert (info["nrows"], info["ncols"], info["nnz"]) == (nrows, ncols, int(S[0][-1]))
    rp, c, v = ctx.matrix_download_csr()
    assert np.array_equal(rp, S[0]) and np.array_equal(c, S[1]) and bits_equal(v, S[2])
    x = rng.standard_normal(ncols)
    y = rng.standard_normal(nrows)
    absS = (S[0], S[1], np.abs(S[2]))
    b = ctx.mult_vector(x)
    ref = orc.spmv(*S, x)
    assert np.all(np.abs(b - ref) <= 1e-12 * orc.spmv(*absS, np.abs(x)) + 1e-300)
    bt = ctx.trans_mult_vector(y)
    reft = orc.spmtv(*S, y, ncols)
    # (without a transposed copy - TFX_ADJ_COPY=0 - the adjoint rounds every product to a fixed grid: absolute, not per column)
    slack = 0.0 if ctx.debug_set("has_adj_copy") else _fixed_point_slack(S, y, ncols)
    assert np.all(np.abs(bt - reft) <= 1e-12 * orc.spmtv(*absS, np.abs(y), ncols) + 1e-300 + slack)
    # add_mult_vector / add_trans_mult_vector
    b0 = rng.standard_normal(nrows)
    assert np.allclose(ctx.mult_vector(x, b0), b0 + ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    t0 = rng.standard_normal(ncols)
    assert np.all(np.abs(ctx.trans_mult_vector(y, t0) - (t0 + reft)) <= 1e-12 * np.abs(t0 + reft) + 1e-12 * np.abs(reft).max() + slack)
    # adjoint identity <S x, y> = <x, S^T y>
    assert abs(np.dot(b, y) - np.dot(x, bt)) <= 1e-11 * np.dot(orc.spmv(*absS, np.abs(x)), np.abs(y))


@pytest.mark.parametrize("shape", [(1, 1), (1, 4096), (1, 4097), (1, 16385), (2048, 3), (2049, 4097), (3, 70000), (8193, 5000)])
def test_matrix_edge_shapes(ctx, shape):
    """Tile-boundary sizes (row block 2048, column tile 4096, forward super block 4 x 2048 rows), single rows / columns, dense
    and nearly empty patterns."""
    nrows, ncols = shape
    rng = np.random.default_rng(nrows * 131 + ncols)
    for pattern in ("dense_row0", "last_col_only", "diag", "empty"):
        rp, cols, vals = [0], [], []
        for r in range(nrows):
            if pattern == "dense_row0":
                c = np.arange(ncols) if r == 0 else np.zeros(0, np.int64)
            elif pattern == "last_col_only":
                c = np.array([ncols - 1])
            elif pattern == "diag":
                c = np.array([r % ncols])
            else:
                c = np.zeros(0, np.int64)
            cols.append((c + 1).astype(np.int32))
            vals.append(rng.standard_normal(c.size).astype(np.float32) + np.float32(3.0))
            rp.append(rp[-1] + c.size)
        S = (np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals))
        ctx.matrix_upload_csr(nrows, ncols, *S)
        back = ctx.matrix_download_csr()
        assert np.array_equal(back[0], S[0]) and np.array_equal(back[1], S[1]) and bits_equal(back[2], S[2]), (shape, pattern)
        x, y = rng.standard_normal(ncols), rng.standard_normal(nrows)
        assert np.allclose(ctx.mult_vector(x), orc.spmv(*S, x), rtol=1e-12, atol=1e-12)
        assert np.allclose(ctx.trans_mult_vector(y), orc.spmtv(*S, y, ncols), rtol=1e-12, atol=1e-12)


def test_matrix_with_more_row_markers_than_entries(ctx):
    """Only the first and the last row of a 2048-row block hold entries, in every column tile: each tile carries 2046 marker
    entries for the empty rows in between - far more than its real entries (found by tools/fuzz_matrix.py: the capacity estimate
    had bounded the markers by the entry count)."""
    nr, nc = 2048, 300000
    cols_row = (np.arange(0, nc, 4099) + 1).astype(np.int32)
    rowptr = np.zeros(nr + 1, np.int64)
    rowptr[1:] = cols_row.size
    rowptr[nr] = 2 * cols_row.size
    cols = np.concatenate([cols_row, cols_row])
    vals = np.random.default_rng(2).standard_normal(cols.size).astype(np.float32)
    ctx.matrix_upload_csr(nr, nc, rowptr, cols, vals)
    back = ctx.matrix_download_csr()
    assert np.array_equal(back[0], rowptr) and np.array_equal(back[1], cols) and bits_equal(back[2], vals)
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal(nc), rng.standard_normal(nr)
    assert np.allclose(ctx.mult_vector(x), orc.spmv(rowptr, cols, vals, x), rtol=1e-13, atol=1e-13)
    assert np.allclose(ctx.trans_mult_vector(y), orc.spmtv(rowptr, cols, vals, y, nc), rtol=1e-13, atol=1e-13)


def _random_csr(rng, nrows, ncols, per_row):
    rp, cols, vals = [0], [], []
    for r in range(nrows):
        k = int(rng.integers(0, per_row + 1))
        c = np.sort(rng.choice(ncols, size=min(k, ncols), replace=False))
        cols.append((c + 1).astype(np.int32))
        vals.append((rng.standard_normal(c.size) * 10.0 ** rng.uniform(-3, 3, c.size)).astype(np.float32))
        rp.append(rp[-1] + c.size)
    return np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals)


def test_normalize_columns_many_rows_and_non_finite_entries(ctx):
    """t_sparse_matrix%normalize_columns has no row limit and propagates NaN / Inf (sparse_matrix.f90:414-443): a matrix with
    more than 2^17 rows (three data components at ~1e5 data, a joint system) - the exact integer column sums coarsen their grid by one
    bit per doubling instead of refusing; a column holding a NaN gets a NaN norm and NaN entries, a column whose square overflows fp32
    gets an infinite norm and zero entries, exactly what the reference's fp32 squares / fp64 sum / division give."""
    rng = np.random.default_rng(5)
    nrows, ncols = (1 << 18) + 37, 300
    per = 2
    cols = (rng.integers(0, ncols, (nrows, per)) + 1).astype(np.int32)
    cols.sort(axis=1)
    cols[:, 1] = np.where(cols[:, 1] == cols[:, 0], cols[:, 1] % ncols + 1, cols[:, 1])
    cols.sort(axis=1)
    assert np.all(cols[:, 1] > cols[:, 0])
    vals = (rng.standard_normal((nrows, per)) * 10.0 ** rng.uniform(-4, 4, (nrows, per))).astype(np.float32)
    rp = np.arange(0, per * nrows + 1, per, dtype=np.int64)
    cols, vals = cols.ravel(), vals.ravel()
    norm_o, vals_o = orc.normalize_columns(rp, cols, vals, ncols)
    ctx.matrix_upload_csr(nrows, ncols, rp, cols, vals)
    norm_g = ctx.normalize_columns()
    vals_g = ctx.matrix_download_csr()[2]
    assert np.all(np.abs(norm_g - norm_o) <= nrows * 2.0 ** -53 * norm_o)
    same = (norm_g == norm_o)[cols - 1]
    assert bits_equal(vals_g[same], vals_o[same]) and np.all(np.abs(vals_g.astype(np.float64) - vals_o) <= 2.0 ** -23 * np.abs(vals_o))
    report("normalize_columns_many_rows", rows=nrows, worst_rel_distance=float(np.max(np.abs(norm_g - norm_o) / norm_o)),
           norms_identical=int(np.count_nonzero(norm_g == norm_o)), norms=ncols)
    # non-finite entries: column 3 holds a NaN, column 7 a value whose square overflows fp32, column 11 an infinity
    nrows, ncols = 400, 16
    A = rng.standard_normal((nrows, ncols)).astype(np.float32)
    A[17, 2] = np.nan
    A[250, 6] = np.float32(3.0e19)
    A[99, 10] = np.inf
    S = kat_cases.dense_to_csr(A.astype(np.float64))
    for adj in (1, 0):
        ctx.debug_set("adj_copy", adj)
        try:
            ctx.matrix_upload_csr(nrows, ncols, S[0], S[1], np.asarray(S[2], np.float32))
            with np.errstate(all="ignore"):
                norm_o, vals_o = orc.normalize_columns(S[0], S[1], np.asarray(S[2], np.float32), ncols)
            norm_g = ctx.normalize_columns()
            vals_g = ctx.matrix_download_csr()[2]
        finally:
            ctx.debug_set("adj_copy", 2)
        assert np.isnan(norm_g[2]) and np.isnan(norm_o[2]) and np.isposinf(norm_g[6]) and np.isposinf(norm_o[6]) and np.isposinf(norm_g[10])
        fin = np.isfinite(norm_o)
        assert np.all(np.abs(norm_g[fin] - norm_o[fin]) <= nrows * 2.0 ** -53 * norm_o[fin])
        c0 = S[1] - 1
        assert np.all(np.isnan(vals_g[c0 == 2])) and np.all(np.isnan(vals_o[c0 == 2]))
        assert np.all(vals_g[c0 == 6] == 0.0) and np.all(vals_o[c0 == 6] == 0.0)
        inf_col = vals_g[c0 == 10]
        assert np.count_nonzero(np.isnan(inf_col)) == 1 and np.count_nonzero(inf_col == 0.0) == inf_col.size - 1      # inf / inf, x / inf
        with np.errstate(all="ignore"):
            assert np.array_equal(np.isnan(vals_g), np.isnan(vals_o))
    ctx.matrix_free()


@pytest.mark.parametrize("group", [0, 1, 2, 4])
def test_forward_super_blocks_give_the_same_product(ctx, group):
    """The forward product shares one staged x tile between `fwd_group` row blocks; every grouping must give the oracle's product
    (10 000 rows = 5 row blocks: super blocks of 1, 2 (ragged last group) and 4 (ragged) blocks)."""
    rng = np.random.default_rng(77)
    nrows, ncols = 10000, 9000
    S = _random_csr(rng, nrows, ncols, 40)
    ctx.debug_set("fwd_group", group)
    try:
        ctx.matrix_upload_csr(nrows, ncols, *S)
        x, y = rng.standard_normal(ncols), rng.standard_normal(nrows)
        ref = orc.spmv(*S, x)
        assert np.allclose(ctx.mult_vector(x), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        reft = orc.spmtv(*S, y, ncols)
        assert np.allclose(ctx.trans_mult_vector(y), reft, rtol=1e-12, atol=1e-12 * np.abs(reft).max())
    finally:
        ctx.debug_set("fwd_group", 0)


def _fixed_point_slack(S, y, ncols):
    """The adjoint kernel on the tiles of S (no transposed copy) rounds every product value * u to a grid of 2^-60 .. 2^-61 of
    (the largest column sum of |value| of its tile group x max|u| of the group's rows) and adds the rounded products exactly
    (matrix.hip k_spmv_adj).  With at most 1024 entries per column and tile group that is, per column, at most
    (entries of the column) * 2^-50 * max|S| * max|u| away from the exact sum - usually 2^-56 or better."""
    per_col = np.bincount(S[1] - 1, minlength=ncols)       # (1-based columns, like the reference's CSR)
    return per_col * 2.0 ** -50 * float(np.abs(S[2]).max(initial=0.0)) * float(np.abs(y).max(initial=0.0))


@pytest.mark.parametrize("shape", [(3000, 20000, 300), (6000, 50000, 2000), (300, 5000, 4000)])
def test_products_are_bit_reproducible(ctx, shape):
    """The production kernels give the same bits on every run, like the reference's sequential sums on a fixed rank count
    (sparse_matrix.f90:316-329, :391-405): the forward kernel (S x, and S^T u on the transposed copy) adds in an order fixed by the
    matrix, the adjoint kernel without a copy accumulates exactly in integers.  Sizes with several chunks per wave and rows that
    straddle the waves' chunk runs (long rows: 4000 entries = 8 chunks per row)."""
    nrows, ncols, per_row = shape
    rng = np.random.default_rng(5 + nrows)
    S = _random_csr(rng, nrows, ncols, per_row)
    x, y = rng.standard_normal(ncols), rng.standard_normal(nrows)
    ref, reft = orc.spmv(*S, x), orc.spmtv(*S, y, ncols)
    absS = (S[0], S[1], np.abs(S[2]))
    tol_f = 1e-12 * orc.spmv(*absS, np.abs(x)) + 1e-300
    tol_a = 1e-12 * orc.spmtv(*absS, np.abs(y), ncols) + 1e-300
    try:
        for mode in (2, 0):
            ctx.debug_set("adj_copy", mode)
            ctx.matrix_upload_csr(nrows, ncols, *S)
            assert ctx.debug_set("has_adj_copy") == (1 if mode else 0)
            f = [ctx.mult_vector(x) for _ in range(4)]
            a = [ctx.trans_mult_vector(y) for _ in range(4)]
            for k in range(1, 4):
                assert bits_equal(f[k], f[0]) and bits_equal(a[k], a[0])
            assert np.all(np.abs(f[0] - ref) <= tol_f)
            assert np.all(np.abs(a[0] - reft) <= tol_a + (0.0 if mode else _fixed_point_slack(S, y, ncols)))
            # a second context on the same matrix: same bits again (nothing depends on allocation addresses or launch history)
            if mode == 2:
                other = tfx.Context(0)
                try:
                    other.debug_set("adj_copy", mode)
                    other.matrix_upload_csr(nrows, ncols, *S)
                    assert bits_equal(other.mult_vector(x), f[0]) and bits_equal(other.trans_mult_vector(y), a[0])
                finally:
                    other.close()
    finally:
        ctx.debug_set("adj_copy", 2)


def test_adjoint_without_a_copy_is_exact_in_integers(ctx):
    """k_spmv_adj: values and u spanning many binades, columns whose products are far below the group's largest one (they keep the
    absolute grid, not a relative one), u = 0, huge and tiny scales, and a NaN / inf in u poisoning the columns it touches."""
    rng = np.random.default_rng(77)
    nrows, ncols = 2500, 9000
    S = list(_random_csr(rng, nrows, ncols, 200))
    S[2] = (S[2] * np.exp(rng.uniform(-12, 12, S[2].size))).astype(np.float32)
    ctx.debug_set("adj_copy", 0)
    try:
        ctx.matrix_upload_csr(nrows, ncols, *S)
        for scale in (1.0, 1e-150, 1e150):
            y = rng.standard_normal(nrows) * np.exp(rng.uniform(-6, 6, nrows)) * scale
            got = ctx.trans_mult_vector(y)
            want = orc.spmtv(*S, y, ncols)
            slack = _fixed_point_slack(S, y, ncols)
            assert np.all(np.abs(got - want) <= 1e-12 * orc.spmtv(S[0], S[1], np.abs(S[2]), np.abs(y), ncols) + slack)
            assert np.linalg.norm(got - want) <= 1e-12 * np.linalg.norm(want)
            assert bits_equal(got, ctx.trans_mult_vector(y))
        assert not np.any(ctx.trans_mult_vector(np.zeros(nrows)))
        t0 = rng.standard_normal(ncols)
        assert bits_equal(ctx.trans_mult_vector(np.zeros(nrows), t0), t0)
        y = rng.standard_normal(nrows)
        for bad in (np.nan, np.inf):
            yb = y.copy()
            yb[1234] = bad
            got = ctx.trans_mult_vector(yb)
            touched = np.zeros(ncols, bool)
            touched[S[1][S[0][1234]:S[0][1235]] - 1] = True
            assert not np.any(np.isfinite(got[touched]))
    finally:
        ctx.debug_set("adj_copy", 2)


def test_matrix_scale_rows_is_the_reload_scaling(ctx):
    """tfx_matrix_scale_rows: value * float32(scale[row]) in fp32, bit for bit what read_sensitivity_kernel does on reload
    (sensitivity_gravmag.F90:834-843); tiled and dense storage."""
    rng = np.random.default_rng(9)
    nrows, ncols = 2500, 9000
    rp, cols, vals = _random_csr(rng, nrows, ncols, 60)
    scale = rng.uniform(0.1, 30.0, nrows)
    ctx.matrix_upload_csr(nrows, ncols, rp, cols, vals)
    ctx.matrix_scale_rows(scale)
    back = ctx.matrix_download_csr()
    want = vals * np.repeat(scale.astype(np.float32), np.diff(rp))
    assert np.array_equal(back[0], rp) and np.array_equal(back[1], cols) and bits_equal(back[2], want.astype(np.float32))


@pytest.mark.parametrize("adj_copy", [0, 1])
def test_normalize_columns_vs_reference(ctx, adj_copy):
    """tfx_matrix_normalize_columns = t_sparse_matrix%normalize_columns (sparse_matrix.f90:414-443): the reference's own unit test
    (tests_sparse_matrix.f90:39-104) and a random kernel with empty columns and values over 12 decades; norms within n * 2^-53 of the
    oracle's sequential sums (the sums here are exact), the fp32 quotients identical except where the norms differ in the last bit,
    the transposed copy normalised with the same bits, two runs identical to the bit; dense storage: bit-identical to the oracle."""
    ctx.debug_set("adj_copy", adj_copy)
    try:
        # the reference's unit test
        A = np.zeros((30, 10))
        A[:, :5] = (np.arange(300).reshape(30, 10) + 1.0)[:, :5]
        S = kat_cases.dense_to_csr(A)
        ctx.matrix_upload_csr(30, 10, *S)
        norm = ctx.normalize_columns()
        assert np.allclose(norm, np.linalg.norm(A, axis=0), rtol=1e-15, atol=0) and np.all(norm[5:] == 0.0)
        for i in range(10):
            col = ctx.mult_vector(np.eye(10)[i])
            assert abs(np.linalg.norm(col) - (1.0 if i < 5 else 0.0)) <= kat_cases.TOL
        # a random kernel
        rng = np.random.default_rng(77)
        nrows, ncols = 3000, 20000
        rp, cols, vals = _random_csr(rng, nrows, ncols, 80)
        vals = (vals * np.float32(10.0) ** rng.integers(-6, 7, vals.size).astype(np.float32)).astype(np.float32)
        dead = rng.choice(ncols, 500, replace=False) + 1
        vals[np.isin(cols, dead)] = 0.0
        norm_o, vals_o = orc.normalize_columns(rp, cols, vals, ncols)
        y = rng.standard_normal(nrows)
        runs = []
        for _ in range(2):
            ctx.matrix_upload_csr(nrows, ncols, rp, cols, vals)
            assert ctx.debug_set("has_adj_copy") == adj_copy
            norm_g = ctx.normalize_columns()
            runs.append((norm_g, ctx.matrix_download_csr()[2], ctx.trans_mult_vector(y)))
        norm_g, vals_g, sty = runs[0]
        assert bits_equal(runs[1][0], norm_g) and bits_equal(runs[1][1], vals_g) and bits_equal(runs[1][2], sty)
        assert np.all(np.abs(norm_g - norm_o) <= nrows * 2.0 ** -53 * norm_o)
        assert np.all(norm_g[dead - 1] == 0.0) and np.array_equal(norm_g == 0.0, norm_o == 0.0)
        same_norm = (norm_g == norm_o)[cols - 1]
        print("normalize_columns: %d of %d norms identical to the sequential sums, worst relative distance %.2e" %
              (np.count_nonzero(norm_g == norm_o), ncols, np.max(np.abs(norm_g - norm_o) / np.maximum(norm_o, 1e-300))))
        report("normalize_columns", norms_identical=int(np.count_nonzero(norm_g == norm_o)), norms=int(ncols),
               worst_rel_distance=float(np.max(np.abs(norm_g - norm_o) / np.maximum(norm_o, 1e-300))))
        assert vals_g.size == vals_o.size and bits_equal(vals_g[same_norm], vals_o[same_norm])
        assert np.all(np.abs(vals_g.astype(np.float64) - vals_o) <= 2.0 ** -23 * np.abs(vals_o))
        # the adjoint (on the copy when there is one) sees the normalised values
        ref = orc.spmtv(rp, cols, vals_g, y, ncols)
        assert np.all(np.abs(sty - ref) <= 1e-12 * orc.spmtv(rp, cols, np.abs(vals_g), np.abs(y), ncols) + 1e-300)
        # dense storage (an uncompressed kernel): the same sequential sums as the reference -> identical bits
        nc, nr = 600, 40
        ctx.set_grid(nc, 1, 1, *[np.arange(nc, dtype=np.float64) + o for o in (0.0, 1.0)], *[np.full(nc, v) for v in (0.0, 1.0, 0.0, 1.0)])
        xs_, ys_, zs_ = np.linspace(0.3, nc - 0.7, nr), np.full(nr, 0.5), np.full(nr, -1.0)
        ctx.calculate_sensit(xs_, ys_, zs_, np.ones(nc), 0, 1.0)
        Sd = ctx.matrix_download_csr()
        nd_o, vd_o = orc.normalize_columns(*Sd, nc)
        nd = ctx.normalize_columns()
        assert bits_equal(nd, nd_o) and bits_equal(ctx.matrix_download_csr()[2], vd_o)
    finally:
        ctx.debug_set("adj_copy", 2)


@pytest.mark.parametrize("shape", CSR_SHAPES + [(2049, 4097, 40, False), (1, 16385, 900, False), (9000, 70, 5, False)])
def test_adjoint_on_the_transposed_copy(ctx, shape):
    """Debug key "adj_copy" = 1: the matrix gets a transposed copy of its tiles (built on the device from the tiles themselves) and
    b (+)= S^T x runs as the FORWARD kernel on that copy - no LDS atomic per non-zero (add_trans_mult_vector,
    sparse_matrix.f90:391-405).  Same results as the one-copy adjoint kernel and the oracle: overwrite and accumulate forms, the
    adjoint identity, bit-reproducible, the reload scaling applied to both copies, explicit zeros dropped."""
    nrows, ncols, mean_nnz, clustered = shape
    rng = np.random.default_rng(nrows * 11 + ncols)
    S = random_csr(rng, nrows, ncols, mean_nnz, clustered=clustered)
    x, y = rng.standard_normal(ncols), rng.standard_normal(nrows)
    absS = (S[0], S[1], np.abs(S[2]))
    reft = orc.spmtv(*S, y, ncols)
    tol = 1e-12 * orc.spmtv(*absS, np.abs(y), ncols) + 1e-300
    ctx.debug_set("adj_copy", 0)
    ctx.matrix_upload_csr(nrows, ncols, *S)
    assert ctx.debug_set("has_adj_copy") == 0 and not ctx.matrix_format()["adjoint_copy"]
    one_copy = ctx.trans_mult_vector(y)
    bytes_one = ctx.matrix_info()["device_bytes"]
    ctx.debug_set("adj_copy", 1)
    try:
        ctx.matrix_upload_csr(nrows, ncols, *S)
        assert ctx.debug_set("has_adj_copy") == 1 and ctx.matrix_format()["adjoint_copy"]
        assert ctx.matrix_info()["device_bytes"] > 1.5 * bytes_one
        back = ctx.matrix_download_csr()
        assert np.array_equal(back[0], S[0]) and np.array_equal(back[1], S[1]) and bits_equal(back[2], S[2])
        bt = ctx.trans_mult_vector(y)
        assert np.all(np.abs(bt - reft) <= tol) and np.all(np.abs(bt - one_copy) <= 2 * tol + _fixed_point_slack(S, y, ncols))
        t0 = rng.standard_normal(ncols)
        assert np.all(np.abs(ctx.trans_mult_vector(y, t0) - (t0 + reft)) <= tol + 1e-15 * np.abs(t0))
        b = ctx.mult_vector(x)
        assert np.all(np.abs(b - orc.spmv(*S, x)) <= 1e-12 * orc.spmv(*absS, np.abs(x)) + 1e-300)
        assert abs(np.dot(b, y) - np.dot(x, bt)) <= 1e-11 * np.dot(orc.spmv(*absS, np.abs(x)), np.abs(y))
        a = [ctx.trans_mult_vector(y) for _ in range(3)]
        assert bits_equal(a[1], a[0]) and bits_equal(a[2], a[0]) and np.all(np.abs(a[0] - reft) <= tol)
        scale = rng.uniform(0.1, 30.0, nrows)
        ctx.matrix_scale_rows(scale)
        Ss = (S[0], S[1], (S[2] * np.repeat(scale.astype(np.float32), np.diff(S[0]))).astype(np.float32))
        back = ctx.matrix_download_csr()
        assert bits_equal(back[2], Ss[2])
        refs = orc.spmtv(*Ss, y, ncols)
        assert np.all(np.abs(ctx.trans_mult_vector(y) - refs) <= 1e-12 * orc.spmtv(Ss[0], Ss[1], np.abs(Ss[2]), np.abs(y), ncols) + 1e-300)
        # a dense (uncompressed) kernel built into the same slot afterwards must not keep using the sparse matrix's copy
        # (found by the randomised host sweep with TFX_ADJ_COPY=1: matrix_begin_dense did not release the slot)
        if nrows * ncols <= 4_000_000:
            ctx.matrix_upload_csr(nrows, ncols, *S)
            assert ctx.debug_set("has_adj_copy") == 1
            ctx.set_grid(ncols, 1, 1, *[np.arange(ncols, dtype=np.float64) + o for o in (0.0, 1.0)], *[np.full(ncols, v) for v in (0.0, 1.0, 0.0, 1.0)])
            xs_, ys_, zs_ = np.linspace(0.3, ncols - 0.7, nrows), np.full(nrows, 0.5), np.full(nrows, -1.0)
            ctx.calculate_sensit(xs_, ys_, zs_, np.ones(ncols), 0, 1.0)              # ctype 0: dense storage in the slot
            assert ctx.debug_set("has_adj_copy") == 0
            Sd = ctx.matrix_download_csr()
            assert np.allclose(ctx.trans_mult_vector(y), orc.spmtv(*Sd, y, ncols), rtol=1e-10, atol=1e-10 * np.abs(Sd[2]).max() * np.abs(y).sum())
        # explicit zeros are stored in S (and come back on download) but not in the copy: the product does not change
        Sz = (S[0], S[1], S[2].copy())
        Sz[2][::3] = 0.0
        ctx.matrix_upload_csr(nrows, ncols, *Sz)
        assert np.all(np.abs(ctx.trans_mult_vector(y) - orc.spmtv(*Sz, y, ncols)) <= tol)
    finally:
        ctx.debug_set("adj_copy", 2)


@pytest.mark.parametrize("shape", [(3000, 40000, 300), (20000, 3000, 40), (9000, 9000, 60)])
def test_transposed_copy_is_the_same_whatever_the_panels(ctx, shape):
    """matrix_build_transpose makes S^T panel by panel; the panel shape follows the matrix and two budgets (entries per panel, ints
    of per-row tile index).  Whatever the cut - one panel, many bands of rows x all column tiles, full height x one column tile, bands
    of single column tiles - the copy holds the same rows, so the adjoint product has the same bits (the forward kernel on S^T
    adds in an order fixed by the rows of S^T, not by where its tiles are stored)."""
    nrows, ncols, per_row = shape
    rng = np.random.default_rng(nrows + 3 * ncols)
    S = _random_csr(rng, nrows, ncols, per_row)
    y = rng.standard_normal(nrows)
    reft = orc.spmtv(*S, y, ncols)
    tol = 1e-12 * orc.spmtv(S[0], S[1], np.abs(S[2]), np.abs(y), ncols) + 1e-300
    out = []
    try:
        for entries, budget in ((0, 0), (200000, 0), (50000, 0), (200000, 3 * ncols), (50000, 2 * ncols + 2), (50000, 5000)):
            ctx.debug_set("tr_panel_entries", entries)
            ctx.debug_set("tr_pos_budget", budget)
            ctx.matrix_upload_csr(nrows, ncols, *S)
            assert ctx.debug_set("has_adj_copy") == 1
            bt = ctx.trans_mult_vector(y)
            assert np.all(np.abs(bt - reft) <= tol), (entries, budget)
            out.append(bt)
        for bt in out[1:]:
            assert bits_equal(bt, out[0])
    finally:
        ctx.debug_set("tr_panel_entries", 0)
        ctx.debug_set("tr_pos_budget", 0)


def test_lsqr_with_the_adjoint_copy_matches_the_one_copy_solver(ctx, golden_dir):
    """The solver sees no difference: same iterates (to the products' rounding) with the adjoint on the transposed copy, also for the
    general constraint matrix C (it gets a copy too) - lsqr_solver2.F90:230-241."""
    g = np.load(os.path.join(golden_dir, "e2e_haar.npz"))
    N = int(g["nx"]) * int(g["ny"]) * int(g["nz"])
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    nd = S[0].size - 1
    b = g["np1_data_observed"]
    rng = np.random.default_rng(3)
    diag, rhs = [np.full(N, np.float32(1e-7), np.float32)], [rng.standard_normal(N) * 1e-8]
    out = {}
    for mode in (0, 1):
        ctx.debug_set("adj_copy", mode)
        ctx.matrix_upload_csr(nd, N, *S)
        out[mode] = [ctx.lsqr_solve_sensit(b, it, 1e-13, 0.0, 0.0, diag, rhs) for it in (3, 12)]
    ctx.debug_set("adj_copy", 2)
    # The two adjoint kernels sum in different orders (1e-16 per product) and the Golub-Kahan recurrence amplifies that: tight after 3
    # iterations; after 12 within the scatter the one-copy solver shows from run to run on this system (the reference-golden test of the
    # same system allows 1e-2 mid-convergence, tools/lsqr_scatter.py)
    for k, (tol_x, tol_r) in enumerate(((1e-9, 1e-10), (1e-3, 1e-6))):
        a, c = out[0][k], out[1][k]
        assert a[1] == c[1]
        dx = np.linalg.norm(a[0] - c[0]) / np.linalg.norm(a[0])
        assert dx <= tol_x and abs(a[2] - c[2]) <= tol_r * a[2], (k, dx, a[2], c[2])


def test_two_contexts_in_one_process(ctx):
    """A second context in the same process registers the large-LDS product kernels for itself (the attribute bookkeeping is per
    context, not a process-wide static) - both contexts give the oracle's products on a matrix that needs > 64 KB of LDS."""
    rng = np.random.default_rng(21)
    nrows, ncols = 9000, 30000
    S = _random_csr(rng, nrows, ncols, 120)
    x, y = rng.standard_normal(ncols), rng.standard_normal(nrows)
    ref, reft = orc.spmv(*S, x), orc.spmtv(*S, y, ncols)
    other = tfx.Context(0)
    try:
        for c in (other, ctx):
            c.debug_set("fwd_group", 4)
            c.matrix_upload_csr(nrows, ncols, *S)
            c.debug_set("fwd_group", 0)
            assert np.allclose(c.mult_vector(x), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
            assert np.allclose(c.trans_mult_vector(y), reft, rtol=1e-12, atol=1e-12 * np.abs(reft).max())
    finally:
        other.close()


def _csr_with_dense_columns(rng, nrows, ncols, ndense, per_row_sparse):
    """Rows that all keep (most of) a scattered set of `ndense` columns - like the coarse wavelet coefficients of a compressed
    kernel - plus a few random others."""
    dense_cols = np.sort(rng.choice(ncols, ndense, replace=False))
    rp, cols, vals = [0], [], []
    for r in range(nrows):
        keep = dense_cols[rng.random(ndense) < 0.85]
        extra = rng.choice(ncols, size=int(rng.integers(0, per_row_sparse + 1)), replace=False)
        c = np.unique(np.concatenate([keep, extra]))
        cols.append((c + 1).astype(np.int32))
        vals.append((rng.standard_normal(c.size) * 10.0 ** rng.uniform(-3, 3, c.size)).astype(np.float32))
        rp.append(rp[-1] + c.size)
    return np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals)


def test_lsqr_two_diagonal_blocks_and_soft_threshold(ctx, golden_dir):
    """Damping AND ADMM block together (two diagonal blocks, joint_inverse_problem.F90:452-527) with soft thresholding
    (lsqr_solver2.F90:478-494) against the oracle on the same [S; a I; b I] system."""
    g = load(golden_dir, "e2e_d4")
    N = int(g["nx"]) * int(g["ny"]) * int(g["nz"])
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    ctx.matrix_upload_csr(g["obs"].shape[0], N, *S)
    b = g["np1_data_observed"]
    rng = np.random.default_rng(9)
    d1 = np.full(N, np.float32(1e-7), np.float32)
    d2 = (np.float32(3e-7) * (1 + rng.random(N))).astype(np.float32)
    r1, r2 = rng.standard_normal(N) * 1e-10, rng.standard_normal(N) * 1e-10
    for gamma in (0.0, 1e-6):
        x, it, r = ctx.lsqr_solve_sensit(b, 6, 1e-13, gamma, 0.0, [d1, d2], [r1, r2])
        D1, D2 = orc.diag_csr(d1), orc.diag_csr(d2)
        Cm = (np.concatenate([D1[0], D1[0][-1] + D2[0][1:]]), np.concatenate([D1[1], D2[1]]), np.concatenate([D1[2], D2[2]]))
        xo, ito, ro = orc.lsqr(S, Cm, N, np.concatenate([b, r1, r2]), 6, 1e-13, gamma)
        assert it == ito == 6
        assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo), gamma
        assert abs(r - ro) <= 1e-9 * ro


def test_products_vs_reference_golden(ctx, golden_dir):
    g = load(golden_dir, "lsqr")
    for case in ("damp", "gen", "noC"):
        S = (orc.rc_to_rowptr(g[case + "_S_rc"]), g[case + "_S_cols"], g[case + "_S_vals"])
        ctx.matrix_upload_csr(int(g[case + "_nl_s"]), int(g[case + "_ncols"]), *S)
        assert np.allclose(ctx.mult_vector(g[case + "_xin"]), g[case + "_Sx"], rtol=1e-13, atol=1e-13)
        assert np.allclose(ctx.trans_mult_vector(g[case + "_yin"]), g[case + "_STy"], rtol=1e-13, atol=1e-13)


def test_matrix_validation_errors(ctx):
    with pytest.raises(tfx.TfxError) as e:
        ctx.matrix_upload_csr(1, 3, [0, 1], [4], [1.0])
    assert "column-index validation failed" in str(e.value)
    with pytest.raises(tfx.TfxError):
        ctx.matrix_upload_csr(1, 3, [0, 2], [2, 1], [1.0, 1.0])


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(kat_cases.cases()))
def test_lsqr_reference_known_answers(ctx, name):
    """The reference's own LSQR unit tests (src/tests/tests_lsqr.f90) through the HIP path."""
    c = kat_cases.cases()[name]
    S = kat_cases.dense_to_csr(c["A"])
    ctx.matrix_upload_csr(c["A"].shape[0], c["A"].shape[1], *S)
    x, it, r = ctx.lsqr_solve_sensit(c["b"], c["niter"], c["rmin"])
    kat_cases.check(c, x)


@pytest.mark.parametrize("case", ["damp", "noC"])
def test_lsqr_vs_reference_golden(ctx, golden_dir, case):
    g = load(golden_dir, "lsqr")
    nl_s, ncols = int(g[case + "_nl_s"]), int(g[case + "_ncols"])
    S = (orc.rc_to_rowptr(g[case + "_S_rc"]), g[case + "_S_cols"], g[case + "_S_vals"])
    ctx.matrix_upload_csr(nl_s, ncols, *S)
    b = g[case + "_b"]
    if case == "damp":
        diag, rhs = [g[case + "_C_vals"]], [b[nl_s:]]          # C = 0.1 I as one diagonal block
    else:
        diag, rhs = [], []
    for (niter, rmin, gamma), xref, rref, itref in zip(g[case + "_runs"], g[case + "_x"], g[case + "_r"], g[case + "_iters"]):
        x, it, r = ctx.lsqr_solve_sensit(b[:nl_s], int(niter), rmin, gamma, 0.0, diag, rhs)
        early = itref < niter
        # Both products and every norm are bit-reproducible since round 4, so these distances are fixed numbers on this hardware, not a
        # run-to-run scatter (rounds 1-3 allowed 1e-2 in x / 1e-3 in r for the LDS-atomic order of the old adjoint).  What remains is the
        # summation ORDER against the reference's sequential loops, which the 30-iteration iterate of this ill-conditioned 40 x 60 toy
        # amplifies: measured (profiles/r06_parity.json) x 5.98e-6, r 1.09e-7 there; every other run <= 5.9e-11 in x, <= 1.7e-15 in r.
        # Pinned at 10 x measured.
        mid = case == "damp" and int(niter) == 30
        if early:
            assert abs(it - itref) <= 0.1 * itref
        else:
            assert it == itref
            assert abs(r - rref) <= (1.1e-6 if mid else 2e-14) * abs(rref)
        tol = 1e-14 if niter <= 5 else (6e-5 if mid else 6e-10)
        report("lsqr_vs_reference_golden[%s, %d iterations]" % (case, int(niter)), x_rel_l2=float(np.linalg.norm(x - xref) / np.linalg.norm(xref)),
               r_rel=float(abs(r - rref) / abs(rref)), iterations=int(it), iterations_reference=int(itref), mid_convergence=bool(mid))
        assert np.linalg.norm(x - xref) <= tol * np.linalg.norm(xref), (case, niter)


def test_lsqr_general_constraint_matrix_vs_reference(ctx, golden_dir):
    """[S; C] with a general sparse C (empty rows included) uploaded as CSR - the reference's matrix_cons path
    (lsqr_solver2.F90:147, :211, :238) - against the reference's own solutions."""
    g = load(golden_dir, "lsqr")
    case = "gen"
    nl_s, nl_c, ncols = int(g[case + "_nl_s"]), int(g[case + "_nl_c"]), int(g[case + "_ncols"])
    S = (orc.rc_to_rowptr(g[case + "_S_rc"]), g[case + "_S_cols"], g[case + "_S_vals"])
    Cm = (orc.rc_to_rowptr(g[case + "_C_rc"]), g[case + "_C_cols"], g[case + "_C_vals"])
    b = g[case + "_b"]
    ctx.matrix_upload_csr(nl_s, ncols, *S)
    ctx.cons_upload_csr(*Cm, b[nl_s:])
    try:
        for (niter, rmin, gamma), xref, rref, itref in zip(g[case + "_runs"], g[case + "_x"], g[case + "_r"], g[case + "_iters"]):
            x, it, r = ctx.lsqr_solve_sensit(b[:nl_s], int(niter), rmin, gamma)
            early = itref < niter
            assert (abs(it - itref) <= 0.1 * itref) if early else (it == itref)
            tol = 1e-13          # measured <= 7.9e-15 on every run of this fixture (profiles/r06_parity.json)
            report("lsqr_general_constraint_matrix[%d iterations]" % int(niter), x_rel_l2=float(np.linalg.norm(x - xref) / np.linalg.norm(xref)),
                   r_rel=float(abs(r - rref) / abs(rref)), iterations=int(it), iterations_reference=int(itref))
            assert np.linalg.norm(x - xref) <= tol * np.linalg.norm(xref), (niter,)
        # diagonal blocks and a general C together: [S; C; 0.05 I] vs the oracle
        d = np.full(ncols, np.float32(0.05), np.float32)
        rhs = np.linspace(-1, 1, ncols)
        x, it, r = ctx.lsqr_solve_sensit(b[:nl_s], 300, 1e-13, 0.0, 0.0, [d], [rhs])
        Dm = orc.diag_csr(d)
        stacked = (np.concatenate([Cm[0], Cm[0][-1] + Dm[0][1:]]), np.concatenate([Cm[1], Dm[1]]), np.concatenate([Cm[2], Dm[2]]))
        xo, ito, ro = orc.lsqr(S, stacked, ncols, np.concatenate([b, rhs]), 300)
        assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)
    finally:
        ctx.cons_clear()


def test_lsqr_spatial_unknowns_equal_wavelet_domain_solution(ctx, golden_dir):
    """WAVELET_DOMAIN = F (lsqr_solver2.F90:200-206, :228-234).  Haar lifting is orthonormal, so LSQR on S.Wav with
    spatial unknowns produces x_k = InvWav(y_k) of the wavelet-domain run, iteration by iteration."""
    g = load(golden_dir, "e2e_haar")
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    ctx.matrix_upload_csr(g["obs"].shape[0], N, *S)
    b = g["np1_data_observed"]
    blocks = ([np.full(N, np.float32(1e-7), np.float32)], [np.zeros(N)])
    try:
        for niter in (1, 4, 15):
            y, it1, r1 = ctx.lsqr_solve_sensit(b, niter, 1e-13, 0.0, 0.0, *blocks)
            ctx.lsqr_set_wavelet_domain(False, 1)
            x, it2, r2 = ctx.lsqr_solve_sensit(b, niter, 1e-13, 0.0, 0.0, *blocks)
            ctx.lsqr_set_wavelet_domain(True)
            assert it1 == it2 == niter
            ref = ctx.inverse_wavelet(y, *dims, 1)
            assert np.linalg.norm(x - ref) <= (1e-11 if niter <= 4 else 1e-6) * np.linalg.norm(ref)
            assert abs(r1 - r2) <= 1e-9 * r1
    finally:
        ctx.lsqr_set_wavelet_domain(True)


def test_lsqr_stepping_api_matches_one_shot(ctx, golden_dir):
    g = load(golden_dir, "lsqr")
    case = "damp"
    nl_s, ncols = int(g[case + "_nl_s"]), int(g[case + "_ncols"])
    S = (orc.rc_to_rowptr(g[case + "_S_rc"]), g[case + "_S_cols"], g[case + "_S_vals"])
    ctx.matrix_upload_csr(nl_s, ncols, *S)
    b = g[case + "_b"]
    diag, rhs = [g[case + "_C_vals"]], [b[nl_s:]]
    x1, it1, r1 = ctx.lsqr_solve_sensit(b[:nl_s], 12, 1e-13, 0.0, 0.0, diag, rhs)
    ctx.lsqr_begin(b[:nl_s], 1e-13, 0.0, 0.0, diag, rhs)
    d1, _ = ctx.lsqr_iterate(5)
    d2, r2 = ctx.lsqr_iterate(7)
    x2 = ctx.lsqr_end()
    assert d1 + d2 == it1 == 12
    assert np.linalg.norm(x1 - x2) <= 1e-9 * np.linalg.norm(x1) and abs(r1 - r2) <= 1e-9 * r1


def test_lsqr_zero_rhs_is_exact(ctx):
    S = kat_cases.dense_to_csr(np.eye(4))
    ctx.matrix_upload_csr(4, 4, *S)
    x, it, r = ctx.lsqr_solve_sensit(np.zeros(4), 10)
    assert it == 0 and np.all(x == 0.0)


def test_lsqr_target_misfit_exit(ctx, golden_dir):
    g = load(golden_dir, "e2e_full")
    S = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    N = int(g["nx"]) * int(g["ny"]) * int(g["nz"])
    ctx.matrix_upload_csr(g["obs"].shape[0], N, *S)
    b = g["np1_data_observed"]
    alpha = np.float32(1e-6)
    blocks = ([np.full(N, alpha, np.float32)], [np.zeros(N)])
    target = 0.3 * np.sqrt(np.mean(b ** 2))
    x, it, r = ctx.lsqr_solve_sensit(b, 200, 1e-13, 0.0, target, *blocks)
    xo, ito, ro = orc.lsqr(S, orc.diag_csr(blocks[0][0]), N, np.concatenate([b, np.zeros(N)]), 200, 1e-13, 0.0, target)
    assert it == ito and it < 200
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)


# ---------------------------------------------------------------------------------------------------------------
def compare_built_matrix(built, ref, nd):
    """built / ref: (rowptr, cols, vals).  Returns (fraction of identical sparsity, max ulp distance on common entries)."""
    same, total, maxulp = 0, 0, 0
    for r in range(nd):
        cb, vb = built[1][built[0][r]:built[0][r + 1]], built[2][built[0][r]:built[0][r + 1]]
        cr, vr = ref[1][ref[0][r]:ref[0][r + 1]], ref[2][ref[0][r]:ref[0][r + 1]]
        assert abs(cb.size - cr.size) <= 2, (r, cb.size, cr.size)
        common, ib, ir = np.intersect1d(cb, cr, return_indices=True)
        same += common.size
        total += max(cb.size, cr.size)
        if common.size:
            ulp = np.abs(vb[ib].view(np.int32).astype(np.int64) - vr[ir].view(np.int32).astype(np.int64))
            maxulp = max(maxulp, int(ulp.max()))
    return same / max(total, 1), maxulp


@pytest.mark.parametrize("name", ["e2e_haar", "e2e_d4", "e2e_full"])
def test_build_kernel_vs_reference_sensit(ctx, golden_dir, name):
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = g["np1_column_weight"]
    obs = g["obs"]
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]), want_hist=True)
    built = ctx.matrix_download_csr()
    ref = (g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    frac, maxulp = compare_built_matrix(built, ref, obs.shape[0])
    assert frac >= 0.9999 and maxulp <= 1, (frac, maxulp)          # >= 99.99 % identical sparsity (measured: 1.0), values within 1 fp32 ulp
    assert abs(res["nnz"] - int(g["np1_nnz_total"])) <= 2 * obs.shape[0]
    assert int(res["nnz_hist"].sum()) == res["nnz"]
    if int(g["ctype"]) > 0:
        assert abs(res["comp_error"] - float(g["np1_comp_error"])) <= 1e-9 * float(g["np1_comp_error"])
    # partition rule on the built histogram vs the reference's 2-rank partition (exact unless a tie flipped)
    nel, nnz = tfx.get_load_balancing_nelements(res["nnz_hist"], 2)
    assert np.all(np.abs(nel - g["np2_nelements_at_cpu"]) <= 2)


@pytest.mark.parametrize("name", ["e2e_haar", "e2e_d4", "e2e_full"])
def test_end_to_end_inversion_vs_reference(ctx, golden_dir, name):
    """Build on the GPU, invert on the GPU, compare with the reference's final model / data."""
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
    obs = g["obs"]
    ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]))
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]))
    ref = g["np1_model_final"]
    assert np.linalg.norm(m - ref) <= 1e-6 * np.linalg.norm(ref)
    cost_ref = np.linalg.norm(g["np1_data_final"] - g["np1_data_observed"]) / np.linalg.norm(g["np1_data_observed"])
    assert abs(hist[-1]["cost"] - cost_ref) <= 1e-5 * cost_ref + 1e-16


def _lsqr_long_double(rp, cols, vals, alpha, d, N, K):
    """lsqr_solve_sensit's recurrence (lsqr_solver2.F90:115-290) on [S; alpha I] x = [d; 0] with every product and sum in numpy long
    double (64-bit mantissa): the reference trajectory of an UNCONVERGED run, against which two fp64 arithmetics can be ranked."""
    LD = np.longdouble
    c0 = cols.astype(np.int64) - 1
    rows = np.repeat(np.arange(rp.size - 1), np.diff(rp))
    order = np.argsort(c0, kind="stable")
    cT, rT, vT = c0[order], rows[order], vals[order].astype(LD)
    colptr = np.concatenate([[0], np.cumsum(np.bincount(cT, minlength=N))])
    v64 = vals.astype(LD)
    ne_r, ne_c = np.diff(rp) > 0, np.diff(colptr) > 0

    def seg(prod, ptr, ne, n):
        out = np.zeros(n, LD)
        out[ne] = np.add.reduceat(prod, ptr[:-1][ne])
        return out

    def A(x):
        return np.concatenate([seg(v64 * x[c0], rp, ne_r, rp.size - 1), LD(alpha) * x])

    def AT(u):
        nd = rp.size - 1
        return seg(vT * u[:nd][rT], colptr, ne_c, N) + LD(alpha) * u[nd:]

    def norm(v):
        return np.sqrt(np.sum(v * v))

    u = np.concatenate([d, np.zeros(N)]).astype(LD)
    beta = norm(u); u /= beta; b1 = beta
    v = AT(u); al = norm(v); v /= al
    w = v.copy(); x = np.zeros(N, LD); phibar = beta; rhobar = al
    for _ in range(K):
        u = A(v) - al * u; beta = norm(u); u /= beta
        v = AT(u) - beta * v; al = norm(v); v /= al
        rho = np.sqrt(rhobar * rhobar + beta * beta)
        c, s_ = rhobar / rho, beta / rho
        theta = s_ * al; rhobar = -c * al; phi = c * phibar; phibar = s_ * phibar
        x = x + (phi / rho) * w
        w = v - (theta / rho) * w
    return x.astype(np.float64), float(phibar / b1)


def test_unconverged_lsqr_is_closer_to_extended_precision_than_sequential_fp64(ctx):
    """Why the GPU host's 1 x 101-iteration run of bench.py's mid-scale problem sits 2.5e-7 from the reference's model (15 x the
    reference's own rank scatter) with a LOWER data cost: 101 iterations of lsqr_solve_sensit (lsqr_solver2.F90:47-308) on the same
    128x128x32-cell x 1024-data system in three arithmetics - fp64 with the reference's sequential sums (sparse_matrix.f90:316-329,
    :391-405: the C oracle), the HIP path (tile / tree order, fused multiply-adds), and 80-bit long double.  The recurrence is not
    converged, so rounding differences of the sums are amplified by orders of magnitude; the HIP path must be the one NEARER the
    long-double trajectory - in the solution and in the residual - i.e. its distance to the reference is the reference's rounding.
    Both adjoints: on the transposed copy (fp64 sums) and on the tiles of S (61-bit fixed-point sums)."""
    nx, ny, nz, ox, oy, rate, K = 128, 128, 32, 32, 32, 0.05, 101
    N = nx * ny * nz
    alpha = np.float32(1e-7)
    gpu = {}
    try:
        for adj_copy in (2, 0):
            ctx.debug_set("adj_copy", adj_copy)
            ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
            cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
            xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
            ctx.calculate_sensit(xs, ys, zs, cw, 1, rate)
            assert bool(ctx.matrix_format()["adjoint_copy"]) == (adj_copy != 0)
            mtrue = tfx.synthetic.true_model(nx, ny, nz)
            d = ctx.calc_data(ctx.forward_wavelet(mtrue / cw, nx, ny, nz, 1), 1.0, None)
            x_gpu, it, r_gpu = ctx.lsqr_solve_sensit(d, K, 1e-300, 0.0, 0.0, [np.full(N, alpha, np.float32)], [np.zeros(N)])
            assert it == K
            csr = ctx.matrix_download_csr()
            if gpu:
                assert all(np.array_equal(p, q) for p, q in zip(csr, gpu[2][3])) and np.array_equal(d, gpu[2][2])    # the same system
            gpu[adj_copy] = (x_gpu, r_gpu, d, csr)
    finally:
        ctx.debug_set("adj_copy", 2)
    rp, cols, vals = gpu[2][3]
    d = gpu[2][2]
    x_seq, it2, r_seq = orc.lsqr((rp, cols, vals), orc.diag_csr(np.full(N, alpha, np.float32)), N, np.concatenate([d, np.zeros(N)]), K)
    x_ext, r_ext = _lsqr_long_double(rp, cols, vals, alpha, d, N, K)
    assert it2 == K

    def rel(p, q):
        return float(np.linalg.norm(p - q) / np.linalg.norm(q))

    s_e = rel(x_seq, x_ext)
    for adj_copy in (2, 0):
        x_gpu, r_gpu = gpu[adj_copy][:2]
        g_e, g_s = rel(x_gpu, x_ext), rel(x_gpu, x_seq)
        print("LSQR x %d, adj_copy %d: residual seq64 %.9e / gpu %.9e / ext80 %.9e; model rel-L2 gpu-ext80 %.2e, seq64-ext80 %.2e, gpu-seq64 %.2e" %
              (K, adj_copy, r_seq, r_gpu, r_ext, g_e, s_e, g_s))
        report("unconverged_lsqr_vs_extended_precision[adj_copy=%d]" % adj_copy, iterations=int(K), r_seq64=r_seq, r_gpu=r_gpu, r_ext80=r_ext,
               model_gpu_vs_ext80=g_e, model_seq64_vs_ext80=s_e, model_gpu_vs_seq64=g_s)
        assert g_e <= s_e, (g_e, s_e)                                   # the solution: nearer the long-double trajectory than the reference's arithmetic
        assert abs(r_gpu - r_ext) <= abs(r_seq - r_ext), (r_gpu, r_seq, r_ext)      # and so is the residual
        assert g_s <= 2.0 * (g_e + s_e)                                 # (triangle: nothing else separates the two fp64 runs)


@pytest.mark.parametrize("name", ["e2e_medium_haar", "e2e_medium_d4"])
def test_medium_scale_end_to_end_vs_reference(ctx, golden_dir, name):
    """Mid-scale parity against the REFERENCE'S OWN OUTPUT (tests/golden/e2e_medium_*.npz: 64x64x32 cells x 1024 data, r = 0.05,
    2 x 100 LSQR iterations, run by oracle/_ref/tomofastx at 1, 2, 4 and 8 ranks): the kernel (nnz, per-column histogram, 8 full
    SENSIT rows, compression error), the final model, the data and the LSQR residuals.  The reference's own scatter between its
    rank counts on these runs is 6e-10 .. 1.4e-9 in the final model (stored in the fixture); the HIP path differs from the
    1-rank run by the same reordering of the sums PLUS last-bit differences of device log / atan2 in the kernel values (a
    1-ulp-of-fp32 perturbation of a few entries of S) - its distance is asserted against a stated tolerance and
    written to gpurun_out/ so that profiles/ can carry the measured number."""
    import json
    g = load(golden_dir, name)
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    N, ctype, rate = nx * ny * nz, int(g["ctype"]), float(g["rate"])
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, int(g["ox"]), int(g["oy"]))
    nd = xs.size
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
    assert np.max(np.abs(cw - g["column_weight"]) / g["column_weight"]) <= 1e-14
    res = ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, want_hist=True)
    # ---- the kernel
    assert abs(res["nnz"] - int(g["nnz_total"])) <= 2 * nd
    assert abs(res["comp_error"] - float(g["comp_error"])) <= 1e-9 * float(g["comp_error"])
    hist_same = float(np.mean(res["nnz_hist"] == g["sensit_nnz"]))
    assert hist_same >= 0.9999, hist_same
    for k in (2, 4, 8):     # the reference's partition at 2 / 4 / 8 ranks from the histogram built here (exact unless a tie flipped)
        nel, _ = tfx.get_load_balancing_nelements(res["nnz_hist"], k)
        assert np.all(np.abs(nel - g["np%d_nelements_at_cpu" % k]) <= 2)
    rp, cols, vals = ctx.matrix_download_csr()
    same = total = 0
    worst_ulp, worst_rel_scale = 0, 0.0
    for i, r in enumerate(g["rows_kept"]):
        cb, vb = cols[rp[r]:rp[r + 1]], vals[rp[r]:rp[r + 1]]
        cr, vr = g["cols"][g["row_ptr"][i]:g["row_ptr"][i + 1]], g["vals"][g["row_ptr"][i]:g["row_ptr"][i + 1]]
        common, ib, ir = np.intersect1d(cb, cr, return_indices=True)
        same += common.size
        total += max(cb.size, cr.size)
        ulp = np.abs(vb[ib].view(np.int32).astype(np.int64) - vr[ir].view(np.int32).astype(np.int64))
        worst_ulp = max(worst_ulp, int(ulp.max()))
        worst_rel_scale = max(worst_rel_scale, float(np.abs(vb[ib].astype(np.float64) - vr[ir]).max() / np.abs(vr).max()))
    frac = same / total
    assert frac >= 0.9999, frac
    assert worst_rel_scale <= 1e-8, (worst_ulp, worst_rel_scale)
    # ---- the inversion
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, ctype, g["data_observed"], int(g["nmajor"]), int(g["nminor"]), alpha=float(g["alpha"]))
    ref = g["model_final"]
    model_rel = float(np.linalg.norm(m - ref) / np.linalg.norm(ref))
    data_rel = float(np.linalg.norm(d - g["data_final"]) / np.linalg.norm(g["data_final"]))
    cost_ref = float(np.linalg.norm(g["data_final"] - g["data_observed"]) / np.linalg.norm(g["data_observed"]))
    cost_rel = abs(hist[-1]["cost"] - cost_ref) / cost_ref
    ref_scatter = max(float(g["np%d_model_rel_l2_vs_np1" % k]) for k in (2, 4, 8))
    out = {"fixture": name, "cells": N, "data": nd, "nnz": int(res["nnz"]), "nnz_reference": int(g["nnz_total"]),
           "column_histogram_identical": hist_same, "rows_identical_sparsity": frac, "rows_worst_fp32_ulp": worst_ulp,
           "rows_worst_diff_over_row_max": worst_rel_scale, "model_rel_l2_vs_reference_np1": model_rel,
           "reference_own_scatter_np2_4_8_vs_np1": ref_scatter, "data_rel_l2": data_rel, "data_cost_rel_diff": cost_rel,
           "data_cost": hist[-1]["cost"], "data_cost_reference": cost_ref}
    print(json.dumps(out))
    report("medium_scale_end_to_end[%s]" % name, **out)
    try:
        os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "medium_parity_%s.json" % name), "w"), indent=1)
    except OSError:
        pass
    # Stated tolerance at this scale: final model 5e-8 (measured 4.7e-9 = 7 x the reference's own rank-count
    # scatter), calculated data 1e-8 (measured 8.5e-10).  The data cost |d_calc - d_obs| / |d_obs| of these converged runs is 2.8e-9,
    # i.e. of the size of the data tolerance itself: it is bounded in absolute terms (1e-8 of |d_obs|), its relative difference is
    # recorded, not asserted.
    assert model_rel <= 5e-8, out
    assert data_rel <= 1e-8 and abs(hist[-1]["cost"] - cost_ref) <= 1e-8, out
    # a second solve gives the same bits (reproducible products, fixed-order reductions)
    m2, d2, _ = tfx.inversion.solve_problem_gravity(ctx, cw, ctype, g["data_observed"], int(g["nmajor"]), int(g["nminor"]), alpha=float(g["alpha"]))
    assert bits_equal(m2, m) and bits_equal(d2, d)


# ---------------------------------------------------------------------------------------------------------------
# gradiometry / multi-component kernels
def test_gradiprism_rows_vs_reference(ctx, golden_dir):
    """gradiprism_zz and gradiprism_full rows vs the reference's.  The tensor components are sums of 8 atan2 (<= 2 pi) or
    8 log terms that cancel; device libm differs by <= 2 ulp per term -> bound = a few ulp of G * sum|terms|."""
    g = load(golden_dir, "gradprism")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *grid)
    obs = g["obs"]
    zz = ctx.sensit_lines(1, obs[:, 0], obs[:, 1], obs[:, 2], data_type=2, ndata_components=1)
    full = ctx.sensit_lines(1, obs[:, 0], obs[:, 1], obs[:, 2], data_type=2, ndata_components=6)
    assert zz.shape == (obs.shape[0], 1, 1, grid[0].size) and full.shape == (obs.shape[0], 6, 1, grid[0].size)
    G = 6.674e-11
    for i, o in enumerate(obs):
        span = np.abs(np.stack([o[0] - grid[0], o[0] - grid[1], o[1] - grid[2], o[1] - grid[3], o[2] - grid[4], o[2] - grid[5]])).max(0)
        atan_scale = G * 8 * 2 * np.pi
        log_scale = G * 8 * (np.abs(np.log(2 * span)) + 40.0)         # |log(R + Z)|, |log((R - Y)/(R + Y))| of near-degenerate corners
        assert np.all(np.abs(zz[i, 0, 0] - g["rows_zz"][i]) <= 8 * 2.3e-16 * atan_scale)
        assert bits_equal(full[i, 2, 0], zz[i, 0, 0])                  # ZZ of the full tensor is the same arithmetic
        for c in range(6):
            scale = atan_scale if c < 3 else log_scale
            assert np.all(np.abs(full[i, c, 0] - g["rows_full"][i, c]) <= 8 * 2.3e-16 * scale), (i, c)
            assert np.max(np.abs(full[i, c, 0] - g["rows_full"][i, c])) <= 1e-9 * np.max(np.abs(g["rows_full"][i, c]))


def g3_term_scale(grid, o):
    """Magnitude of the 24 (x 3) terms a graviprism_full entry is the cancelling sum of (gravity_field.f90:110-112)."""
    X1, X2, Y1, Y2, Z1, Z2 = grid
    s = np.zeros(X1.size)
    for xx in (o[0] - X1, o[0] - X2):
        for yy in (o[1] - Y1, o[1] - Y2):
            for zz in (o[2] - Z1, o[2] - Z2):
                R = np.sqrt(xx * xx + yy * yy + zz * zz)
                lg = np.abs(np.log(R + xx)) + np.abs(np.log(R + yy)) + np.abs(np.log(R + zz))
                s += (np.abs(xx) + np.abs(yy) + np.abs(zz)) * (2 * np.pi + lg)
    return 6.674e-11 * s


def test_graviprism_full_rows_vs_reference(ctx, golden_dir):
    """graviprism_full (gravity_field.f90:41-126): gx, gy, gz rows of the HIP path vs the compiled reference's LineX /
    LineY / LineZ on the 8x6x5 non-uniform grid (observations above, beside and inside the mesh).  The gz sub-row has the bits of the
    graviprism_z row (same corner expression, same order); gx / gy sit within a few ulp of the cancelling TERMS and 1e-9 of the row
    maximum - the row-scale bound of DESIGN section 4."""
    g = load(golden_dir, "prism")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *grid)
    obs = g["obs"]
    full = ctx.graviprism_full(obs[:, 0], obs[:, 1], obs[:, 2])
    gz = ctx.graviprism_z(obs[:, 0], obs[:, 1], obs[:, 2])
    assert full.shape == (obs.shape[0], 3, grid[0].size) == g["rows_full"].shape
    worst = 0.0
    for i, o in enumerate(obs):
        assert bits_equal(full[i, 2], gz[i])
        scale = g3_term_scale(grid, o)
        for c in range(3):
            ref = g["rows_full"][i, c]
            assert np.all(np.abs(full[i, c] - ref) <= 8 * 2.3e-16 * scale), (i, c)
            rel = float(np.max(np.abs(full[i, c] - ref)) / np.max(np.abs(ref)))
            worst = max(worst, rel)
            assert rel <= 1e-9, (i, c, rel)
    report("graviprism_full_rows_vs_reference", worst_row_scale_distance=worst, rows=int(obs.shape[0] * 3))


def test_graviprism_full_tensor_path_bit_identical_to_general(ctx, golden_dir):
    """k_prism_g3_tensor (shared nodes in LDS) vs k_prism_g3 (six arrays): same bits; a grid without shared faces takes the general
    kernel and agrees with the oracle; and the whole 3-component build gives the same matrix on either generator."""
    g = load(golden_dir, "prism")
    cases = [((int(g["nx"]), int(g["ny"]), int(g["nz"])), [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")], g["obs"])]
    xs, ys, zs = tfx.synthetic.observations(70, 37, 3, 2)
    cases.append(((70, 37, 19), list(tfx.synthetic.grid(70, 37, 19)), np.stack([xs, ys, zs], 1)))
    for dims, grid, obs in cases:
        ctx.set_grid(*dims, *grid)
        assert ctx.debug_set("tensor_grid") == 1
        rows_t = ctx.graviprism_full(obs[:, 0], obs[:, 1], obs[:, 2])
        cw = orc.column_weight_type1(grid, 2.0, 0.0, 1.0)
        ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, 0.2, data_type=1, ndata_components=3)
        A = ctx.matrix_download_csr()
        ctx.debug_set("force_general_prism", 1)
        assert ctx.debug_set("tensor_grid") == 0
        rows_g = ctx.graviprism_full(obs[:, 0], obs[:, 1], obs[:, 2])
        ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, 0.2, data_type=1, ndata_components=3)
        B = ctx.matrix_download_csr()
        ctx.debug_set("force_general_prism", 0)
        assert bits_equal(rows_t, rows_g)
        assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and A[2].tobytes() == B[2].tobytes()
    dims, grid, obs = cases[0]
    bent = [a.copy() for a in grid]
    bent[1][7] += 1e-9
    ctx.set_grid(*dims, *bent)
    assert ctx.debug_set("tensor_grid") == 0
    ierr, ref = orc.graviprism_full(bent, *obs[0])
    r = ctx.graviprism_full(obs[:1, 0], obs[:1, 1], obs[:1, 2])[0]
    assert ierr == 0 and np.all(np.abs(r - ref) <= 8 * 2.3e-16 * g3_term_scale(bent, obs[0]))


def test_graviprism_full_geometry_errors(ctx):
    """The three abort tests of graviprism_full (gravity_field.f90:96-104), with the reference's messages; (XY) is the one graviprism_z
    does not have - the same observation is fine for the one-component rows."""
    one = [np.array([v], np.float64) for v in (0.0, 1.0, 0.0, 1.0, 0.0, 1.0)]
    for force_general in (0, 1):
        ctx.set_grid(1, 1, 1, *one)
        ctx.debug_set("force_general_prism", force_general)
        try:
            for o, plane in (((-1.0, 0.0, 0.0), "(YZ)"), ((0.0, -1.0, 0.0), "(XZ)"), ((0.0, 0.0, -1.0), "(XY)")):
                with pytest.raises(tfx.TfxError) as e:
                    ctx.graviprism_full([o[0]], [o[1]], [o[2]])
                assert e.value.code == -3 and "coincides with model grid boundary " + plane in str(e.value), (o, str(e.value))
            assert ctx.graviprism_z([0.0], [0.0], [-1.0]).shape == (1, 1)
            assert ctx.graviprism_full([2.0], [3.0], [-1.0]).shape == (1, 3, 1)
        finally:
            ctx.debug_set("force_general_prism", 0)
    with pytest.raises(tfx.TfxError):
        ctx.sensit_lines(1, [2.0], [3.0], [-1.0], data_type=1, ndata_components=2)


@pytest.mark.parametrize("ctype,rate", [(1, 0.15), (2, 0.15), (0, 1.0)])
def test_build_graviprism_full_kernel_vs_oracle(ctx, ctype, rate):
    """The three-component gravity kernel through the whole build (weights, wavelets, threshold, compaction; matrix rows 3 i + c) vs
    the oracle's build on graviprism_full lines, and its gz rows vs the one-component build, bit for bit."""
    nx, ny, nz = 20, 14, 9
    N = nx * ny * nz
    rng = np.random.default_rng(77)
    grid = nonuniform(nx, ny, nz, rng)
    xe0, xe1, ye0, ye1 = grid[0].min(), grid[1].max(), grid[2].min(), grid[3].max()
    obs = np.stack([rng.uniform(xe0 - 50, xe1 + 50, 7), rng.uniform(ye0 - 50, ye1 + 50, 7), -rng.uniform(0.5, 30.0, 7)], 1)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = orc.column_weight_type1(grid, 2.0, 0.0, 1.0)
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, ctype, rate, want_hist=True, data_type=1, ndata_components=3)
    info = ctx.matrix_info()
    assert info["nrows"] == 3 * obs.shape[0] and info["ncols"] == N
    built = ctx.matrix_download_csr()
    rp, cols, vals, hist, err = orc.build_matrix_comp("g3", grid, (nx, ny, nz), cw, obs, ctype, rate, None, 1, 3)
    frac, maxulp = compare_built_matrix(built, (rp, cols, vals), 3 * obs.shape[0])
    assert frac >= 0.9999 and maxulp <= 2, (frac, maxulp)
    assert abs(res["nnz"] - int(rp[-1])) <= 2 * 3 * obs.shape[0]
    if ctype > 0:
        assert abs(res["comp_error"] - err) <= 1e-6 * err
    ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, ctype, rate)
    one = ctx.matrix_download_csr()
    for i in range(obs.shape[0]):
        a0, a1 = built[0][3 * i + 2], built[0][3 * i + 3]
        b0, b1 = one[0][i], one[0][i + 1]
        assert np.array_equal(built[1][a0:a1], one[1][b0:b1]) and built[2][a0:a1].tobytes() == one[2][b0:b1].tobytes(), i
    report("build_graviprism_full_kernel_vs_oracle[%d]" % ctype, identical_sparsity=frac, max_fp32_ulp=maxulp)


def test_magprism_components_vs_reference(ctx, golden_dir):
    """magprism with a magnetisation-vector model (3 model components) and / or three-component data."""
    g = load(golden_dir, "magprism_comp")
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(int(g["nx"]), int(g["ny"]), int(g["nz"]), *grid)
    obs, field = g["obs"], g["field"]
    for ncm, ncd in ((1, 3), (3, 1), (3, 3)):
        ref = g["rows_m%d_d%d" % (ncm, ncd)]
        rows = ctx.sensit_lines(2, obs[:, 0], obs[:, 1], obs[:, 2], ndata_components=ncd, nmodel_components=ncm, mag_field=field)
        assert rows.shape == ref.shape
        # vector models are scaled by mu0*1e9 = 400 pi instead of the field intensity (magnetic_field.f90:286-291)
        inten = field[3] if ncm == 1 else 400.0 * np.pi
        for i, o in enumerate(obs):
            scale = mag_term_scale(grid, o, inten)
            for d in range(ncd):
                for k in range(ncm):
                    assert np.all(np.abs(rows[i, d, k] - ref[i, d, k]) <= 8 * 2.3e-16 * scale), (ncm, ncd, i, d, k)


COMP_CASES = ["e2e_gzz", "e2e_ftg", "e2e_mag13", "e2e_mag31", "e2e_mag33"]


def comp_kwargs(g):
    kw = dict(data_type=int(g["gtype"]), ndata_components=int(g["ncd"]), nmodel_components=int(g["ncm"]))
    if int(g["prob"]) == 2:
        kw["mag_field"] = g["field"]
    return kw


@pytest.mark.parametrize("name", COMP_CASES)
def test_build_multicomponent_kernel_vs_reference_sensit(ctx, golden_dir, name):
    """Gzz / full-tensor / three-component magnetic kernels built on the GPU vs the lines of the reference's SENSIT file."""
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    ncm, ncd = int(g["ncm"]), int(g["ncd"])
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = g["np1_column_weight"]
    obs = g["obs"]
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]), want_hist=True, **comp_kwargs(g))
    info = ctx.matrix_info()
    assert info["nrows"] == obs.shape[0] * ncd and info["ncols"] == ncm * N
    built = ctx.matrix_download_csr()
    # reference lines (i, d, k) -> matrix rows (i, d) with component k in columns k*N + cell
    sub_rp = g["np1_row_ptr"]
    kk = np.repeat(np.tile(np.arange(ncm), (sub_rp.size - 1) // ncm), np.diff(sub_rp))
    ref = (sub_rp[::ncm], (g["np1_cols"] + kk * N).astype(np.int32), g["np1_vals"])
    frac, maxulp = compare_built_matrix(built, ref, obs.shape[0] * ncd)
    # magnetic tensor entries cancel heavily: fp32 values of small coefficients may move by more than 2 ulp
    assert frac >= 0.9999 and (maxulp <= 1 or int(g["prob"]) == 2), (frac, maxulp)
    assert abs(res["nnz"] - int(g["np1_nnz_total"])) <= 2 * obs.shape[0] * ncd * ncm
    assert int(res["nnz_hist"].sum()) == res["nnz"]
    if int(g["ctype"]) > 0:
        assert abs(res["comp_error"] - float(g["np1_comp_error"])) <= 1e-6 * float(g["np1_comp_error"])
    nel, nnz = tfx.get_load_balancing_nelements(res["nnz_hist"], 2)
    assert np.all(np.abs(nel - g["np2_nelements_at_cpu"]) <= 2)


@pytest.mark.parametrize("name", COMP_CASES)
@pytest.mark.parametrize("matrix", ["reference", "built"])
def test_multicomponent_end_to_end_vs_reference(ctx, golden_dir, name, matrix):
    """Inversion on the GPU vs the reference's final model.  matrix = "reference": the SENSIT lines of the reference uploaded as
    CSR (isolates the solver); "built": the kernel built on the GPU (fp32 values differ by <= 2 ulp, a few threshold ties).
    The fixtures run LSQR to convergence (solves that stop mid-convergence amplify rounding differences chaotically - the
    reference's own 1- and 2-rank runs then differ by up to 1e-2); self_diff is the reference's 1- vs 2-rank difference."""
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    ncm, ncd = int(g["ncm"]), int(g["ncd"])
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = g["np1_column_weight"]
    obs = g["obs"]
    if matrix == "built":
        ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]), **comp_kwargs(g))
    else:
        sub_rp = g["np1_row_ptr"]
        kk = np.repeat(np.tile(np.arange(ncm), (sub_rp.size - 1) // ncm), np.diff(sub_rp))
        ctx.matrix_upload_csr(obs.shape[0] * ncd, ncm * N, sub_rp[::ncm], (g["np1_cols"] + kk * N).astype(np.int32), g["np1_vals"])
    d_obs = g["np1_data_observed"].ravel()
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), d_obs, int(g["nmajor"]), int(g["nminor"]),
                                                     alpha=float(g["alpha"]), nmodel_components=ncm)
    ref = np.ascontiguousarray(g["np1_model_final"].T).ravel()
    ref2 = np.ascontiguousarray(g["np2_model_final"].T).ravel()
    self_diff = np.linalg.norm(ref2 - ref) / np.linalg.norm(ref)
    tol = max(1e-6, (10.0 if matrix == "reference" else 100.0) * self_diff)
    assert np.linalg.norm(m - ref) <= tol * np.linalg.norm(ref), (np.linalg.norm(m - ref) / np.linalg.norm(ref), self_diff)
    assert np.allclose(hist[0]["r"], g["np1_lsqr_r"][0], rtol=1e-3)
    dref = g["np1_data_final"].ravel()
    assert np.linalg.norm(d - dref) <= 10.0 * tol * np.linalg.norm(dref)


def test_multicomponent_dense_and_column_range(ctx):
    """Model-component column blocks with a column range (what a rank of a partitioned build keeps), compressed and dense:
    the built block equals the same columns of the full build."""
    nx, ny, nz = 12, 10, 6
    N = nx * ny * nz
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, 3, 2)
    field = (-62.0, 11.0, 20.0, 57000.0)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = orc.column_weight_type1(grid, 3.0, 0.0, 1.0)
    dw = np.linspace(0.5, 1.5, xs.size * 3).reshape(xs.size, 3)
    for ctype, rate in ((1, 0.3), (0, 1.0)):
        kw = dict(mag_field=field, ndata_components=3, nmodel_components=3, data_weight=dw, problem_weight=0.7)
        ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, **kw)
        rp, cols, vals = ctx.matrix_download_csr()
        c0, c1 = 137, 500
        ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, col_range=(c0, c1), **kw)
        info = ctx.matrix_info()
        assert info["nrows"] == xs.size * 3 and info["ncols"] == 3 * (c1 - c0)
        rp2, cols2, vals2 = ctx.matrix_download_csr()
        for r in range(xs.size * 3):
            c, v = cols[rp[r]:rp[r + 1]] - 1, vals[rp[r]:rp[r + 1]]
            comp, cell = c // N, c % N
            keep = (cell >= c0) & (cell < c1)
            want_c = comp[keep] * (c1 - c0) + (cell[keep] - c0) + 1
            assert np.array_equal(cols2[rp2[r]:rp2[r + 1]], want_c), r
            assert bits_equal(vals2[rp2[r]:rp2[r + 1]], v[keep]), r


# ---------------------------------------------------------------------------------------------------------------
# joint inversion: two kernels, one LSQR system (BASELINE config 4 at fixture size)
def build_joint(ctx, g, matrix):
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    probs = []
    for i, tag in enumerate(("grav", "magn")):
        cw = g["np1_%s_column_weight" % tag]
        pw = float(g["pw"][i])
        obs = g["obs_%s" % tag]
        ctx.select_problem(i)
        if matrix == "built":
            ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]), problem_weight=pw,
                                 mag_field=g["field"] if i == 1 else None)
        else:
            vals = (g["np1_%s_vals" % tag] * np.float32(pw)).astype(np.float32)
            ctx.matrix_upload_csr(obs.shape[0], N, g["np1_%s_row_ptr" % tag], g["np1_%s_cols" % tag], vals)
        probs.append(dict(column_weight=cw, data_obs=g["np1_%s_data_observed" % tag], problem_weight=pw, alpha=float(g["alpha"][i])))
    ctx.select_problem(0)
    return dims, N, probs


@pytest.mark.parametrize("matrix", ["reference", "built"])
def test_joint_inversion_vs_reference(ctx, golden_dir, matrix):
    g = load(golden_dir, "e2e_joint")
    try:
        dims, N, probs = build_joint(ctx, g, matrix)
        assert ctx.system_dims() == (g["obs_grav"].shape[0] + g["obs_magn"].shape[0], 2 * N)
        m, d, hist = tfx.inversion.solve_problem_joint(ctx, probs, int(g["ctype"]), int(g["nmajor"]), int(g["nminor"]))
        for i, tag in enumerate(("grav", "magn")):
            ref = g["np1_%s_model_final" % tag]
            self_diff = np.linalg.norm(g["np2_%s_model_final" % tag] - ref) / np.linalg.norm(ref)
            tol = max(1e-6, (10.0 if matrix == "reference" else 100.0) * self_diff)
            assert np.linalg.norm(m[i] - ref) <= tol * np.linalg.norm(ref), (tag, np.linalg.norm(m[i] - ref) / np.linalg.norm(ref))
            dref = g["np1_%s_data_final" % tag]
            assert np.linalg.norm(d[i] - dref) <= 10 * tol * np.linalg.norm(dref)
        assert np.allclose(hist[0]["r"], g["np1_lsqr_r"][0], rtol=0.5) or hist[0]["r"] < 1e-9
    finally:
        ctx.select_problem(1)
        ctx.matrix_free()
        ctx.select_problem(0)


def test_joint_products_are_block_diagonal(ctx, golden_dir):
    """LSQR on blockdiag(S1, S2) with zero right-hand side in one block leaves that block's unknowns at zero, and equals the
    single-problem solve of the other block."""
    g = load(golden_dir, "e2e_joint")
    try:
        dims, N, probs = build_joint(ctx, g, "reference")
        nd1, nd2 = g["obs_grav"].shape[0], g["obs_magn"].shape[0]
        b = np.concatenate([probs[0]["data_obs"], np.zeros(nd2)])
        xj, itj, rj = ctx.lsqr_solve_sensit(b, 30, 1e-13)
        assert np.all(xj[N:] == 0.0)
        ctx.select_problem(1)
        ctx.matrix_free()
        ctx.select_problem(0)
        xs, its, rs = ctx.lsqr_solve_sensit(probs[0]["data_obs"], 30, 1e-13)
        assert its == itj and np.linalg.norm(xs - xj[:N]) <= 1e-9 * np.linalg.norm(xs)
    finally:
        ctx.select_problem(1)
        try:
            ctx.matrix_free()
        finally:
            ctx.select_problem(0)


# ---------------------------------------------------------------------------------------------------------------
# degenerate shapes the reference accepts
def test_build_edge_cases(ctx):
    """rate 0 (nothing kept), one datum, a 1 x 1 x 1 grid, fewer cells than a wave, a single kept entry per row."""
    # (1) compression rate 0: K = 0 -> every row empty (sensitivity_gravmag.F90:64-77, :244-256); the matrix is valid and S x = 0
    nx, ny, nz = 6, 5, 4
    N = nx * ny * nz
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, 3, 2)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = orc.column_weight_type1(grid)
    res = ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.0, want_hist=True)
    assert res["nnz"] == 0 and int(res["nnz_hist"].sum()) == 0
    info = ctx.matrix_info()
    assert info["nrows"] == xs.size and info["ncols"] == N and info["nnz"] == 0
    assert np.all(ctx.mult_vector(np.ones(N)) == 0.0) and np.all(ctx.trans_mult_vector(np.ones(xs.size)) == 0.0)
    rp, cols, vals = ctx.matrix_download_csr()
    assert int(rp[-1]) == 0
    with pytest.raises(tfx.TfxError) as e:                      # lsqr_solver2.F90:150-153: v = A^T u = 0
        ctx.lsqr_solve_sensit(np.ones(xs.size), 5)
    assert "normalize" in str(e.value)
    # (2) K = 1: exactly the largest coefficient of every row survives
    res = ctx.calculate_sensit(xs, ys, zs, cw, 2, 1.0 / N + 1e-12)
    rp, cols, vals = ctx.matrix_download_csr()
    assert np.all(np.diff(rp) <= 1) and res["nnz"] == int(rp[-1]) >= xs.size - 1
    for r in range(xs.size):
        c_ref, v_ref, _ = orc.build_row_grav(grid, (nx, ny, nz), cw, (xs[r], ys[r], zs[r]), 2, 1)
        assert np.array_equal(cols[rp[r]:rp[r + 1]], c_ref)
    # (3) one datum, no compression, then compressed with rate 1 (everything above 1e-30 kept)
    for ctype, rate in ((0, 1.0), (1, 1.0)):
        res = ctx.calculate_sensit(xs[:1], ys[:1], zs[:1], cw, ctype, rate)
        rp, cols, vals = ctx.matrix_download_csr()
        c_ref, v_ref, _ = orc.build_row_grav(grid, (nx, ny, nz), cw, (xs[0], ys[0], zs[0]), ctype, N)
        assert rp.size == 2 and abs(int(rp[1]) - c_ref.size) <= 1
    # (4) a single cell
    one = [np.array([v], np.float64) for v in (0.0, 100.0, 0.0, 100.0, 0.0, 100.0)]
    ctx.set_grid(1, 1, 1, *one)
    cw1 = ctx.calculate_depth_weight(2.0, 0.0, 1.0)
    assert cw1.shape == (1,) and np.isfinite(cw1[0])
    res = ctx.calculate_sensit([50.3], [49.2], [-1.0], cw1, 1, 1.0)
    rp, cols, vals = ctx.matrix_download_csr()
    ierr, row = orc.graviprism_z(one, 50.3, 49.2, -1.0)
    assert int(rp[-1]) == 1 and cols[0] == 1 and abs(vals[0] - np.float32(row[0] * cw1[0])) <= 2 * np.spacing(np.float32(vals[0]))
    x, it, r = ctx.lsqr_solve_sensit(np.array([2.0 * float(vals[0])]), 5)
    assert abs(x[0] - 2.0) <= 1e-12


@pytest.mark.parametrize("matrix", ["reference", "built"])
def test_gradient_damping_end_to_end_vs_reference(ctx, golden_dir, matrix):
    """Gradient damping: 3 N first-difference rows built on the host go into the general constraint matrix (tfx_cons_upload_csr)
    and LSQR runs with spatial unknowns (WAVELET_DOMAIN = false: S through the per-iteration device transform) - the whole
    inversion against the reference's final model."""
    g = load(golden_dir, "e2e_dgrad")
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(*dims, *grid)
    cw = g["np1_column_weight"]
    obs = g["obs"]
    if matrix == "built":
        ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]))
    else:
        ctx.matrix_upload_csr(obs.shape[0], N, g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    # the host-side builder against the oracle's loop-by-loop restatement
    mtest = np.random.default_rng(2).standard_normal(N)
    G1, r1 = tfx.inversion.gradient_damping_rows(mtest, dims, ctx.spacing, cw, 1.0, float(g["beta"]))
    G2, r2 = oinv.gradient_damping_rows(mtest, dims, grid, cw, 1.0, float(g["beta"]))
    assert np.array_equal(G1[0], G2[0]) and np.array_equal(G1[1], G2[1]) and bits_equal(G1[2], G2[2]) and bits_equal(r1, r2)
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]), beta=float(g["beta"]))
    ref = g["np1_model_final"]
    tol = 1e-7 if matrix == "reference" else 1e-5
    assert np.linalg.norm(m - ref) <= tol * np.linalg.norm(ref), np.linalg.norm(m - ref) / np.linalg.norm(ref)
    assert np.allclose([h["r"] for h in hist], g["np1_lsqr_r"], rtol=1e-5)


def test_lp_norm_damping_end_to_end_vs_reference(ctx, golden_dir):
    """normPower = 1.5: non-constant damping block and right-hand side, spatial unknowns, D4 transform inside every LSQR
    iteration - three major iterations vs the reference's final model."""
    g = load(golden_dir, "e2e_lp")
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = g["np1_column_weight"]
    ctx.matrix_upload_csr(g["obs"].shape[0], N, g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]), norm_power=float(g["norm_power"]))
    ref = g["np1_model_final"]
    assert np.linalg.norm(m - ref) <= 1e-6 * np.linalg.norm(ref), np.linalg.norm(m - ref) / np.linalg.norm(ref)
    assert np.allclose([h["r"] for h in hist], g["np1_lsqr_r"], rtol=1e-5)


def test_admm_local_bounds_end_to_end_vs_reference(ctx, golden_dir):
    """ADMM with per-cell intervals and weights (boundType 2): non-constant ADMM block, spatial unknowns."""
    g = load(golden_dir, "e2e_admm_local")
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = g["np1_column_weight"]
    ctx.matrix_upload_csr(g["obs"].shape[0], N, g["np1_row_ptr"], g["np1_cols"], g["np1_vals"])
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]),
                                                     admm=dict(bounds=g["bounds"], weight=g["bound_weight"], rho=float(g["rho"])))
    ref = g["np1_model_final"]
    assert np.linalg.norm(m - ref) <= 1e-6 * np.linalg.norm(ref), np.linalg.norm(m - ref) / np.linalg.norm(ref)
    assert np.allclose([h["r"] for h in hist], g["np1_lsqr_r"], rtol=1e-5)


def test_data_errors_end_to_end_vs_reference(ctx, golden_dir):
    """forward.data.grav.useError: data_weight = 1 / error goes into the kernel build (row scaling), the residuals and
    calculate_data."""
    g = load(golden_dir, "e2e_err")
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = g["np1_column_weight"]
    obs = g["obs"]
    dw = 1.0 / g["data_error"]
    ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]), data_weight=dw)
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]), data_weight=dw)
    ref = g["np1_model_final"]
    assert np.linalg.norm(m - ref) <= 1e-6 * np.linalg.norm(ref), np.linalg.norm(m - ref) / np.linalg.norm(ref)
    assert np.allclose(d, g["np1_data_final"], rtol=1e-6, atol=1e-9 * np.abs(g["np1_data_final"]).max())


@pytest.mark.parametrize("name", ["e2e_localw", "e2e_localw_lp"])
def test_local_weights_end_to_end_vs_reference(ctx, golden_dir, name):
    """Local depth weights (a zero among them) and local model-damping weights: kernel, damping block, spatial unknowns; alone and
    together with an Lp norm (value = alpha * pw * Lp multiplier * local weight, one fp32 cast)."""
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    ctx.set_grid(*dims, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    lw = g["lw_depth"]
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
    cw = np.where(lw != 0.0, cw / np.where(lw != 0.0, lw, 1.0), 0.0)             # apply_local_depth_weighting
    ref_cw = g["np1_column_weight"]
    assert np.all(np.abs(cw - ref_cw) <= 1e-14 * np.abs(ref_cw)) and cw[7] == 0.0
    obs = g["obs"]
    ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, int(g["ctype"]), float(g["rate"]))
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, int(g["ctype"]), g["np1_data_observed"], int(g["nmajor"]),
                                                     int(g["nminor"]), alpha=float(g["alpha"]), damping_weight=g["lw_damp"],
                                                     norm_power=float(g["norm_power"]))
    ref = g["np1_model_final"]
    # e2e_localw: 400 iterations on 9 data rows (converged, sensitive at 5e-6); e2e_localw_lp: the well-conditioned 5-iteration
    # case (oracle 1-ulp sensitivity 2e-14), which pins the damping-block formula order
    tol = 1e-5 if name == "e2e_localw" else 1e-9
    assert np.linalg.norm(m - ref) <= tol * np.linalg.norm(ref), np.linalg.norm(m - ref) / np.linalg.norm(ref)
    assert m[7] == 0.0


@pytest.mark.parametrize("name", ["e2e_xgrad", "e2e_xgrad_cnt"])
def test_joint_inversion_with_cross_gradient_vs_reference(ctx, golden_dir, name):
    """Joint inversion with structural coupling: the host builds the cross-gradient rows over both models' columns
    (tfx_cons_upload_csr), LSQR runs over blockdiag(S_grav, S_magn) with spatial unknowns (two components through the device
    transform per iteration) - four major iterations vs the reference."""
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(*dims, *grid)
    # the vectorised builder against the oracle's loop-by-loop restatement
    rng = np.random.default_rng(4)
    t1, t2 = rng.standard_normal(N), rng.standard_normal(N)
    cwg, cwm = g["np1_grav_column_weight"], g["np1_magn_column_weight"]
    for der in (1, 2):
        A, ra, ca = tfx.inversion.cross_gradient_rows(t1, t2, dims, ctx.spacing, cwg, cwm, 0.37, der)
        B, rb, cb = oinv.cross_gradient_rows(t1, t2, dims, grid, cwg, cwm, 0.37, der)
        assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and bits_equal(A[2], B[2]) and bits_equal(ra, rb)
        assert np.allclose(ca, cb, rtol=1e-13)
    probs = []
    try:
        for i, tag in enumerate(("grav", "magn")):
            ctx.select_problem(i)
            nd = g["obs_%s" % tag].shape[0]
            ctx.matrix_upload_csr(nd, N, g["np1_%s_row_ptr" % tag], g["np1_%s_cols" % tag], g["np1_%s_vals" % tag])
            probs.append(dict(column_weight=g["np1_%s_column_weight" % tag], data_obs=g["np1_%s_data_observed" % tag], problem_weight=1.0,
                              alpha=float(g["alpha"][i])))
        ctx.select_problem(0)
        m, d, hist = tfx.inversion.solve_problem_joint(ctx, probs, int(g["ctype"]), int(g["nmajor"]), int(g["nminor"]),
                                                       cross_gradient=dict(weight=float(g["xgrad_weight"]), der_type=int(g["der_type"])))
        for i, tag in enumerate(("grav", "magn")):
            ref = g["np1_%s_model_final" % tag]
            assert np.linalg.norm(m[i] - ref) <= 1e-7 * np.linalg.norm(ref), (tag, np.linalg.norm(m[i] - ref) / np.linalg.norm(ref))
        assert np.allclose([h["r"] for h in hist], g["np1_lsqr_r"], rtol=1e-6)
        assert np.allclose(np.array([h["xgrad_cost"] for h in hist])[2:], g["np1_xgrad_cost"][2:], rtol=1e-5)
    finally:
        ctx.select_problem(1)
        ctx.matrix_free()
        ctx.select_problem(0)


@pytest.mark.parametrize("name", ["e2e_clust", "e2e_clust_normal", "e2e_clust_grav"])
def test_joint_inversion_with_clustering_vs_reference(ctx, golden_dir, name):
    """Joint inversion with the clustering (Gaussian-mixture) constraint: 2 N single-entry rows built on the host, uploaded as the
    general constraint matrix, spatial unknowns - four major iterations vs the reference."""
    g = load(golden_dir, name)
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    N = int(np.prod(dims))
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    ctx.set_grid(*dims, *grid)
    local = None if int(g["cons_type"]) == 1 else g["cell_weights"]
    cwg, cwm = g["np1_grav_column_weight"], g["np1_magn_column_weight"]
    # the vectorised builder against the oracle's cell-by-cell restatement
    rng = np.random.default_rng(5)
    t1, t2 = rng.uniform(-100, 400, N), rng.uniform(-0.01, 0.05, N)
    cellw = tfx.inversion.clustering_cell_weights(g["mixtures"], N, local)
    assert np.array_equal(cellw, oinv.clustering_setup(g["mixtures"], N, local))
    for opt in (1, 2):
        A, ra, ca = tfx.inversion.clustering_rows(t1, t2, cwg, cwm, g["clust_weight"], g["mixtures"], cellw, opt)
        B, rb, cb = oinv.clustering_rows(t1, t2, cwg, cwm, g["clust_weight"], g["mixtures"], cellw, opt)
        assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1])
        assert np.allclose(A[2], B[2], rtol=1e-6, atol=0) and np.allclose(ra, rb, rtol=1e-12, atol=1e-300) and np.allclose(ca, cb, rtol=1e-12)
    probs = []
    try:
        for i, tag in enumerate(("grav", "magn")):
            ctx.select_problem(i)
            nd = g["obs_%s" % tag].shape[0]
            ctx.matrix_upload_csr(nd, N, g["np1_%s_row_ptr" % tag], g["np1_%s_cols" % tag], g["np1_%s_vals" % tag])
            probs.append(dict(column_weight=g["np1_%s_column_weight" % tag], data_obs=g["np1_%s_data_observed" % tag], problem_weight=1.0,
                              alpha=float(g["alpha"][i])))
        ctx.select_problem(0)
        m, d, hist = tfx.inversion.solve_problem_joint(ctx, probs, int(g["ctype"]), int(g["nmajor"]), int(g["nminor"]),
                                                       clustering=dict(weight=g["clust_weight"], mixtures=g["mixtures"],
                                                                       opt_type=int(g["opt_type"]), cell_weights=local))
        for i, tag in enumerate(("grav", "magn")):
            ref = g["np1_%s_model_final" % tag]
            assert np.linalg.norm(m[i] - ref) <= 1e-7 * np.linalg.norm(ref), (tag, np.linalg.norm(m[i] - ref) / np.linalg.norm(ref))
        assert np.allclose([h["r"] for h in hist], g["np1_lsqr_r"], rtol=1e-6)
        assert np.allclose(np.array([h["clustering_cost"] for h in hist])[2:], g["np1_clust_cost"][2:], rtol=1e-5)
    finally:
        ctx.select_problem(1)
        ctx.matrix_free()
        ctx.select_problem(0)


def test_config1_mansf_end_to_end(ctx, golden_dir):
    """BASELINE config 1 (parfiles/Parfile_mansf_slice.txt: 2x128x32 cells, 256 obs, Haar 0.15, ADMM, 60 x 100 LSQR
    iterations) entirely on the HIP path vs the reference's final model."""
    g = load(golden_dir, "mansf")
    ctx.set_grid(2, 128, 32, *[g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")])
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
    obs = g["obs"]
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, 1, 0.15, want_hist=True)
    assert abs(res["nnz"] - 314368) <= 16
    assert abs(res["comp_error"] - 2.1542534704846925e-03) <= 1e-9
    built = ctx.matrix_download_csr()
    n8 = int(g["row_ptr"][-1])
    frac, maxulp = compare_built_matrix((built[0][:9], built[1], built[2]), (g["row_ptr"], g["cols"][:n8], g["vals"][:n8]), 8)
    assert frac >= 0.9999 and maxulp <= 1, (frac, maxulp)       # values within 1 fp32 ulp
    m, d, hist = tfx.inversion.solve_problem_gravity(ctx, cw, 1, g["data_observed"], 60, 100, alpha=0.0,
                                                     admm=dict(bounds=g["admm_bounds"], rho=float(g["admm_weight"])))
    ref = g["model_final"]
    rel = np.linalg.norm(m - ref) / np.linalg.norm(ref)
    assert rel <= 1e-6, rel
    assert abs(m.min() - (-19.951562372333093)) < 1e-4 and abs(m.max() - 259.9972445968676) < 1e-4
    # final data cost: SURVEY 8d asks for <= 1e-5 relative; the reference's own 1 / 2 / 4-rank runs scatter by 1.5e-6
    dcost = abs(hist[-1]["cost"] - 9.339172972115141e-11) / 9.339172972115141e-11
    print("config 1: final model rel-L2 %.3e, final data cost %.15e (relative distance %.3e; reference rank scatter 1.5e-6)" %
          (rel, hist[-1]["cost"], dcost))
    report("config1_mansf_end_to_end[python host]", model_rel_l2=rel, data_cost=hist[-1]["cost"], data_cost_rel_distance=dcost,
           reference_rank_scatter_model=4e-12, reference_rank_scatter_cost=1.5e-6, sparsity_identical=frac, max_fp32_ulp=maxulp)
    assert dcost <= 5e-6, dcost          # measured 1.9-2.0e-6 on either host


def test_medium_synthetic_build_vs_oracle_and_adjoint_identity(ctx):
    """64x64x32 cells, 8x8 obs, Haar 0.05 (SURVEY 8d generator): sampled rows vs the oracle, then the size-independent
    properties on the whole matrix: <S x, y> = <x, S^T y>, linearity, and S x against a dense re-evaluation of rows."""
    nx, ny, nz, ox, oy = 64, 64, 32, 8, 8
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    res = ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.05)
    N = nx * ny * nz
    K = int(0.05 * N)
    rp, cols, vals = ctx.matrix_download_csr()
    cw_o = orc.column_weight_type1(grid)
    for r in (0, 27, 63):
        c_ref, v_ref, _ = orc.build_row_grav(grid, (nx, ny, nz), cw_o, (xs[r], ys[r], zs[r]), 1, K)
        cb, vb = cols[rp[r]:rp[r + 1]], vals[rp[r]:rp[r + 1]]
        common, ib, ir = np.intersect1d(cb, c_ref, return_indices=True)
        assert common.size >= 0.9999 * c_ref.size
        assert np.max(np.abs(vb[ib].view(np.int32).astype(np.int64) - v_ref[ir].view(np.int32).astype(np.int64))) <= 2
    rng = np.random.default_rng(0)
    x, x2, y = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal(xs.size)
    Sx, Sx2, STy = ctx.mult_vector(x), ctx.mult_vector(x2), ctx.trans_mult_vector(y)
    scale = np.dot(np.abs(orc.spmv(rp, cols, np.abs(vals), np.abs(x))), np.abs(y))
    assert abs(np.dot(Sx, y) - np.dot(x, STy)) <= 1e-12 * scale
    assert np.allclose(ctx.mult_vector(2.0 * x - 3.0 * x2), 2.0 * Sx - 3.0 * Sx2, rtol=0, atol=1e-12 * np.abs(Sx).max() * 10)
    assert np.allclose(Sx, orc.spmv(rp, cols, vals, x), rtol=0, atol=1e-12 * np.abs(Sx).max() * 10)


@pytest.mark.parametrize("ctype,rate,expect_fallback", [(1, 0.02, False), (2, 0.1, False), (1, 0.5, False), (2, 0.9995, False),
                                                        (1, 0.5, True)])
def test_band_select_gives_the_same_matrix_as_the_full_select(ctx, ctype, rate, expect_fallback):
    """Rows of >= 2^20 cells find their threshold by the sample-bracketed band select inside the compaction's count pass; it
    must give exactly the order statistic of the full radix select: same matrix bits, nnz histogram and compression error.
    Rows that are mostly exact zeros (column weight zero almost everywhere) have the threshold under the 1e-30 floor, which
    the band cannot represent -> the batch falls back to the full select (same result)."""
    nx, ny, nz, ox, oy = 64, 64, 32, 7, 6
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    if expect_fallback:
        cw = np.where(np.arange(cw.size) % 257 == 0, cw, 0.0)
    out = []
    try:
        for min_cells in (-1, 0):
            ctx.debug_set("band_select_min_cells", min_cells)
            b0, f0 = ctx.debug_set("band_batches"), ctx.debug_set("band_fallbacks")
            res = ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, want_hist=True)
            used, fell = ctx.debug_set("band_batches") - b0, ctx.debug_set("band_fallbacks") - f0
            out.append((res, ctx.matrix_download_csr(), used, fell))
    finally:
        ctx.debug_set("band_select_min_cells", 1 << 20)
    (ra, A, used_a, _), (rb, B, used_b, fell_b) = out
    assert used_a == 0 and used_b > 0 and (fell_b > 0) == expect_fallback
    assert ra["nnz"] == rb["nnz"] and np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and bits_equal(A[2], B[2])
    assert np.array_equal(ra["nnz_hist"], rb["nnz_hist"]) and abs(ra["error_sum"] - rb["error_sum"]) <= 1e-12 * ra["error_sum"]   # summation order of the discarded energy differs
    # and in a column range with the statistics of all columns (the multi-GPU direct build)
    N = nx * ny * nz
    try:
        ctx.debug_set("band_select_min_cells", 0)
        rc = ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, col_range=(N // 3, N // 2), want_hist=True)
        C = ctx.matrix_download_csr()
    finally:
        ctx.debug_set("band_select_min_cells", 1 << 20)
    assert np.array_equal(rc["nnz_hist"], ra["nnz_hist"]) and abs(rc["error_sum"] - ra["error_sum"]) <= 1e-12 * ra["error_sum"]
    for r in (0, 17, xs.size - 1):
        sel = (A[1][A[0][r]:A[0][r + 1]] > N // 3) & (A[1][A[0][r]:A[0][r + 1]] <= N // 2)
        assert np.array_equal(C[1][C[0][r]:C[0][r + 1]], A[1][A[0][r]:A[0][r + 1]][sel] - N // 3)
        assert bits_equal(C[2][C[0][r]:C[0][r + 1]], A[2][A[0][r]:A[0][r + 1]][sel])


@pytest.mark.parametrize("expect_fallback", [False, True])
def test_overlapped_build_reads_batch_statistics_late_and_redoes_band_misses_out_of_order(ctx, expect_fallback):
    """The overlapped build (debug key "build_overlap" = 2 forces it on a build of a few batches) queues batch b + 1 before it reads
    batch b's statistics; a batch whose band missed is redone afterwards - out of order - from its kept rows with the full select.
    Same matrix bits, histogram and compression error as the one-stream build, whether no batch or every batch misses."""
    nx, ny, nz, ox, oy = 64, 64, 32, 13, 11                      # 143 observations = 5 batches of up to 32 lines
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    if expect_fallback:
        cw = np.where(np.arange(cw.size) % 257 == 0, cw, 0.0)     # thresholds under the 1e-30 floor: every band misses
    out = []
    try:
        ctx.debug_set("band_select_min_cells", 0)
        # one stream / three streams (the chain of a batch beside the wavelet passes of the next) / two streams (chain on the main stream)
        for mode, chain_late in ((0, 1), (2, 1), (2, 0)):
            ctx.debug_set("build_overlap", mode)
            ctx.debug_set("chain_under_wavelet", chain_late)
            b0, f0 = ctx.debug_set("band_batches"), ctx.debug_set("band_fallbacks")
            res = ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.5 if expect_fallback else 0.05, want_hist=True)   # (Haar r = 0.5 on 1/257 of the cells: the kept count exceeds the non-zeros)
            out.append((res, ctx.matrix_download_csr(), ctx.debug_set("band_batches") - b0, ctx.debug_set("band_fallbacks") - f0))
    finally:
        ctx.debug_set("band_select_min_cells", 1 << 20)
        ctx.debug_set("build_overlap", 1)
        ctx.debug_set("chain_under_wavelet", 1)
    ra, A, used_a, fell_a = out[0]
    for rb, B, used_b, fell_b in out[1:]:
        assert used_a == used_b >= 5 and fell_a == fell_b and (fell_b > 0) == expect_fallback
        assert ra["nnz"] == rb["nnz"] and np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and bits_equal(A[2], B[2])
        assert np.array_equal(ra["nnz_hist"], rb["nnz_hist"]) and abs(ra["error_sum"] - rb["error_sum"]) <= 1e-12 * abs(ra["error_sum"])


@pytest.mark.parametrize("kind", ["normal", "lognormal", "ties", "quantised", "mostly_zero", "denormal", "constant", "lattice", "binade", "two_runs"])
def test_band_select_on_adversarial_rows(ctx, kind):
    """Single rows with awkward value distributions through tfx_compress_row, band select against the full select and against
    numpy's order statistic: heavy ties at the threshold, few distinct values, zeros, denormals, large values on index lattices
    (what wavelet coefficients look like), a threshold on a binade boundary, long runs of values one ulp apart.  Whatever the band does (hit, overflow, miss -> fallback), the result is the same."""
    rng = np.random.default_rng(sum(kind.encode()))
    N = 300007
    if kind == "normal":
        row = rng.standard_normal(N)
    elif kind == "lognormal":
        row = np.exp(8.0 * rng.standard_normal(N)) * rng.choice([-1.0, 1.0], N)
    elif kind == "ties":
        row = rng.standard_normal(N)
        row[rng.random(N) < 0.3] = 0.731                       # 30 % of the row on one value, the threshold lands inside the run
    elif kind == "quantised":
        row = rng.integers(-6, 7, N).astype(np.float64)
    elif kind == "mostly_zero":
        row = np.where(rng.random(N) < 0.01, rng.standard_normal(N), 0.0)
    elif kind == "denormal":
        row = rng.standard_normal(N) * 1e-312
    elif kind == "constant":
        row = np.full(N, -2.5)
    elif kind == "binade":
        row = rng.uniform(0.75, 1.125, N) * rng.choice([-1.0, 1.0], N)      # K = N // 3: the threshold sits at 1.0, the band straddles the binade boundary
    elif kind == "two_runs":
        row = rng.standard_normal(N)                                       # two long runs of equal values one ulp apart around the threshold:
        m = rng.random(N)                                                  # the band select's first digit cannot separate them, the finishing
        row[m < 0.25] = 0.5                                                # block walks a long candidate list digit by digit
        row[(m >= 0.25) & (m < 0.5)] = -np.nextafter(0.5, 1.0)
    else:
        row = 1e-6 * rng.standard_normal(N)
        for step, amp in ((8, 1e-3), (64, 1.0), (512, 1e3)):
            row[::step] = amp * rng.standard_normal(row[::step].size)
    for K in (1, 700, N // 50, N // 3, N - 5):
        out = []
        try:
            for min_cells in (-1, 0):
                ctx.debug_set("band_select_min_cells", min_cells)
                out.append(ctx.compress_row(row, K))
        finally:
            ctx.debug_set("band_select_min_cells", 1 << 20)
        (ca, va, ta, da), (cb, vb, tb, db) = out
        thr_ref = max(np.partition(np.abs(row), N - K - 1)[N - K - 1], 1e-30)          # sort(|row|)[N - K] 1-based, floored (:240-256)
        assert ta == tb == thr_ref, (kind, K, ta, tb, thr_ref)
        keep = np.nonzero(np.abs(row) > thr_ref)[0]
        assert np.array_equal(ca, cb) and np.array_equal(ca, keep + 1) and bits_equal(va, vb) and bits_equal(va, row[keep].astype(np.float32))
        assert abs(da - db) <= 1e-12 * max(da, 1e-300)


def stratified_rows(ox, oy, RB=2048):
    """Rows of an ox x oy observation lattice (row = j * ox + i) spread over it: the four corners, two edge midpoints, the centre, and
    the first / last row of three interior row blocks of the device matrix (RB rows each) - 13 rows when they are all distinct."""
    D = ox * oy
    rows = [0, ox - 1, (oy - 1) * ox, D - 1,                        # corners
            ox // 2, (oy // 2) * ox,                                # midpoints of two edges
            (oy // 2) * ox + ox // 2]                               # centre
    nblk = (D + RB - 1) // RB
    for blk in sorted({max(1, nblk // 4), max(1, nblk // 2), max(1, (3 * nblk) // 4)}):
        if blk * RB < D:
            rows += [blk * RB - 1, blk * RB]                        # last row of a block, first of the next
    out = []
    for r in rows:
        if 0 <= r < D and r not in out:
            out.append(r)
    return out


FULL_SIZE = {
    # BASELINE.json configs[4] (the configuration the metric is quoted on), configs[2] and configs[1] at their full sizes
    "config5_hamersley_d4": dict(nx=256, ny=256, nz=152, ox=316, oy=316, ctype=2, rate=0.02, min_rows=12),
    "config3_haar_512": dict(nx=512, ny=512, nz=128, ox=256, oy=256, ctype=1, rate=0.01, min_rows=12),
    "config2_dense_256": dict(nx=256, ny=256, nz=64, ox=64, oy=64, ctype=0, rate=1.0, min_rows=8),
}


@pytest.mark.parametrize("name", sorted(FULL_SIZE))
def test_full_size_build_properties_and_sampled_rows_vs_oracle(ctx, name):
    """BASELINE's single-GPU configurations at FULL size (config 5: 9.96e6 cells x 99 856 data, D4 r = 0.02, nnz 1.99e10; config 3:
    3.36e7 cells x 65 536 data, Haar r = 0.01, nnz 2.2e10; config 2: 4.19e6 cells x 4096 data kept dense, 1.72e10 entries).
    The oracle cannot build such a matrix, so the check is through size-independent properties of the whole device matrix - entry
    count, adjoint identity <S x, y> = <x, S^T y>, linearity - plus rows stratified over the observation lattice (corners, edges,
    centre, first / last rows of interior row blocks: stratified_rows) pulled out with S^T e_r and compared with the oracle's rows
    (sparsity and fp32 values), one forward product entry per pulled row against the oracle's row, and an LSQR run AT THE SIZE THE
    BENCH TIMES (lsqr_solver2.F90:163-290): the residual it reports is the residual of the augmented system recomputed from the
    products, two solves give identical bits, the residual falls and the gradient of the damped least-squares functional falls."""
    # (a smaller part FAILS here with the reason instead of skipping: an unexercised configuration must not hide in a green suite)
    assert ctx.device_info()["hbm_bytes"] >= 200e9, "the full-size configurations need the 288 GB of an MI355X"
    from concurrent.futures import ThreadPoolExecutor
    c = FULL_SIZE[name]
    nx, ny, nz = c["nx"], c["ny"], c["nz"]
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, c["ox"], c["oy"])
    N, D = nx * ny * nz, xs.size
    K = int(c["rate"] * N) if c["ctype"] > 0 else N
    rows = stratified_rows(c["ox"], c["oy"])
    assert len(rows) >= c["min_rows"], rows
    # the oracle's rows (scalar C, seconds each at 1e7 cells) on host threads while the GPU builds the matrix
    cw_o = orc.column_weight_type1(grid)
    pool = ThreadPoolExecutor(max_workers=min(len(rows), max(1, (os.cpu_count() or 2) // 2)))
    futures = {r: pool.submit(orc.build_row_grav, grid, (nx, ny, nz), cw_o, (xs[r], ys[r], zs[r]), c["ctype"], K) for r in rows}
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    b0, f0 = ctx.debug_set("band_batches"), ctx.debug_set("band_fallbacks")
    rep = {"cells": N, "data": D, "rows_checked": len(rows)}
    try:
        res = ctx.calculate_sensit(xs, ys, zs, cw, c["ctype"], c["rate"])
        rep["nnz"] = int(res["nnz"])
        if c["ctype"] > 0:
            nb = ctx.debug_set("band_batches") - b0
            assert nb > 1000 and ctx.debug_set("band_fallbacks") - f0 <= 0.02 * nb    # thresholds by the band select, rare fallbacks
            assert res["nnz"] <= K * D and res["nnz"] >= 0.9999 * K * D               # every row keeps K entries, fewer only on exact ties
            assert 0.0 < res["comp_error"] < 0.1
        else:
            assert res["nnz"] == N * D                                                # nothing is dropped (no entry is |.| <= 1e-30 here)
        rng = np.random.default_rng(1)
        x, x2, y = rng.standard_normal(N), rng.standard_normal(N), rng.standard_normal(D)
        Sx, Sx2, STy = ctx.mult_vector(x), ctx.mult_vector(x2), ctx.trans_mult_vector(y)
        rep["adjoint_identity"] = float(abs(np.dot(Sx, y) - np.dot(x, STy)) / (np.linalg.norm(Sx) * np.linalg.norm(y)))
        assert rep["adjoint_identity"] <= 1e-11
        rep["linearity"] = float(np.abs(ctx.mult_vector(2.0 * x - 3.0 * x2) - (2.0 * Sx - 3.0 * Sx2)).max() / np.abs(Sx).max())
        assert rep["linearity"] <= 1e-11
        worst_scale, worst_ulp, ident, total = 0.0, 0.0, 0, 0
        for r in rows:
            e = np.zeros(D)
            e[r] = 1.0
            row = ctx.trans_mult_vector(e)                                    # row r of S, dense
            cb = np.nonzero(row)[0] + 1
            c_ref, v_ref, _ = futures[r].result()
            common, ib, ir = np.intersect1d(cb, c_ref, return_indices=True)
            assert common.size >= 0.9999 * c_ref.size and abs(cb.size - c_ref.size) <= 0.0001 * c_ref.size + 2
            vb = row[cb - 1].astype(np.float32)
            # fp32 values: within 1 ulp, or within `slack` of the row's largest entry.  The device log / atan2 and the host libm differ
            # in the last bits; the 8 corner terms of a cell (~1e5 each) cancel to ~1e-3 of their size and the transform adds up to 1e7
            # such cells into a coefficient, so the difference shows at ~1e-10 of the row scale - far below the fp32 resolution of the
            # large entries, above it for the smallest kept ones.  (Measured: compressed kernels <= 1.2e-10 of the row maximum, 86-99 %
            # of the kept values identical in fp32; the dense rows of config 2 - no transform between the prism sums and the
            # comparison, entries over seven decades - 4.3e-9.  The numbers of every run: gpurun_out/parity_report.jsonl.)
            dv = np.abs(vb[ib].astype(np.float64) - v_ref[ir].astype(np.float64))
            ulp = np.spacing(np.abs(v_ref[ir])).astype(np.float64)
            worst_row = float((dv / np.abs(v_ref).max()).max())
            worst_scale, worst_ulp = max(worst_scale, worst_row), max(worst_ulp, float((dv / ulp).max()))
            ident, total = ident + int(np.count_nonzero(dv == 0.0)), total + dv.size
            print("%s row %d: %d of %d kept values identical in fp32, worst |difference| / row maximum %.2e, worst distance %.1f fp32 ulp" %
                  (name, r, int(np.count_nonzero(dv == 0.0)), dv.size, worst_row, float((dv / ulp).max())))
            slack = 8e-9 if c["ctype"] == 0 else 5e-10
            assert np.all(dv <= 1.0 * ulp + slack * float(np.abs(v_ref).max())), worst_row
            assert abs(Sx[r] - np.dot(v_ref.astype(np.float64), x[c_ref - 1])) <= 1e-6 * np.abs(v_ref).astype(np.float64) @ np.abs(x[c_ref - 1])
        rep.update(rows=rows, worst_value_distance_over_row_scale=worst_scale, worst_fp32_ulp=worst_ulp, fp32_identical_fraction=ident / max(total, 1))

        # ---- LSQR at this size: b = S m for a smooth model, damping alpha = 1e-7 (lsqr_solver2.F90:163-290)
        m_true = 1e-3 * np.sin(np.arange(N) * (2 * np.pi / 977.0)) * np.exp(-np.arange(N) / N)
        b = ctx.mult_vector(m_true)
        alpha = np.full(N, 1e-7, np.float32)
        zeros = np.zeros(N)

        def residual(xv):
            return np.sqrt(np.sum((b - ctx.mult_vector(xv)) ** 2) + np.sum((alpha.astype(np.float64) * xv) ** 2)) / np.linalg.norm(b)

        def gradient(xv):      # of 1/2 |b - S x|^2 + 1/2 |alpha x|^2
            return np.linalg.norm(ctx.trans_mult_vector(b - ctx.mult_vector(xv)) - alpha.astype(np.float64) ** 2 * xv)

        x5, it5, r5 = ctx.lsqr_solve_sensit(b, 5, 1e-13, 0.0, 0.0, [alpha], [zeros])
        x10, it10, r10 = ctx.lsqr_solve_sensit(b, 10, 1e-13, 0.0, 0.0, [alpha], [zeros])
        x10b, it10b, r10b = ctx.lsqr_solve_sensit(b, 10, 1e-13, 0.0, 0.0, [alpha], [zeros])
        assert it5 == 5 and it10 == it10b == 10
        assert bits_equal(x10, x10b) and r10 == r10b                                  # (ii) two solves, identical bits
        r5_true, r10_true = residual(x5), residual(x10)
        rep.update(lsqr_r5=float(r5), lsqr_r10=float(r10), lsqr_r10_from_products=float(r10_true),
                   lsqr_r_rel_err=float(max(abs(r5 - r5_true) / r5_true, abs(r10 - r10_true) / r10_true)))
        assert rep["lsqr_r_rel_err"] <= 1e-10, rep                                      # (i) reported r = residual of the augmented system
        g0, g5, g10 = gradient(np.zeros(N)), gradient(x5), gradient(x10)
        rep.update(lsqr_grad0=float(g0), lsqr_grad5=float(g5), lsqr_grad10=float(g10), lsqr_bits_identical=True)
        assert r10 < r5 < 1.0 and g10 < g5 < g0, rep                                    # (iii) residual and gradient fall
        report("full_size[%s]" % name, **rep)
    finally:
        pool.shutdown(wait=True, cancel_futures=True)
        ctx.matrix_free()


def test_joint_two_kernels_at_a_single_gpu_size(ctx):
    """BASELINE config 4's system (a gravity and a magnetic kernel in one LSQR, `joint_inverse_problem.F90:547-554`) at a size one
    GPU builds in seconds: 192 x 192 x 64 cells (2.36e6), 9216 gravity + 4096 TMI data, Haar r = 0.02 (4.3e8 + 1.9e8 non-zeros).
    Per kernel: entry count, adjoint identity, two rows pulled out with S^T e_r against the oracle's rows; jointly: the residual
    LSQR reports after 25 iterations is the residual of the augmented block-diagonal system computed from the products, and the
    second block of unknowns stays exactly zero while its right-hand side is zero (tests/joint_check.py)."""
    from joint_check import joint_system_check
    out = joint_system_check(ctx, 192, 192, 64, (96, 96), (64, 64), 0.02, rows_per_kernel=2, lsqr_iters=25)
    assert [k["data"] for k in out["kernels"]] == [9216, 4096]


def test_joint_two_kernels_at_baseline_config4_size(ctx):
    """BASELINE config 4 AT ITS STATED SIZE on one GPU: 512 x 512 x 128 cells (3.36e7), 256 x 256 gravity + 256 x 256 TMI data,
    Haar r = 0.01 - two kernels of 2.2e10 non-zeros each, 124 GB each, both resident on the one MI355X (the configuration spreads
    them over 4 GPUs).  Same properties as the reduced-size test, one oracle row per kernel.  On a part with less than 270 GB the
    test FAILS with the byte budget instead of skipping: an unexercised configuration must not hide in a green suite."""
    from joint_check import joint_system_check
    hbm = ctx.device_info()["hbm_bytes"]
    assert hbm >= 270e9, "config 4 at full size needs 2 x 124 GB of matrix + 12 GB of build scratch; this device has %.0f GB" % (hbm / 1e9)
    out = joint_system_check(ctx, 512, 512, 128, (256, 256), (256, 256), 0.01, rows_per_kernel=1, lsqr_iters=10)
    assert [k["data"] for k in out["kernels"]] == [65536, 65536] and all(k["nnz"] >= 2.19e10 for k in out["kernels"])


def test_reference_named_entry_points(ctx):
    """tomofast-x_amd/host/tfx_reference_demo drives the path ONLY through the reference's own procedure names and argument
    orders (module tfx_reference_api: calculate_depth_weight, calculate_and_write_sensit, calculate_new_partitioning,
    read_sensitivity_kernel, model%calculate_data, matrix_cons%add / new_row, lsqr_solve_sensit, inverse_wavelet) with a problem
    weight of 2.5 and non-unit data weights; the Python host does the same problem through the C ABI and the oracle checks both."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tomofast-x_amd", "host", "tfx_reference_demo")
    if not os.path.isfile(exe):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout + out.stderr
    assert "Entered subroutine lsqr_solve_sensit" in out.stdout and "End of subroutine lsqr_solve_sensit" in out.stdout
    f = {k: float(v) for k, v in re.findall(r"(nnz_total|model min|max|data cost|u consumed|jinv vs direct) =\s*([-+0-9.Ee]+)", out.stdout)}
    # the same major iteration through t_joint_inversion%solve (b_RHS, damping rows, lsqr_solve_sensit, inverse transform, rescaling
    # assembled inside) gives the model of the hand-written call sequence, up to the products' run-to-run rounding
    assert f["jinv vs direct"] <= 1e-6, f["jinv vs direct"]
    nx, ny, nz = 16, 12, 8
    N = nx * ny * nz
    pw = 2.5
    grid = tfx.synthetic.grid(nx, ny, nz)
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    xs, ys, zs = tfx.synthetic.observations(nx, ny, 6, 5)
    dw = 1.0 + 0.125 * (np.arange(1, xs.size + 1) % 4)
    res = ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.1, pw, dw)             # scaling fused into the build ...
    assert int(f["nnz_total"]) == res["nnz"]
    d_obs = ctx.calc_data(ctx.forward_wavelet(tfx.synthetic.true_model(nx, ny, nz) / cw, nx, ny, nz, 1), pw, dw)
    x, it, r = ctx.lsqr_solve_sensit(pw * dw * d_obs, 20, 1e-13, 0.0, 0.0, [np.full(N, np.float32(1e-7 * pw), np.float32)], [np.zeros(N)])
    dm = ctx.inverse_wavelet(x, nx, ny, nz, 1) * cw
    assert abs(f["model min"] - dm.min()) <= 1e-6 * abs(dm.min()) and abs(f["max"] - dm.max()) <= 1e-6 * abs(dm.max())
    d_calc = ctx.calc_data(ctx.forward_wavelet(dm / cw, nx, ny, nz, 1), pw, dw)
    cost = np.linalg.norm(d_calc - d_obs) / np.linalg.norm(d_obs)
    assert abs(f["data cost"] - cost) <= 5e-2 * cost        # a 1e-7 residual of a 20-iteration solve: last-bit differences of the products show at 1e-3 of it
    assert f["u consumed"] == 0.0                                          # the right-hand side is consumed like the reference's
    # ... and the two-step form (unscaled build + row scaling on "reload") gives the fused build's matrix bit for bit
    fused = ctx.matrix_download_csr()
    ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.1)
    ctx.matrix_scale_rows(pw * dw)
    two_step = ctx.matrix_download_csr()
    assert np.array_equal(fused[0], two_step[0]) and np.array_equal(fused[1], two_step[1]) and bits_equal(fused[2], two_step[2])
    # oracle anchor: the same system on the CPU restatement
    S_o = orc.build_matrix_grav(grid, (nx, ny, nz), cw, np.stack([xs, ys, zs], 1), 1, 0.1)
    vals_o = (S_o[2] * np.repeat((pw * dw).astype(np.float32), np.diff(S_o[0]))).astype(np.float32)
    xo, ito, ro = orc.lsqr((S_o[0], S_o[1], vals_o), orc.diag_csr(np.full(N, np.float32(1e-7 * pw), np.float32)), N,
                           np.concatenate([pw * dw * d_obs, np.zeros(N)]), 20)
    dmo = orc.wavelet(xo, nx, ny, nz, 1, inverse=True) * cw
    assert abs(f["model min"] - dmo.min()) <= 1e-5 * abs(dmo.min()) and abs(f["max"] - dmo.max()) <= 1e-5 * abs(dmo.max())


def test_fortran_host_through_c_abi(ctx):
    """The Fortran host (tomofast-x_amd/host, amdflang + iso_c_binding) drives the same C ABI; its fingerprints must match
    the Python host on the same synthetic problem."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tomofast-x_amd", "host", "tfx_host_demo")
    if not os.path.isfile(exe):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout + out.stderr
    f = {k: float(v) for k, v in re.findall(r"(nnz_total|r|lsqr iters|model min|max|data cost) =\s*([-+0-9.Ee]+)", out.stdout)}
    nx, ny, nz = 16, 12, 8
    N = nx * ny * nz
    ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
    cw = ctx.calculate_depth_weight()
    xs, ys, zs = tfx.synthetic.observations(nx, ny, 6, 5)
    res = ctx.calculate_sensit(xs, ys, zs, cw, 1, 0.1)
    assert int(f["nnz_total"]) == res["nnz"]
    d_obs = ctx.calc_data(ctx.forward_wavelet(tfx.synthetic.true_model(nx, ny, nz) / cw, nx, ny, nz, 1))
    x, it, r = ctx.lsqr_solve_sensit(d_obs, 20, 1e-13, 0.0, 0.0, [np.full(N, np.float32(1e-7), np.float32)], [np.zeros(N)])
    dm = ctx.inverse_wavelet(x, nx, ny, nz, 1) * cw
    assert int(f["lsqr iters"]) == it
    assert abs(f["model min"] - dm.min()) <= 1e-6 * abs(dm.min()) and abs(f["max"] - dm.max()) <= 1e-6 * abs(dm.max())
