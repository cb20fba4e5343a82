"""Randomised parity sweep of the multi-component kernel build (gradiometry Gzz / full tensor, magnetic kernels with 1 | 3 data
and 1 | 3 model components), with column ranges and nnz histograms, against the CPU oracle.  Test infrastructure; GPU box."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
ctx = tfx.Context(0)
KINDS = [("gz", 1, 1, 1), ("g3", 1, 3, 1), ("gzz", 2, 1, 1), ("ftg", 2, 6, 1), ("mag", 1, 1, 1), ("mag", 1, 3, 1), ("mag", 1, 1, 3), ("mag", 1, 3, 3)]
for case in range(ncases):
    nx, ny, nz = (int(rng.integers(2, 17)) for _ in range(3))
    ex = np.concatenate([[0.0], np.cumsum(rng.uniform(20.0, 180.0, nx))])
    ey = np.concatenate([[0.0], np.cumsum(rng.uniform(20.0, 180.0, ny))])
    ez = np.concatenate([[0.0], np.cumsum(rng.uniform(20.0, 180.0, nz))])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    grid = (ex[i], ex[i + 1], ey[j], ey[j + 1], ez[k], ez[k + 1])
    if rng.random() < 0.25:
        grid = (grid[0], grid[1] - 1e-3, grid[2], grid[3] - 2e-3, grid[4], grid[5])
    N = nx * ny * nz
    nd = int(rng.integers(1, 5))
    obs = np.stack([rng.uniform(ex[0] - 50, ex[-1] + 50, nd), rng.uniform(ey[0] - 50, ey[-1] + 50, nd), -rng.uniform(0.5, 80.0, nd)], 1)
    kind, dtype, ncd, ncm = KINDS[int(rng.integers(0, len(KINDS)))]
    ctype = int(rng.integers(0, 3))
    rate = float(rng.choice([0.05, 0.2, 0.6, 1.0])) if ctype > 0 else 1.0
    if ctype > 0 and int(rate * N) == 0:
        continue
    field = (float(rng.uniform(-80, 80)), float(rng.uniform(-30, 30)), float(rng.uniform(-20, 20)), 48000.0)
    cw = orc.column_weight_type1(grid, 2.0, 0.0)
    dw = rng.uniform(0.5, 2.0, (nd, ncd))
    pw = float(rng.choice([1.0, 0.37]))
    c0 = int(rng.integers(0, N // 2 + 1))
    c1 = int(rng.integers(c0 + 1, N + 1)) if rng.random() < 0.5 else N
    if rng.random() < 0.5:
        c0, c1 = 0, N
    ctx.set_grid(nx, ny, nz, *grid)
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, ctype, rate, problem_weight=pw, data_weight=dw, col_range=(c0, c1),
                               want_hist=True, mag_field=field if kind == "mag" else None, data_type=dtype, ndata_components=ncd,
                               nmodel_components=ncm)
    built = ctx.matrix_download_csr()
    rp, cols, vals, hist, err = orc.build_matrix_comp(kind, grid, (nx, ny, nz), cw, obs, ctype, rate, field, ncm, ncd)
    assert np.abs(res["nnz_hist"].astype(np.int64) - hist).sum() <= 2 * nd * ncd * ncm, (case, "hist")
    nl = c1 - c0
    for r in range(nd * ncd):
        cr, vr = cols[rp[r]:rp[r + 1]].astype(np.int64), vals[rp[r]:rp[r + 1]]
        comp, cell = (cr - 1) // N, (cr - 1) % N
        keep = (cell >= c0) & (cell < c1)
        cr_loc = comp[keep] * nl + cell[keep] - c0 + 1
        sc = np.float32(pw * dw[r // ncd, r % ncd])
        vr_loc = (vr[keep] * sc).astype(np.float32)
        cb, vb = built[1][built[0][r]:built[0][r + 1]].astype(np.int64), built[2][built[0][r]:built[0][r + 1]]
        common, ib, ir = np.intersect1d(cb, cr_loc, return_indices=True)
        assert abs(cb.size - cr_loc.size) <= 2 * ncm and common.size >= min(cb.size, cr_loc.size) - 2 * ncm, (case, r, kind, cb.size, cr_loc.size, common.size)
        if common.size:
            scale = float(np.abs(vr_loc).max())
            dv = np.abs(vb[ib].astype(np.float64) - vr_loc[ir].astype(np.float64))
            ulp = np.spacing(np.abs(vr_loc[ir])).astype(np.float64)
            assert np.all(dv <= 2.0 * ulp + 1e-9 * scale), (case, r, kind, float((dv / scale).max()))
    print("case %2d %2dx%2dx%2d nd %d %-3s ncd %d ncm %d ctype %d rate %.2f cols [%d, %d) ok" % (case, nx, ny, nz, nd, kind, ncd, ncm, ctype, rate, c0, c1))
print("OK")
