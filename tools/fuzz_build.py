"""Randomised parity sweep of the kernel build against the CPU oracle (test infrastructure; run on a GPU box):
non-uniform tensor grids of odd sizes, random observation points above the surface, Haar / D4 / none, random rates, gravity
and magnetic (TMI) kernels; compares sparsity and fp32 values row by row (ties may move +-2 entries, values within 2 ulp)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = tfx.Context(0)
worst = (1.0, 0, 0.0)
for case in range(ncases):
    nx, ny, nz = (int(rng.integers(2, 23)) for _ in range(3))
    ex = np.concatenate([[0.0], np.cumsum(rng.uniform(20.0, 180.0, nx))])
    ey = np.concatenate([[0.0], np.cumsum(rng.uniform(20.0, 180.0, ny))])
    ez = np.concatenate([[0.0], np.cumsum(rng.uniform(20.0, 180.0, nz))])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    grid = (ex[i], ex[i + 1], ey[j], ey[j + 1], ez[k], ez[k + 1])
    if rng.random() < 0.25:          # cells that do not share their faces bit for bit: the six-array generators
        grid = (grid[0], grid[1] - 1e-3, grid[2], grid[3] - 2e-3, grid[4], grid[5])
    N = nx * ny * nz
    nd = int(rng.integers(1, 7))
    obs = np.stack([rng.uniform(ex[0] - 50, ex[-1] + 50, nd), rng.uniform(ey[0] - 50, ey[-1] + 50, nd), -rng.uniform(0.5, 80.0, nd)], 1)
    ctype = int(rng.integers(0, 3))
    rate = float(rng.choice([0.03, 0.1, 0.25, 0.6, 1.0])) if ctype > 0 else 1.0
    mag = bool(rng.integers(0, 2))
    ctx.set_grid(nx, ny, nz, *grid)
    cw = orc.column_weight_type1(grid, 2.0 + float(rng.integers(0, 2)), 0.0)
    field = (float(rng.uniform(-80, 80)), float(rng.uniform(-30, 30)), 0.0, 50000.0)
    K = int(rate * N) if ctype > 0 else N
    if K == 0:
        continue
    res = ctx.calculate_sensit(obs[:, 0], obs[:, 1], obs[:, 2], cw, ctype, rate, mag_field=field if mag else None)
    built = ctx.matrix_download_csr()
    if mag:
        ref = orc.build_matrix_mag(grid, (nx, ny, nz), cw, obs, field, ctype, rate)[:3]
    else:
        ref = orc.build_matrix_grav(grid, (nx, ny, nz), cw, obs, ctype, rate)[:3]
    same, total, maxulp, maxrel = 0, 0, 0, 0.0
    for r in range(nd):
        cb, vb = built[1][built[0][r]:built[0][r + 1]], built[2][built[0][r]:built[0][r + 1]]
        cr, vr = ref[1][ref[0][r]:ref[0][r + 1]], ref[2][ref[0][r]:ref[0][r + 1]]
        assert abs(cb.size - cr.size) <= 2, (case, r, cb.size, cr.size)
        common, ib, ir = np.intersect1d(cb, cr, return_indices=True)
        same += common.size
        total += max(cb.size, cr.size)
        if common.size:
            scale = float(np.abs(vr).max())
            dv = np.abs(vb[ib].astype(np.float64) - vr[ir].astype(np.float64))
            ulp = np.spacing(np.abs(vr[ir])).astype(np.float64)
            bad = dv > 2.0 * ulp + 1e-9 * scale
            assert not bad.any(), (case, r, float((dv / scale).max()))
            maxulp = max(maxulp, int((dv / ulp).max()))
            maxrel = max(maxrel, float((dv / scale).max()))
    frac = same / max(total, 1)
    worst = (min(worst[0], frac), max(worst[1], maxulp), max(worst[2], maxrel))
    assert frac >= 0.9999, (case, frac, (nx, ny, nz), ctype, rate, mag)
    print("case %2d %2dx%2dx%2d nd %d %s ctype %d rate %.2f: sparsity %.4f max ulp %d" % (case, nx, ny, nz, nd, "mag " if mag else "grav", ctype, rate, frac, maxulp))
# the ulp distance of the smallest kept coefficients is large by construction (cancellation); what is bounded is the
# distance in units of the row's largest entry: asserted <= 1e-9 above, the run's worst is printed
print("OK: worst sparsity agreement %.6f, worst fp32-ulp distance %d, worst |difference| / row maximum %.3e (bound 1e-9 + 2 ulp)" % worst)
