"""Randomised sweeps of the smaller entry points against the CPU oracle (test infrastructure; GPU box): depth weights type 1 and
2, the nnz-balanced column partition, forward data (calc_data)."""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 41)
ctx = tfx.Context(0)
for case in range(ncases):
    nx, ny, nz = (int(rng.integers(1, 25)) for _ in range(3))
    ex = np.concatenate([[0.0], np.cumsum(rng.uniform(10.0, 300.0, nx))])
    ey = np.concatenate([[0.0], np.cumsum(rng.uniform(10.0, 300.0, ny))])
    ez = np.concatenate([[rng.uniform(0.0, 50.0)], rng.uniform(0.0, 50.0) + np.cumsum(rng.uniform(10.0, 300.0, nz))])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    grid = (ex[i], ex[i + 1], ey[j], ey[j + 1], ez[k], ez[k + 1])
    N = nx * ny * nz
    ctx.set_grid(nx, ny, nz, *grid)
    power, z0, mult = float(rng.choice([1.0, 2.0, 3.0, 2.5])), float(rng.choice([0.0, 10.0, 123.4])), float(rng.choice([1.0, 4e3]))
    a, b = ctx.calculate_depth_weight(power, z0, mult), orc.column_weight_type1(grid, power, z0, mult)
    assert np.allclose(a, b, rtol=1e-13, atol=0), ("type1", case, float(np.abs(a / b - 1).max()))
    nd = int(rng.integers(1, 30))
    obs = np.stack([rng.uniform(ex[0], ex[-1], nd), rng.uniform(ey[0], ey[-1], nd), -rng.uniform(0.5, 80.0, nd)], 1)
    beta = float(rng.choice([1.0, 1.5]))
    a, b = ctx.calculate_distance_weight(obs[:, 0], obs[:, 1], obs[:, 2], power, beta, mult), orc.column_weight_type2(grid, obs, power, beta, mult)
    assert np.allclose(a, b, rtol=1e-12, atol=0), ("type2", case, float(np.abs(a / b - 1).max()))
    a, b = ctx.calculate_mindist_weight(obs[:, 0], obs[:, 1], obs[:, 2], power, mult), orc.column_weight_type3(grid, obs, power, mult)
    assert np.allclose(a, b, rtol=1e-12, atol=0), ("type3", case, float(np.abs(a / b - 1).max()))
    # partition
    n = int(rng.integers(1, 5000))
    P = int(rng.integers(1, 9))
    hist = rng.integers(0, int(rng.choice([2, 50, 100000])), n).astype(np.int32)
    if rng.random() < 0.3:
        hist[rng.random(n) < 0.7] = 0
    if n >= P and hist.sum() > 0:
        nel, nnz = tfx.sensitivity.get_load_balancing_nelements(hist, P)
        nel_o, nnz_o = orc.partition(hist, P)
        assert np.array_equal(nel, nel_o) and np.array_equal(nnz, nnz_o), ("partition", case, nel, nel_o)
    # calc_data on a random matrix over this grid
    if N >= 2:
        nr = int(rng.integers(1, 40))
        rp, cs, vs = [0], [], []
        for r in range(nr):
            m = int(min(N, rng.poisson(6)))
            c = np.sort(rng.choice(N, m, replace=False)) if m else np.zeros(0, np.int64)
            cs.append(c.astype(np.int32) + 1)
            vs.append(rng.standard_normal(c.size).astype(np.float32))
            rp.append(rp[-1] + c.size)
        if rp[-1] > 0:
            S = (np.array(rp, np.int64), np.concatenate(cs), np.concatenate(vs))
            ctx.matrix_upload_csr(nr, N, *S)
            ctype = int(rng.integers(0, 3))
            model = rng.standard_normal(N)
            cw = np.abs(rng.standard_normal(N)) + 0.1
            cw[rng.random(N) < 0.1] = 0.0
            pw, dw = float(rng.choice([1.0, 0.25])), rng.uniform(0.5, 2.0, nr)
            scaled = np.where(cw != 0.0, model / np.where(cw != 0.0, cw, 1.0), 0.0)
            xw = ctx.forward_wavelet(scaled, nx, ny, nz, ctype) if ctype > 0 else scaled
            d = ctx.calc_data(xw, pw, dw)
            d_ref = orc.calc_data(model, cw, (nx, ny, nz), ctype, S, pw, dw)
            sc = np.abs(orc.spmv(S[0], S[1], np.abs(S[2]), np.abs(xw))) / pw / dw + 1e-300
            assert np.all(np.abs(d - d_ref) <= 1e-12 * sc), ("calc_data", case, float((np.abs(d - d_ref) / sc).max()))
print("OK (%d cases)" % ncases)
