"""Randomised sweep: the kernel built with the band select forced on against the same build with the full radix select (must be
bit-identical: matrix, nnz histogram; compression error to 1e-12) - odd grid sizes, Haar / D4, rates from 0.001 to 0.9, column
ranges, multi-component generators.  Test infrastructure; GPU box."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tfx = importlib.import_module("tomofast-x_amd")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
ctx = tfx.Context(0)
KINDS = [("gz", 1, 1, 1)] * 4 + [("gzz", 2, 1, 1), ("ftg", 2, 6, 1), ("mag", 1, 1, 1), ("mag", 1, 3, 1), ("mag", 1, 1, 3)]
fell = 0
for case in range(ncases):
    nx, ny, nz = int(rng.integers(17, 90)), int(rng.integers(17, 90)), int(rng.integers(9, 40))
    grid = tfx.synthetic.grid(nx, ny, nz, h=float(rng.uniform(20, 200)))
    N = nx * ny * nz
    nd = int(rng.integers(1, 40))
    xs = rng.uniform(0, nx * 100.0, nd) + 0.123
    ys = rng.uniform(0, ny * 100.0, nd) + 0.321
    zs = -rng.uniform(0.5, 50.0, nd)
    kind, dtype, ncd, ncm = KINDS[int(rng.integers(0, len(KINDS)))]
    ctype = int(rng.integers(1, 3))
    rate = float(rng.choice([0.001, 0.01, 0.02, 0.1, 0.5, 0.9]))
    if int(rate * N) == 0:
        continue
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight()
    if rng.random() < 0.3:
        cw = np.where(rng.random(N) < 0.5, cw, 0.0)        # half of the columns weighted out: many exact zeros
    c0, c1 = (0, N) if rng.random() < 0.6 else sorted(int(v) for v in rng.choice(N + 1, 2, replace=False))
    if c1 == c0:
        c1 = min(N, c0 + 1)
    field = (65.0, -12.0, 3.0, 52000.0)
    out = []
    try:
        for mc in (-1, 0):
            ctx.debug_set("band_select_min_cells", mc)
            f0 = ctx.debug_set("band_fallbacks")
            res = ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, col_range=(c0, c1), want_hist=True, mag_field=field if kind == "mag" else None,
                                       data_type=dtype, ndata_components=ncd, nmodel_components=ncm)
            out.append((res, ctx.matrix_download_csr(), ctx.debug_set("band_fallbacks") - f0))
    finally:
        ctx.debug_set("band_select_min_cells", 1 << 20)
    (ra, A, _), (rb, B, fb) = out
    fell += fb
    assert ra["nnz"] == rb["nnz"] and np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and A[2].tobytes() == B[2].tobytes(), (case, kind, nx, ny, nz, ctype, rate)
    assert np.array_equal(ra["nnz_hist"], rb["nnz_hist"]), (case, "hist")
    assert abs(ra["error_sum"] - rb["error_sum"]) <= 1e-12 * max(ra["error_sum"], 1e-300), (case, ra["error_sum"], rb["error_sum"])
    print("case %2d %2dx%2dx%2d (%6d) nd %2d %-3s ncd %d ncm %d ctype %d rate %.3f cols [%d, %d): nnz %d, fallbacks %d" % (
        case, nx, ny, nz, N, nd, kind, ncd, ncm, ctype, rate, c0, c1, ra["nnz"], fb))
print("OK (batches that fell back to the full select: %d)" % fell)
