"""Per-iteration latency of LSQR on a launch-bound system of BASELINE config 1's size (256 x 8192, 3.1e5 non-zeros, one damping
block): time of `tfx_lsqr_solve` for 2000 iterations / 2000."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tfx = importlib.import_module("tomofast-x_amd")

rng = np.random.default_rng(0)
nr, nc, per = 256, 8192, 1228
cols = np.concatenate([np.sort(rng.choice(nc, per, replace=False)) + 1 for _ in range(nr)]).astype(np.int32)
rowptr = (np.arange(nr + 1) * per).astype(np.int64)
vals = rng.standard_normal(nr * per).astype(np.float32)
ctx = tfx.Context(0)
ctx.matrix_upload_csr(nr, nc, rowptr, cols, vals)
b = rng.standard_normal(nr)
diag = [np.full(nc, 1e-3, np.float32)]
rhs = [np.zeros(nc)]
ctx.lsqr_solve_sensit(b, 50, 0.0, 0.0, 0.0, diag, rhs)
for n in (2000, 2000):
    t0 = time.perf_counter()
    x, it, r = ctx.lsqr_solve_sensit(b, n, 0.0, 0.0, 0.0, diag, rhs)
    dt = time.perf_counter() - t0
    print("iterations %d  r %.3e  %.1f us / iteration" % (it, r, 1e6 * dt / max(it, 1)))
