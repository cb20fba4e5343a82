"""Per-iteration latency of LSQR on a launch-bound system of BASELINE config 1's size (256 x 8192, 3.1e5 non-zeros, one damping
block): (time of a 56-iteration solve - time of an 8-iteration solve) / 48."""
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


def timed(n, reps=20):
    t0 = time.perf_counter()
    for _ in range(reps):
        x, it, r = ctx.lsqr_solve_sensit(b, n, 0.0, 0.0, 0.0, diag, rhs)
    assert it == n
    return (time.perf_counter() - t0) / reps


for _ in range(2):
    t_a, t_b = timed(8), timed(56)
    print("solve(8) %.0f us, solve(56) %.0f us -> %.1f us / iteration" % (1e6 * t_a, 1e6 * t_b, 1e6 * (t_b - t_a) / 48))
