"""Forward / adjoint product time vs matrix size (random rows, 4096 x 2^20, K entries per row): where the kernels leave the
bandwidth regime.  Timed with tfx_profile_* (HIP events around the matrix kernels)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tfx = importlib.import_module("tomofast-x_amd")
rng = np.random.default_rng(0)
nr, nc = 4096, 1 << 20
ctx = tfx.Context(0)
for K in (1024, 5000, 21000, 84000):
    base = np.sort(rng.choice(nc, K, replace=False)).astype(np.int64)
    cols = np.concatenate([np.sort((base + 7919 * r) % nc) + 1 for r in range(nr)]).astype(np.int32)
    rowptr = (np.arange(nr + 1, dtype=np.int64) * K)
    vals = rng.standard_normal(nr * K).astype(np.float32)
    ctx.matrix_upload_csr(nr, nc, rowptr, cols, vals)
    x, y = rng.standard_normal(nc), rng.standard_normal(nr)
    ctx.mult_vector(x); ctx.trans_mult_vector(y)
    ctx.profile_enable(True)
    for _ in range(20):
        ctx.mult_vector(x)
        ctx.trans_mult_vector(y)
    f, a = ctx.profile_get(0), ctx.profile_get(1)
    ctx.profile_enable(False)
    nnz = nr * K
    print("K %6d nnz %.2e  fwd %8.1f us (%5.2f TB/s stored)  adj %8.1f us (%5.2f TB/s)" % (
        K, nnz, 1e3 * f[0] / f[1], 6e-12 * nnz / (1e-3 * f[0] / f[1]), 1e3 * a[0] / a[1], 6e-12 * nnz / (1e-3 * a[0] / a[1])))
