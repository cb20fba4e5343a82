#!/usr/bin/env python3
"""Builds the headline kernel for one column range ([c0, c1) from argv) and runs five products of each kind: the target of a
`rocprofv3 --pmc ... --kernel-include-regex k_spmv` run that compares the densest and the sparsest eighth of the columns."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
c0, c1 = int(sys.argv[1]), int(sys.argv[2])
w = bench.WORKLOADS["hamersley_1e7"]
nx, ny, nz = w["nx"], w["ny"], w["nz"]
xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
ctx = tfx.Context(0)
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
res = ctx.calculate_sensit(xs, ys, zs, cw, 2, 0.02, col_range=(c0, c1))
y = np.random.default_rng(0).standard_normal(xs.size)
x = np.random.default_rng(1).standard_normal(c1 - c0)
ctx.profile_enable(True)
for _ in range(5):
    ctx.trans_mult_vector(y)
    ctx.mult_vector(x)
f, a = ctx.profile_get(0), ctx.profile_get(1)
print("range", c0, c1, "nnz", res["nnz"], "fwd ms", f[0] / f[1], "adj ms", a[0] / a[1], "stored", ctx.matrix_format()["stored_entries"])
