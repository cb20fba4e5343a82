#!/usr/bin/env python3
"""The two ways to run b += S^T u on ONE matrix in ONE process: the forward kernel on the transposed copy, then (after the copy is
dropped) the integer-accumulating adjoint kernel on the tiles of S.  HIP-event times per launch, LSQR iterations; also the distance
between the two adjoint results and the adjoint identity of each.
  python tools/ab_adjoint.py [workload] [rounds] [steps]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa

tfx = importlib.import_module("tomofast-x_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "hamersley_1e7"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
w = bench.WORKLOADS[name]
nx, ny, nz = w["nx"], w["ny"], w["nz"]
N = nx * ny * nz
xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
ctx = tfx.Context(0)
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
res = ctx.calculate_sensit(xs, ys, zs, cw, w["ctype"], w["rate"])
assert ctx.debug_set("has_adj_copy") == 1, "the copy did not fit"
rng = np.random.default_rng(0)
d = rng.standard_normal(xs.size)
x, y = rng.standard_normal(N), rng.standard_normal(xs.size)
diag, rhs = [np.full(N, np.float32(1e-7), np.float32)], [np.zeros(N)]
out = {"workload": name, "nnz": int(res["nnz"])}
Sx = ctx.mult_vector(x)
at = {}
for mode in ("copy", "tiles"):
    if mode == "tiles":
        ctx.debug_set("drop_adj_copy", 0)
        assert ctx.debug_set("has_adj_copy") == 0
    at[mode] = ctx.trans_mult_vector(y)
    ident = abs(np.dot(Sx, y) - np.dot(x, at[mode])) / (np.linalg.norm(Sx) * np.linalg.norm(y))
    ctx.lsqr_begin(d, 1e-300, 0.0, 0.0, diag, rhs)
    ctx.lsqr_iterate(2)
    f, a, it = [], [], []
    for _ in range(rounds):
        ctx.profile_enable(True)
        ctx.timer_start()
        ctx.lsqr_iterate(steps)
        ms = ctx.timer_stop_ms()
        pf, pa = ctx.profile_get(0), ctx.profile_get(1)
        ctx.profile_enable(False)
        f.append(round(pf[0] / pf[1], 4)); a.append(round(pa[0] / pa[1], 4)); it.append(round(ms / steps, 4))
    ctx.lsqr_end()
    out[mode] = {"fwd_ms": f, "adj_ms": a, "ms_per_iteration": it, "adjoint_identity": float(ident),
                 "device_bytes": ctx.matrix_info()["device_bytes"], "repeat_is_bit_identical": bool(np.array_equal(at[mode], ctx.trans_mult_vector(y)))}
out["adjoint_results_rel_l2_distance"] = float(np.linalg.norm(at["copy"] - at["tiles"]) / np.linalg.norm(at["copy"]))
out["adjoint_results_max_abs_distance_over_max_abs"] = float(np.abs(at["copy"] - at["tiles"]).max() / np.abs(at["copy"]).max())
print(json.dumps(out))
ctx.close()
