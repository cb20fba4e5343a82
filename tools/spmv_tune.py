#!/usr/bin/env python3
"""Builds the headline matrix once and times the two product kernels (HIP events inside LSQR iterations) under different
work-list knobs: forward / adjoint group (row blocks sharing the staged vectors) and work items per CU.
  python tools/spmv_tune.py [workload] [steps]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa

tfx = importlib.import_module("tomofast-x_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "hamersley_1e7"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = bench.WORKLOADS[name]
nx, ny, nz = w["nx"], w["ny"], w["nz"]
N = nx * ny * nz
xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
ctx = tfx.Context(0)
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
d = np.random.default_rng(0).standard_normal(xs.size)
diag = [np.full(N, np.float32(1e-7), np.float32)]
rhs = [np.zeros(N)]
settings = [(4, 16)]
if os.environ.get("TFX_TUNE_SETTINGS"):
    settings = [tuple(int(v) for v in s.split(":")) for s in os.environ["TFX_TUNE_SETTINGS"].split(",")]
taus = [int(v) for v in os.environ.get("TFX_TUNE_TAUS", "250").split(",")]       # 0 = purely sparse
for tau in taus:
    ctx.debug_set("hybrid", 1 if tau > 0 else 0)
    if tau > 0:
        ctx.debug_set("hybrid_tau_permille", tau)
    t0 = time.time()
    res = ctx.calculate_sensit(xs, ys, zs, cw, w["ctype"], w["rate"])
    print("tau %d: build %.1f s nnz %d head columns %d head entries %.1f %% device bytes %.2f GB" % (tau, time.time() - t0, res["nnz"],
          ctx.debug_set("head_columns"), 0.1 * ctx.debug_set("head_entries_permille"), ctx.matrix_info()["device_bytes"] / 1e9), flush=True)
    for group, ipc in settings:
        ctx.debug_set("fwd_group", group)
        ctx.debug_set("items_per_cu", ipc)
        ctx.debug_set("refinish", 0)
        ctx.lsqr_begin(d, 1e-300, 0.0, 0.0, diag, rhs)
        ctx.lsqr_iterate(2)
        ctx.profile_enable(True)
        ctx.timer_start()
        ctx.lsqr_iterate(steps)
        ms = ctx.timer_stop_ms()
        f, a = ctx.profile_get(0), ctx.profile_get(1)
        ctx.profile_enable(False)
        ctx.lsqr_end()
        rec = dict(tau=tau, group=group, items_per_cu=ipc, ms_per_iter=ms / steps, fwd_ms=f[0] / max(f[1], 1), adj_ms=a[0] / max(a[1], 1),
                   device_bytes=ctx.matrix_info()["device_bytes"])
        print(json.dumps(rec), flush=True)
    ctx.matrix_free()
ctx.close()
