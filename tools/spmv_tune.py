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
settings = [(0, 16)]
if os.environ.get("TFX_TUNE_SETTINGS"):
    settings = [tuple(int(v) for v in s.split(":")) for s in os.environ["TFX_TUNE_SETTINGS"].split(",")]
# TFX_TUNE_KEYS="key=v1:v2,key2=v" : extra debug keys applied before each (re)build, one build per combination of the FIRST key's values
build_keys = {}
if os.environ.get("TFX_TUNE_KEYS"):
    for kv in os.environ["TFX_TUNE_KEYS"].split(","):
        k, v = kv.split("=")
        build_keys[k] = [int(x) for x in v.split(":")]
first = next(iter(build_keys), None)
for val in (build_keys[first] if first else [None]):
    for k, vs in build_keys.items():
        ctx.debug_set(k, val if k == first else vs[0])
    t0 = time.time()
    res = ctx.calculate_sensit(xs, ys, zs, cw, w["ctype"], w["rate"])
    print("%s=%s: build %.1f s nnz %d device bytes %.2f GB" % (first, val, time.time() - t0, res["nnz"],
          ctx.matrix_info()["device_bytes"] / 1e9), flush=True)
    for setting in settings:
        group, ipc = setting[0], setting[1]
        if len(setting) > 2:
            ctx.debug_set("fwd_run", setting[2])          # TFX_TUNE_SETTINGS="group:items_per_cu:fwd_run,..."
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
        rec = dict(key=first, value=val, group=group, items_per_cu=ipc, fwd_run=setting[2] if len(setting) > 2 else None, ms_per_iter=ms / steps, fwd_ms=f[0] / max(f[1], 1),
                   adj_ms=a[0] / max(a[1], 1), device_bytes=ctx.matrix_info()["device_bytes"])
        print(json.dumps(rec), flush=True)
    ctx.matrix_free()
ctx.close()
