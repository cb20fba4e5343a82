#!/usr/bin/env python3
"""A/B of the two product kernels between two builds of libtfx.so IN ONE PROCESS (the boxes of the pool differ by +-5 % and a box
drifts by 3 % within minutes, so only interleaved launches on the same device compare two kernels): both libraries build the same
matrix, then LSQR iterations alternate between them and the HIP-event times of the products are reported per library.
  python tools/ab_products.py <other libtfx.so> [workload] [rounds] [steps]     (the matrices are built without the adjoint copy)"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa

tfx = importlib.import_module("tomofast-x_amd")
L = importlib.import_module("tomofast-x_amd.lib")
other = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "hamersley_1e7"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
w = bench.WORKLOADS[name]
nx, ny, nz = w["nx"], w["ny"], w["nz"]
N = nx * ny * nz
xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
d = np.random.default_rng(0).standard_normal(xs.size)
diag, rhs = [np.full(N, np.float32(1e-7), np.float32)], [np.zeros(N)]


def make(path):
    keep = L._lib
    if path:
        lib = C.CDLL(path)
        lib.tfx_last_error.restype = C.c_char_p
        L._lib = lib
    ctx = tfx.Context(0)
    L._lib = keep
    ctx.debug_set("adj_copy", int(os.environ.get("TFX_AB_ADJ_COPY", "0")))
    ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
    res = ctx.calculate_sensit(xs, ys, zs, cw, w["ctype"], w["rate"])
    print("built %s: nnz %d, %.1f GB" % (path or "this tree", res["nnz"], ctx.matrix_info()["device_bytes"] / 1e9), flush=True)
    ctx.lsqr_begin(d, 1e-300, 0.0, 0.0, diag, rhs)
    ctx.lsqr_iterate(2)
    return ctx


ctxs = {"new": make(None), "old": make(other)}
out = {k: {"fwd": [], "adj": []} for k in ctxs}
for r in range(rounds):
    for k, ctx in ctxs.items():
        ctx.profile_enable(True)
        ctx.lsqr_iterate(steps)
        f, a = ctx.profile_get(0), ctx.profile_get(1)
        ctx.profile_enable(False)
        out[k]["fwd"].append(round(f[0] / max(f[1], 1), 4))
        out[k]["adj"].append(round(a[0] / max(a[1], 1), 4))
for k in out:
    out[k]["fwd_median"] = float(np.median(out[k]["fwd"]))
    out[k]["adj_median"] = float(np.median(out[k]["adj"]))
print(json.dumps(out))
