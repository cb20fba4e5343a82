#!/usr/bin/env python3
"""BASELINE config 4 at its stated size on ONE GPU (512x512x128 cells, 65 536 gravity + 65 536 TMI data, Haar r = 0.01, two
kernels in one LSQR): builds both kernels, checks the properties of tests/joint_check.py (entry counts, adjoint identity, one
oracle row per kernel, LSQR residual against the products) and prints a JSON record with the build rates, the device bytes and
the per-iteration product times.   python tools/joint_512.py [nx ny nz ox oy rate]  > gpurun_out/r03_joint_512.json"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from joint_check import joint_system_check  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
a = sys.argv[1:]
nx, ny, nz, ox, oy = [int(v) for v in a[:5]] if len(a) >= 5 else (512, 512, 128, 256, 256)
rate = float(a[5]) if len(a) >= 6 else 0.01
ctx = tfx.Context(0)
out = joint_system_check(ctx, nx, ny, nz, (ox, oy), (ox, oy), rate, rows_per_kernel=1, lsqr_iters=20)
out["device"] = ctx.device_info()
it = out["joint_lsqr"]
it["iterations_per_s_products_only"] = 1e3 / (it["spmv_fwd_ms_per_iteration"] + it["spmv_adj_ms_per_iteration"])
print(json.dumps(out))
ctx.close()
