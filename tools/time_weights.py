#!/usr/bin/env python3
"""Times the distance weighting (forward.depthWeighting.type = 2, the reference's default: weights_gravmag.f90:81-138) at the
headline size: 9.96e6 cells x 99 856 data = 9.9e11 (cell, datum) pairs, 8 sub-points each.
  python tools/time_weights.py [workload] [ndata_sample ...]     -> JSON lines (pairs/s, sub-point evaluations/s)"""
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
w = bench.WORKLOADS[name]
nx, ny, nz = w["nx"], w["ny"], w["nz"]
N = nx * ny * nz
xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
samples = [int(v) for v in sys.argv[2:]] or [1024, xs.size]
ctx = tfx.Context(0)
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
for nd in samples:
    idx = np.linspace(0, xs.size - 1, nd).astype(int)
    t0 = time.time()
    cw = ctx.calculate_distance_weight(xs[idx], ys[idx], zs[idx], 2.0, 1.0, 4.0e3)
    dt = time.time() - t0
    print(json.dumps({"workload": name, "cells": N, "data": nd, "seconds": round(dt, 3), "pairs_per_s": N * nd / dt,
                      "subpoint_evaluations_per_s": 8.0 * N * nd / dt, "weight_min": float(cw.min()), "weight_max": float(cw.max()),
                      "checksum": float(np.sum(cw))}), flush=True)
ctx.close()
