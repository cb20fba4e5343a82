#!/usr/bin/env python3
"""What ONE rank of a P-GPU run does per LSQR iteration, measured on one GPU: the headline kernel is built for the column range
rank r of P would own (the reference's nnz-balanced partition of the all-rows histogram), a world-size-1 RCCL communicator with the
collectives forced on runs the multi-rank code path (both all-reduces of every iteration are real in-stream ncclAllReduce calls),
and K iterations are timed.  1 / (ms per iteration) bounds the P-GPU rate from above (no peer latency, no imbalance); printed next to
P x the one-GPU rate it says how much of the strong-scaling loss is the path's own fixed cost per iteration.
  python tools/one_rank_share.py [workload] [P ...]      -> JSON lines"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "hamersley_1e7"
Ps = [int(v) for v in sys.argv[2:]] or [2, 4, 8]
w = bench.WORKLOADS[name]
nx, ny, nz = w["nx"], w["ny"], w["nz"]
N = nx * ny * nz
xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
D = xs.size
ctx = tfx.Context(0)
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
t0 = time.time()
res = ctx.calculate_sensit(xs, ys, zs, cw, w["ctype"], w["rate"], col_range=(0, 0), want_hist=True)
hist = res["nnz_hist"].astype(np.int32)
print(json.dumps({"histogram_pass_s": round(time.time() - t0, 2), "nnz_total": int(hist.sum())}), flush=True)
ctx.comm_init_rccl(ctx.comm_unique_id(), 0, 1)
ctx.debug_set("force_collectives", 1)
rng = np.random.default_rng(0)
d = rng.standard_normal(D)
steps = 50
for P in Ps:
    nel, nnz = tfx.sensitivity.get_load_balancing_nelements(hist, P)
    bounds = np.concatenate([[0], np.cumsum(np.asarray(nel, np.int64))])
    for r in sorted({0, P - 1}):
        c0, c1 = int(bounds[r]), int(bounds[r + 1])
        t0 = time.time()
        ctx.matrix_reserve(int(nnz[r]))               # what the all-reduced histogram says the range holds (tfx_matrix_reserve)
        got = ctx.calculate_sensit(xs, ys, zs, cw, w["ctype"], w["rate"], col_range=(c0, c1))
        t_build = time.time() - t0
        assert got["nnz"] == int(nnz[r]), (got["nnz"], nnz[r])
        ncl = c1 - c0
        ctx.lsqr_begin(d, 1e-300, 0.0, 0.0, [np.full(ncl, np.float32(1e-7), np.float32)], [np.zeros(ncl)])
        ctx.lsqr_iterate(5)
        ctx.profile_enable(True)
        ctx.timer_start()
        ctx.lsqr_iterate(steps)
        ms = ctx.timer_stop_ms() / steps
        prof = [ctx.profile_get(k) for k in range(3)]
        ctx.profile_enable(False)
        ctx.lsqr_end()
        print(json.dumps({"P": P, "rank": r, "cells": ncl, "nnz": int(nnz[r]), "share_of_nnz": round(float(nnz[r]) / float(hist.sum()), 5),
                          "ms_per_iteration": round(ms, 4), "iterations_per_s_upper_bound_at_P": round(1e3 / ms, 2),
                          "spmv_fwd_ms": round(prof[0][0] / prof[0][1], 4), "spmv_adj_ms": round(prof[1][0] / prof[1][1], 4),
                          "allreduce_ms_each_world_size_1": round(prof[2][0] / max(prof[2][1], 1), 4), "allreduces_per_iteration": prof[2][1] / steps,
                          "other_ms_per_iteration": round(ms - prof[0][0] / prof[0][1] - prof[1][0] / prof[1][1], 4),
                          "build_for_the_range_s": round(t_build, 2), "device_bytes": ctx.matrix_info()["device_bytes"]}), flush=True)
        ctx.matrix_free()
ctx.comm_destroy()
ctx.close()
