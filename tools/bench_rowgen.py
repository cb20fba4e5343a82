"""Times the build of a small batch of observations on a headline-size grid for every row generator
(run under `rocprofv3 --kernel-trace --stats` to get the per-kernel split).  Prints cell.obs pairs per second of the whole
build (generator + wavelet + select + compaction) per generator."""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

tfx = importlib.import_module("tomofast-x_amd")


def main():
    nx, ny, nz = (256, 256, 152) if len(sys.argv) < 4 else [int(v) for v in sys.argv[1:4]]
    nobs = 64
    ctx = tfx.Context(0)
    grid = tfx.synthetic.grid(nx, ny, nz)
    ctx.set_grid(nx, ny, nz, *grid)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, 8, 8)
    cw = ctx.calculate_depth_weight(2.0, 0.0, 1.0)
    field = (-62.0, 11.0, 20.0, 57000.0)
    cases = [
        ("gz", dict()),
        ("gzz", dict(data_type=2)),
        ("ftg", dict(data_type=2, ndata_components=6)),
        ("mag_tmi", dict(mag_field=field)),
        ("mag_1x3", dict(mag_field=field, ndata_components=3)),
        ("mag_3x1", dict(mag_field=field, nmodel_components=3)),
        ("mag_3x3", dict(mag_field=field, ndata_components=3, nmodel_components=3)),
    ]
    out = {}
    only = os.environ.get("TFX_ROWGEN_ONLY")          # e.g. "gz": one generator only (tools/gen_occupancy_probe.sh)
    for name, kw in cases:
        if only and name not in only.split(","):
            continue
        t0 = time.time()
        res = ctx.calculate_sensit(xs[:nobs], ys[:nobs], zs[:nobs], cw, 2, 0.02, **kw)
        dt = time.time() - t0
        nl = kw.get("ndata_components", 1) * kw.get("nmodel_components", 1)
        out[name] = dict(seconds=round(dt, 3), pairs_per_s=nx * ny * nz * nobs / dt, lines_per_obs=nl, nnz=res["nnz"])
        print("[rowgen] %-8s %7.3f s  %.3e pairs/s  (%d lines per observation)" % (name, dt, out[name]["pairs_per_s"], nl), file=sys.stderr)
    print(json.dumps(dict(grid=[nx, ny, nz], nobs=nobs, results=out)))


if __name__ == "__main__":
    main()
