#!/bin/bash
# the software-pipelined wavelet pass (k_wavelet_axis_pipe, debug key wave_pipe = resident workgroups per
# CU) against the one-workgroup-per-tile form (k_wavelet_axis), measured ALONE: a sequential (TFX_BUILD_OVERLAP=0) D4 build of 52
# observations = 2 batches of 26 rows on the headline grid (256 x 256 x 152 = 9.96e6 cells, 2.07 GB per batch and axis pass read + the
# same written), per-launch durations from rocprofv3 --kernel-trace.  -> gpurun_out/wavelet_pipe/probe.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wavelet_pipe
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/wv_build.py <<PY
import importlib, sys, hashlib
import numpy as np
sys.path.insert(0, "$R")
tfx = importlib.import_module("tomofast-x_amd")
ctx = tfx.Context(0)
nx, ny, nz = 256, 256, 152
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
xs, ys, zs = tfx.synthetic.observations(nx, ny, 13, 4)
cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
res = ctx.calculate_sensit(xs, ys, zs, cw, 2, 0.02)
rp, cols, vals = ctx.matrix_download_csr()
print("nnz", res["nnz"], "sha", hashlib.sha256(cols.tobytes() + vals.tobytes()).hexdigest()[:16])
PY
for cfg in ${CFGS:-0 1 2 3 4 6}; do
  rm -rf $O/trace_$cfg
  TFX_WAVE_PIPE=$cfg TFX_BUILD_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$cfg -- python /tmp/wv_build.py > $O/run_$cfg.log 2>&1 < /dev/null
  tail -1 $O/run_$cfg.log
done
python - <<PY
import csv, glob, json
N = 256 * 256 * 152
B = 2 * 26 * 8 * N
out = {"what": "k_wavelet_axis (one workgroup per tile) vs k_wavelet_axis_pipe (persistent, next tile's loads in flight over the lifting) ALONE: "
               "sequential D4 build of 2 x 26 rows on 256x256x152 cells, rocprofv3 --kernel-trace per-launch durations; second batch (warm) reported",
       "bytes_per_launch_read_plus_write": B, "configs": []}
for cfg in "${CFGS:-0 1 2 3 4 6}".split():
    L = []
    for f in glob.glob("$O/trace_%s/**/*kernel_trace.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_wavelet_axis" in r["Kernel_Name"]:
                L.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"].split("(")[0].replace("void tfx::", "")))
    L.sort()          # x, y, z of batch 1, then of batch 2
    log = [l for l in open("$O/run_%s.log" % cfg).read().splitlines() if l.startswith("nnz")]
    out["configs"].append({"wave_pipe_workgroups_per_cu": int(cfg), "kernel": sorted({x[2] for x in L}), "launches": len(L),
                           "ms_x_y_z_first_batch": [round(x[1], 4) for x in L[:3]], "ms_x_y_z_second_batch": [round(x[1], 4) for x in L[3:6]],
                           "TBs_read_plus_write_second_batch": [round(B / (x[1] * 1e-3) / 1e12, 2) for x in L[3:6]], "matrix": log[-1] if log else ""})
json.dump(out, open("$O/probe.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
