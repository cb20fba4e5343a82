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
import csv, glob, json, collections
N = 256 * 256 * 152
out = {"grid": "256x256x152", "rows_per_launch": 26, "bytes_per_launch_read_plus_write": 2 * 26 * 8 * N, "configs": {}}
shas = {}
for cfg in "${CFGS:-0 1 2 3 4 6}".split():
    d = collections.defaultdict(list)
    for f in glob.glob("$O/trace_%s/**/*kernel_trace.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_wavelet_axis" in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("(")[0][-60:]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    per = {}
    for k, v in d.items():
        v.sort()
        # launches come in x, y, z order per batch; 2 batches of 26 rows
        per[k] = {"launches": len(v), "ms_per_axis_pass": [round(sum(x[1] for x in v[a::3]) / max(1, len(v[a::3])) / 1e6, 4) for a in range(3)]}
        per[k]["TBs_read_plus_write"] = [round(out["bytes_per_launch_read_plus_write"] / (m * 1e-3) / 1e12, 3) if m else None for m in per[k]["ms_per_axis_pass"]]
    log = open("$O/run_%s.log" % cfg).read().strip().splitlines()
    out["configs"]["wave_pipe=%s" % cfg] = {"kernels": per, "result": log[-1] if log else ""}
json.dump(out, open("$O/probe.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
