#!/bin/bash
# HBM bytes of the three wavelet passes by PMC (separate FETCH_SIZE / WRITE_SIZE passes, no tracing domains): a 64-observation
# gravity build on the headline grid (256x256x152 = 9.96e6 cells, 80 MB per row: far beyond the caches).  -> gpurun_out/wavelet_pmc/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wavelet_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/wv_build.py <<PY
import importlib, sys
sys.path.insert(0, "$R")
tfx = importlib.import_module("tomofast-x_amd")
ctx = tfx.Context(0)
nx, ny, nz = 256, 256, 152
ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
xs, ys, zs = tfx.synthetic.observations(nx, ny, 8, 8)
cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
print(ctx.calculate_sensit(xs, ys, zs, cw, 2, 0.02)["nnz"])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "k_wavelet_axis" --output-format csv -d $O/pmc_$c -- python /tmp/wv_build.py > $O/pmc_$c.log 2>&1 < /dev/null
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /tmp/wv_build.py > $O/stats.log 2>&1 < /dev/null
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                per[(r["Dispatch_Id"], r["Grid_Size"])] += float(r["Counter_Value"])
        for (d, g), v in per.items():
            agg[g].append(v)
    out[c] = {g: {"launches": len(v), "KiB_avg": sum(v) / len(v)} for g, v in agg.items()}
st = {}
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_wavelet_axis" in r["Name"]:
            st = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
N = 256 * 256 * 152
rows = 26
out["note"] = "per launch: one axis of a batch of lines (26 or 12 rows of 9 961 472 doubles); algorithmic bytes = rows x 8 B x N read + the same written"
out["kernel_stats"] = st
json.dump(out, open("$O/wavelet_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
