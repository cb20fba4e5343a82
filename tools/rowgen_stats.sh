#!/bin/bash
# per-kernel times of the row generators on the headline grid: `gpurun -- 'bash tools/rowgen_stats.sh [tag]'` -> gpurun_out/rowgen_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/rowgen_${1:-x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/bench_rowgen.py > $O/rowgen.json 2> $O/rowgen.err < /dev/null
find $O -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for f in glob.glob("$O/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "prism" in n and "fix" not in n:
            print("  %-40s calls %3s avg %.3f ms" % (n.split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
