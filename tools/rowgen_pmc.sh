#!/bin/bash
# SQ counters of the row generators (gravity and magnetic TMI) on the headline grid: `gpurun -- 'bash tools/rowgen_pmc.sh'`
# -> gpurun_out/rowgen_pmc/.  Counter passes only (no tracing domains in the same run).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/rowgen_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TFX_ROWGEN_ONLY=${TFX_ROWGEN_ONLY:-gz,mag_tmi}
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU \
    --kernel-include-regex "k_prism_gz_tensor|k_magprism_tensor" --output-format csv -d $O/p1 -- python $R/tools/bench_rowgen.py > $O/p1.json 2> $O/p1.err < /dev/null
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS \
    --kernel-include-regex "k_prism_gz_tensor|k_magprism_tensor" --output-format csv -d $O/p2 -- python $R/tools/bench_rowgen.py > $O/p2.json 2> $O/p2.err < /dev/null
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % p):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-32:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in agg:
            print(p, k, {c: "%.4g" % (v / n[(k, c)]) for c, v in agg[k].items()})
PY
find $O -name '*.csv' -size +2M -delete
