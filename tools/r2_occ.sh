#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_occ
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for fg in 4 2; do
TFX_FWD_GROUP=$fg timeout 900 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex "k_spmv" --output-format csv -d $O/pmc$fg -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-profile > $O/pmc$fg.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/pmc$fg/*counter_collection.csv')
if not f: print('no counter file'); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    k=('fwd' if 'k_spmv_fwd' in r['Kernel_Name'] else 'adj', r['Counter_Name'])
    agg[k]+=float(r['Counter_Value']); n[k]+=1
print('fwd_group $fg', {k:'%.4g'%(v/ max(1,n[k])) for k,v in agg.items()})
print(rows[0].get('LDS_Block_Size'), rows[0].get('VGPR_Count'), rows[0].get('SGPR_Count'), rows[0].get('Workgroup_Size'), rows[0].get('Grid_Size'))
PY
tail -3 $O/pmc$fg.log | cut -c1-200
done
