#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TFX_BUILD_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o seq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile > $O/trace.log 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r2_gpu17/trace/*kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:30],r['Grid_Size_X'],r['Grid_Size_Y'],r['LDS_Block_Size'],r['VGPR_Count']) for r in rows]
ks.sort()
i0=[i for i,k in enumerate(ks) if 'k_prism_gz' in k[2]][2000]
t0=ks[i0][0]
for k in ks[i0:i0+14]:
    print('%9.1f %9.1f %8.1f  %s grid=%s,%s lds=%s vgpr=%s'%((k[0]-t0)/1e3,(k[1]-t0)/1e3,(k[1]-k[0])/1e3,k[2],k[3],k[4],k[5],k[6]))
PY
rm -f $O/trace/*kernel_trace.csv
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-include-regex "k_wavelet_axis" --output-format csv -d $O/pmc1 -o p -- python $R/bench.py --workload medium --steps 1 --warmup 1 --no-cpu --no-profile > $O/pmc1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-include-regex "k_wavelet_axis" --output-format csv -d $O/pmc2 -o p -- python $R/bench.py --workload medium --steps 1 --warmup 1 --no-cpu --no-profile > $O/pmc2.log 2>&1
python - <<'PY'
import csv,glob,os,collections
for d in ('pmc1','pmc2'):
    f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r2_gpu17/%s/*counter_collection.csv'%d)
    if not f: print('no',d); continue
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in rows:
        key=(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X','?'), r['Counter_Name'])
        agg[key]+=float(r['Counter_Value']); n[key]+=1
    for k in sorted(agg): print(d,k,'%.4g'%agg[k],n[k])
PY
