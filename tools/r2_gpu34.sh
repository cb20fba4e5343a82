#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu34
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wavelet or build or prism or compress or threshold" > $O/t.log 2>&1; tail -3 $O/t.log
cd /tmp && export TMPDIR=/tmp
TFX_BUILD_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o seq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile > $O/trace.log 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r2_gpu34/trace/*kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:30]) for r in rows]
ks.sort()
i0=[i for i,k in enumerate(ks) if 'k_prism_gz' in k[2]][2000]
t0=ks[i0][0]
for k in ks[i0:i0+6]:
    print('%9.1f %9.1f %8.1f  %s'%((k[0]-t0)/1e3,(k[1]-t0)/1e3,(k[1]-k[0])/1e3,k[2]))
import statistics
w=[(k[1]-k[0])/1e3 for k in ks if 'k_wavelet_axis' in k[2]]
for ax in range(3): print('axis',ax,'median us', statistics.median(w[ax::3]))
PY
rm -f $O/trace/*kernel_trace.csv
cd $R; tail -3 $O/trace.log | cut -c1-200
