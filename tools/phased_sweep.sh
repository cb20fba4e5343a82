#!/bin/bash
# Sweep of the phase kernels' grid cap and poll interval against separate launches on the reduced workloads -> gpurun_out/phased_sweep/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/phased_sweep
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lsqr_phased.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for w in medium small; do
  for cfg in "0 0 1" "1 0 1" "1 512 1" "1 256 1" "1 128 1" "1 256 0" "1 256 4" "1 64 1" "0 0 1"; do
    set -- $cfg
    line=$(TFX_LSQR_PHASED=$1 TFX_LSQR_PHASE_GRID=$2 TFX_LSQR_PHASE_NAP=$3 timeout 600 python $R/bench.py --workload $w --no-cpu --steps 200 --warmup 20 --no-profile 2> $O/err.log | tail -1)
    echo "$line" >> $O/bench_$w.jsonl
    python -c "
import json,sys
d=json.loads(sys.argv[1]); print('$w phased=$1 grid=$2 nap=$3', d['value'], d['ms_per_step_runs'], d.get('final_r'))" "$line"
  done
done
