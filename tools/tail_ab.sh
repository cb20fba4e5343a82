#!/bin/bash
# A/B of the merged tail launch of the LSQR iteration (TFX_LSQR_MERGE_TAIL=1, the default) against three separate launches (=0) on the
# reduced workloads: `gpurun -- 'bash tools/tail_ab.sh'` -> gpurun_out/tail_ab/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tail_ab
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lsqr_tail.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for w in medium small; do
  for cfg in "0 --no-profile" "1 --no-profile" "0 --no-profile" "1 --no-profile" "0" "1"; do
    set -- $cfg
    line=$(TFX_LSQR_MERGE_TAIL=$1 timeout 600 python $R/bench.py --workload $w --no-cpu --steps 200 --warmup 20 ${2:-} 2> $O/err.log | tail -1)
    echo "$line" >> $O/bench_$w.jsonl
    python -c "
import json,sys
d=json.loads(sys.argv[1]); print('$w merge_tail=$1 ${2:-with-events}', d['value'], d['ms_per_step_runs'], d.get('final_r'))" "$line"
  done
done
