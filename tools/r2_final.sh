#!/bin/bash
# Round-end evidence run: GPU test suite, fuzz sweeps, the profile round, the other two BASELINE workloads.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_final
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 1200 bash tools/run_fuzz.sh ${1:-303} > $O/fuzz.log 2>&1; tail -12 $O/fuzz.log
bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
cd /tmp && export TMPDIR=/tmp
for w in haar_512 dense_256 medium small; do
  timeout 900 python $R/bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err < /dev/null; tail -c 600 $O/bench_$w.json
done
