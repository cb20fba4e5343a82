#!/bin/bash
# Round-6 evidence run in ONE gpurun call (tools/r2_final.sh + what round 6 added): GPU suite with durations and the parity report, fuzz
# sweeps, the profile round (bench plain / under rocprofv3 --kernel-trace --stats / PMC passes), the other BASELINE workloads, the
# DRAM-resident CPU-baseline probe.  Afterwards, locally:  python tools/collect_profiles.py 6;  python tools/reduce_parity_report.py ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_final
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu --durations=25 > $O/gpu_tests.log 2>&1; tail -32 $O/gpu_tests.log
timeout 1200 bash tools/run_fuzz.sh ${1:-606} > $O/fuzz.log 2>&1; tail -8 $O/fuzz.log
bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
cd /tmp && export TMPDIR=/tmp
for w in haar_512 dense_256 medium small; do
  timeout 900 python $R/bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err < /dev/null; tail -c 400 $O/bench_$w.json
done
cd $R
timeout 600 python tools/cpu_baseline_large_probe.py > $O/cpu_large_probe.log 2>&1; tail -30 $O/cpu_large_probe.log
bash tools/validate_reforder.sh > $O/validate_reforder.log 2>&1; tail -45 $O/validate_reforder.log
