#!/bin/bash
# Short form of tools/r6_final.sh for a late GPU window: the GPU suite (with the parity report), the plain bench line, the same under
# rocprofv3 --kernel-trace --stats, and the reference-order branch's validation.  ~25 minutes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_final
P=$R/gpurun_out/profile_round
mkdir -p $O $P
cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > $O/gpu_tests.log 2>&1; tail -32 $O/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $P/bench_plain.json 2> $P/bench_plain.err < /dev/null; tail -c 1200 $P/bench_plain.json
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python $R/bench.py --no-cpu > $P/bench_profiled.json 2> $P/stats.err < /dev/null
find $P -name '*kernel_trace.csv' -size +4M -delete
cd $R
bash tools/validate_reforder.sh > $O/validate_reforder.log 2>&1; tail -45 $O/validate_reforder.log
