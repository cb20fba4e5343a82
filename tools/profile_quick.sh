#!/bin/bash
# The plain bench line + the rocprofv3 kernel statistics of the same command (the first two steps of tools/profile_round.sh), for a
# refresh after a change that does not touch the product kernels' memory traffic: `gpurun -- 'bash tools/profile_quick.sh'`.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err < /dev/null
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu > $O/bench_profiled.json 2> $O/stats.err < /dev/null
TFX_ADJ_COPY=0 timeout 900 python $R/bench.py --no-cpu > $O/bench_plain_nocopy.json 2> $O/bench_plain_nocopy.err < /dev/null
find $O -name '*kernel_trace.csv' -size +4M -delete
tail -c 600 $O/bench_plain.json
