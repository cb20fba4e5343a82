#!/bin/bash
# Refreshes the evidence under profiles/ on a GPU box: run as `gpurun -- 'bash tools/profile_round.sh'`; everything lands in
# gpurun_out/profile_round/ (copy what is to be judged into profiles/).  rocprofv3 needs cwd /tmp and TMPDIR=/tmp on this pool;
# the PMC passes are separate runs without tracing domains.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err < /dev/null
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py > $O/bench_profiled.json 2> $O/stats.err < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-include-regex k_spmv --output-format csv -d $O/pmc_$c -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc_$c.json 2> $O/pmc_$c.err < /dev/null
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rowgen -- python $R/tools/bench_rowgen.py > $O/rowgen.json 2> $O/rowgen.err < /dev/null
# keep only the small summaries (the raw kernel traces are tens of MB)
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*.csv' | head -40
tail -c 600 $O/bench_plain.json
