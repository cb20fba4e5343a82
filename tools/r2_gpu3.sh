#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu3
mkdir -p $O
cd $R
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt; cat $O/nproc.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1
tail -15 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
