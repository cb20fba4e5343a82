#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1
tail -40 $O/gpu_tests.log
