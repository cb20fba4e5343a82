#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix or product or super or determin or scale_rows or contexts or lsqr" > $O/parity.log 2>&1
tail -3 $O/parity.log
timeout 900 python tools/spmv_tune.py > $O/tune.log 2>&1
cat $O/tune.log
