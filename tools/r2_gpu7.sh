#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hybrid or matrix or product or super" > $O/t1.log 2>&1
tail -15 $O/t1.log
TFX_HYBRID_MIN_NNZ=0 timeout 1800 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "not full_size" > $O/t2.log 2>&1
tail -15 $O/t2.log
timeout 900 python tools/spmv_tune.py > $O/tune.log 2>&1
cat $O/tune.log | tail -12
