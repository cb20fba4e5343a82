#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu8
mkdir -p $O
cd $R
TFX_HYBRID_MIN_NNZ=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hybrid or matrix or product or super or lsqr" > $O/t1.log 2>&1
tail -4 $O/t1.log
TFX_TUNE_TAUS=${TFX_TUNE_TAUS:-250,150,80,0} TFX_TUNE_SETTINGS=${TFX_TUNE_SETTINGS:-4:16} timeout 1500 python tools/spmv_tune.py > $O/tune.log 2>&1
cat $O/tune.log | tail -6
