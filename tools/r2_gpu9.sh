#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu9
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu > $O/gpu_default.log 2>&1
tail -6 $O/gpu_default.log
TFX_HYBRID=1 TFX_HYBRID_MIN_NNZ=0 TFX_HYBRID_TAU=80 timeout 2400 python -m pytest tests/ -q -m gpu -k "not full_size" > $O/gpu_hybrid.log 2>&1
tail -12 $O/gpu_hybrid.log
