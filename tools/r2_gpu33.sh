#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu33
mkdir -p $O
cd $R
TFX_BUILD_OVERLAP=0 timeout 2400 python -m pytest tests -q -m gpu -x -k "not full_size" > $O/t_nooverlap.log 2>&1; tail -2 $O/t_nooverlap.log
TFX_HYBRID=1 TFX_HYBRID_MIN_NNZ=0 TFX_HYBRID_TAU=80 timeout 2400 python -m pytest tests -q -m gpu -x -k "not full_size" > $O/t_hybrid.log 2>&1; tail -2 $O/t_hybrid.log
TFX_GEN_AFTER_WAVELET=0 TFX_GEN_WGS_PER_CU=2 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -m gpu -x -k "not full_size" > $O/t_gen.log 2>&1; tail -2 $O/t_gen.log
