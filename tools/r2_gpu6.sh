#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu6
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_size or reference_named" > $O/fullsize.log 2>&1
tail -8 $O/fullsize.log
bash tools/profile_round.sh > $O/profile_round.log 2>&1
tail -30 $O/profile_round.log
