#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu5
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_fortran_host.py tests/test_gpu_parity.py -q -m gpu -k "fortran or reference_named or host" > $O/gpu_tests.log 2>&1
tail -30 $O/gpu_tests.log
