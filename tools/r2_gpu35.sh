#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2 3 4 5; do
TFX_BUILD_OVERLAP=0 timeout 600 python -m pytest tests/test_gpu_fortran_host.py -x -q -m gpu -k "row_parallel_build_with_mpi_relayout" 2>&1 | grep -E "passed|failed|AssertionError:" | tr '\n' ' '; echo " [overlap 0]"
done
for i in 1 2 3 4 5; do
timeout 600 python -m pytest tests/test_gpu_fortran_host.py -x -q -m gpu -k "row_parallel_build_with_mpi_relayout" 2>&1 | grep -E "passed|failed|AssertionError:" | tr '\n' ' '; echo " [default]"
done
