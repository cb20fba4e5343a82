#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu36
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x -k "not full_size" > $O/t.log 2>&1; tail -3 $O/t.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fortran_host.py tests/test_gpu_multirank.py -x -q -m gpu -k "row_parallel_build_with_mpi_relayout or multi_rank_run or two_ranks_on_one_gpu" 2>&1 | tail -1; done
bash tools/r2_gpu34.sh 2>&1 | tail -12
for i in 1 2; do TFX_BUILD_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build_s', d['build_s'])"; grep "build timing" $O/err.log; done
