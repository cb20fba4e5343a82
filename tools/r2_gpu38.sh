#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu38
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -2 $O/t.log
for i in 1 2 3; do
TFX_BUILD_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build_s', d['build_s'])"; grep "build timing" $O/err.log | cut -c1-120
done
TFX_BUILD_OVERLAP=0 TFX_BUILD_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no-overlap build_s', d['build_s'])"; grep "build timing" $O/err.log | cut -c1-120
