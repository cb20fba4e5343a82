#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu37
mkdir -p $O
cd $R
for vc in 0 1 2 3 6 13 0; do
TFX_WAVE_VCHUNK=$vc TFX_BUILD_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vchunk $vc build_s', d['build_s'])"; grep "build timing" $O/err.log | cut -c1-120
done
