#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -3 $O/t.log
for xt in "16,16,16" "8,16,16" "4,16,16" "8,16,32" "8,8,16" "16,16,16" "8,16,16"; do
  TFX_WAVE_XT=$xt timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err_$xt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('XT $xt build_s', d['build_s'])"
done
