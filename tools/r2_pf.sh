#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_pf
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/t.log 2>&1; tail -2 $O/t.log
TFX_HYBRID=1 TFX_HYBRID_MIN_NNZ=0 TFX_HYBRID_TAU=80 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not full_size" > $O/t_hybrid.log 2>&1; tail -2 $O/t_hybrid.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', d['value'], 'ms', d['ms_per_step'], d['roofline']['avg_launch_ms'], 'build', d['build_s'])"
done
