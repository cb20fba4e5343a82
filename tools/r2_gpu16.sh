#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu16
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -3 $O/t.log
for w in 2 1 3 4 0; do
TFX_GEN_WGS_PER_CU=$w timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err_$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgs/cu $w build_s', d['build_s'], 'it/s', d['value'], 'nnz', d['config']['nnz'])"
done
