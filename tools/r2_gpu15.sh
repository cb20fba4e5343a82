#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu15
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fortran_host.py tests/test_gpu_multirank.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -3 $O/t.log
for ov in 1 0 1; do
TFX_BUILD_OVERLAP=$ov timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-profile 2> $O/err_$ov.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap $ov build_s', d['build_s'], 'it/s', d['value'], 'nnz', d['config']['nnz'])"
done
cd /tmp && export TMPDIR=/tmp
TFX_BUILD_OVERLAP=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ov -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-profile > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
