#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu14
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fortran_host.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -5 $O/t.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build_s', d['build_s'], 'it/s', d['value']); print(json.dumps(d.get('build_kernels', d.get('profile', {})))[:1500])"
grep -i "prism\|wavelet\|build" $O/err.log | head -20
