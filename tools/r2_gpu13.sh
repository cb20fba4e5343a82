#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -3 $O/t.log
TFX_CHUNK_SPAN=14 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err.log | cut -c1-300
grep -i "chunk" $O/err.log
TFX_CHUNK_SPAN=14 timeout 600 python bench.py --workload haar_512 --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err3.log | cut -c1-300
grep -i "chunk" $O/err3.log
