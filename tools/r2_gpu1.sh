#!/bin/bash
# round 2, GPU call 1: parity of the new matrix format + headline bench at forward group 4 / 2 / 1
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.log 2>&1
tail -5 $O/parity.log
for g in 4 2; do
  TFX_FWD_GROUP=$g timeout 600 python bench.py --no-cpu --steps 20 --warmup 3 > $O/bench_g$g.json 2> $O/bench_g$g.err
  tail -c 1500 $O/bench_g$g.json
done
