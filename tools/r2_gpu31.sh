#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu31
mkdir -p $O
cd $R
time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5
time (python bench.py > $O/bench.json 2> $O/bench.err); tail -c 400 $O/bench.json; grep -c . $O/bench.json
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wavelet or two_contexts" 2>&1 | tail -2
