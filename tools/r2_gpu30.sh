#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_gpu30
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fortran_host.py -x -q -m gpu -k "not full_size" > $O/t.log 2>&1; tail -3 $O/t.log
TFX_LINES_CAP=64 TFX_LINES_BYTES_LOG2=32 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > $O/t64.log 2>&1; tail -3 $O/t64.log
for cfg in "32 31" "64 32" "48 32" "32 31" "64 32"; do
set -- $cfg
TFX_BUILD_TIMING=1 TFX_LINES_CAP=$1 TFX_LINES_BYTES_LOG2=$2 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-profile 2> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap $1 log2 $2 build_s', d['build_s'], 'nnz', d['config']['nnz'])"
grep "build timing" $O/err.log
done
timeout 300 python tools/fuzz_misc.py 30 77 | tail -1
python tools/fuzz_build.py 60 103 | tail -1
cd /tmp && export TMPDIR=/tmp
TFX_BUILD_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rowgen -o rg -- python $R/tools/bench_rowgen.py > $O/rowgen.json 2> $O/rowgen.err < /dev/null
cat $O/rowgen.json | cut -c1-1200
grep prism $O/rowgen/rg_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
rm -f $O/rowgen/*kernel_trace.csv
