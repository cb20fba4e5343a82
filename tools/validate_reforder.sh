#!/bin/bash
# Validates the `reference-order` development branch on a GPU box WITHOUT touching the tree under test: copies the repository to /tmp,
# applies tools/reforder.patch (git diff main reference-order), rebuilds libtfx.so there and runs the branch's GPU tests.
# -> gpurun_out/reforder/{build.log,tests.log,parity_report/}
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/reforder
mkdir -p $O
rm -rf /tmp/reforder_tree && cp -a $R/. /tmp/reforder_tree && cd /tmp/reforder_tree || exit 1
rm -rf gpurun_out
patch -p1 < tools/reforder.patch > $O/patch.log 2>&1 || { tail -5 $O/patch.log; exit 1; }
make -C tomofast-x_amd/csrc > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -q -s \
    -k "reference_order or lsqr_vs_reference_golden or lsqr_reference_known or hamersley_joint" > $O/tests.log 2>&1
tail -40 $O/tests.log
mkdir -p $O/parity_report && cp -f gpurun_out/parity_report/*.jsonl $O/parity_report/ 2>/dev/null
