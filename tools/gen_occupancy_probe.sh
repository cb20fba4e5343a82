#!/bin/bash
# How much the gravity row generator depends on its resident workgroups per CU (4 x 256 threads today): one-stream builds of 64 rows on
# the headline grid with the generator as a persistent grid of n workgroups per CU (TFX_GEN_WGS_PER_CU), kernel time by rocprofv3.
# A fused generator + x-lifting would run at 3 (256-thread form) or at 1 x 1024 threads: `gpurun -- 'bash tools/gen_occupancy_probe.sh'`.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gen_probe
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 0 4 3 2 1; do
  rm -rf $O/n$n
  TFX_BUILD_OVERLAP=0 TFX_GEN_WGS_PER_CU=$n TFX_ROWGEN_ONLY=gz timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$n -- python $R/tools/bench_rowgen.py > /dev/null 2>&1
  find $O/n$n -name '*kernel_trace.csv' -delete
  echo "wgs_per_cu=$n $(grep k_prism_gz_tensor $O/n$n/*/*kernel_stats.csv | awk -F, '{print "launches", $(NF-6), "avg_ns", $(NF-4)}')"
done
