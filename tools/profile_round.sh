#!/bin/bash
# Refreshes the evidence under profiles/ on a GPU box: run as `gpurun -- 'bash tools/profile_round.sh'`; everything lands in
# gpurun_out/profile_round/ (copy what is to be judged into profiles/).  rocprofv3 needs cwd /tmp and TMPDIR=/tmp on this pool;
# the PMC passes are separate runs without tracing domains (kernel-trace / stats only in their own run).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profile_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err < /dev/null
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu > $O/bench_profiled.json 2> $O/stats.err < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-include-regex k_spmv --output-format csv -d $O/pmc_$c -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc_$c.json 2> $O/pmc_$c.err < /dev/null
done
# SQ counters of the two product kernels (two passes of <= 8 SQ counters)
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU \
    --kernel-include-regex k_spmv --output-format csv -d $O/pmc_SQ1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc_SQ1.json 2> $O/pmc_SQ1.err < /dev/null
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
    --kernel-include-regex k_spmv --output-format csv -d $O/pmc_SQ2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc_SQ2.json 2> $O/pmc_SQ2.err < /dev/null
# the same without the transposed copy (TFX_ADJ_COPY=0): the adjoint on the tiles of S, k_spmv_adj with exact integer accumulation
TFX_ADJ_COPY=0 timeout 900 python $R/bench.py --no-cpu > $O/bench_plain_nocopy.json 2> $O/bench_plain_nocopy.err < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  TFX_ADJ_COPY=0 timeout 900 rocprofv3 --pmc $c --kernel-include-regex k_spmv --output-format csv -d $O/pmc0_$c -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc0_$c.json 2> $O/pmc0_$c.err < /dev/null
done
TFX_ADJ_COPY=0 timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU \
    --kernel-include-regex k_spmv --output-format csv -d $O/pmc0_SQ1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc0_SQ1.json 2> $O/pmc0_SQ1.err < /dev/null
TFX_ADJ_COPY=0 timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
    --kernel-include-regex k_spmv --output-format csv -d $O/pmc0_SQ2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc0_SQ2.json 2> $O/pmc0_SQ2.err < /dev/null
python $R/tools/pmc_reduce.py $O $O/pmc_summary.json 4 pmc bench_plain.json > $O/pmc_reduce.log 2>&1
python $R/tools/pmc_reduce.py $O $O/nocopy_pmc_summary.json 4 pmc0 bench_plain_nocopy.json >> $O/pmc_reduce.log 2>&1
# SQ counters of the two dominant build kernels (a reduced build: the 'medium' workload has the same kernels, fewer launches)
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY \
    --kernel-include-regex "k_prism_gz_tensor|k_wavelet_axis" --output-format csv -d $O/pmc_build -- \
    python $R/bench.py --workload medium --steps 3 --warmup 1 --no-cpu --no-profile > $O/pmc_build.json 2> $O/pmc_build.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rowgen -- python $R/tools/bench_rowgen.py > $O/rowgen.json 2> $O/rowgen.err < /dev/null
# keep only the small summaries (the raw kernel traces are tens of MB)
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*.csv' | head -60
tail -c 1500 $O/bench_plain.json
