#!/bin/bash
# A/B of the LSQR phase kernels (TFX_LSQR_PHASED=1, the default) against separate launches (=0) on the reduced workloads, the
# headline workload and one rank's share of a P = 8 run: `gpurun -- 'bash tools/phased_ab.sh'` -> gpurun_out/phased_ab/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/phased_ab
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lsqr_phased.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for w in small medium; do
  for q in 1 0 1n 0n; do
    p=${q%n}; extra=""; [ "$q" != "$p" ] && extra="--no-profile"       # (n: without the HIP events around the two matrix kernels)
    TFX_LSQR_PHASED=$p timeout 600 python $R/bench.py --workload $w --no-cpu --steps 200 --warmup 20 $extra 2> $O/bench_${w}_p$p.err | tail -1 >> $O/bench_${w}_p$p.jsonl
    python - <<EOF
import json
l=open("$O/bench_${w}_p$p.jsonl").read().strip().splitlines()[-1]
d=json.loads(l); print("$w phased=$q", d["value"], d["ms_per_step_runs"], d.get("final_r"))
EOF
  done
done
if [ "${1:-}" != "quick" ]; then
  for p in 1 0; do
    TFX_LSQR_PHASED=$p timeout 900 python $R/bench.py --no-cpu 2> $O/bench_headline_p$p.err | tail -1 > $O/bench_headline_p$p.json
    python -c "import json;d=json.load(open('$O/bench_headline_p$p.json'));print('headline phased=$p',d['value'],d['ms_per_step_runs'],d.get('final_r'))"
  done
  for p in 1 0; do
    TFX_LSQR_PHASED=$p timeout 900 python $R/tools/one_rank_share.py hamersley_1e7 8 > $O/one_rank_share_p$p.jsonl 2> $O/one_rank_share_p$p.err
    cut -c1-400 $O/one_rank_share_p$p.jsonl
  done
fi
