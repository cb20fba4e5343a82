#!/bin/bash
# Rehearsal of bench.py --gpus N on a ONE-GPU box: N ranks share GPU 0 (the start-up ladder's pre-flight sends all of them to the hook rung)
# (functional check of the multi-rank build / partition / LSQR path; the rates mean nothing - on an N-GPU node the ranks use RCCL).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
# (ranks whose LOCAL_RANK exceeds the visible GPUs share GPU 0: the start-up ladder's pre-flight sends every rank to the hook rung)
for n in 2 4 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700+n)) bench.py --gpus $n --steps 10 --warmup 2 --workload medium --no-cpu 2> /tmp/reh_$n.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n', d['n_gpus'], 'it/s', d['value'], 'build', d['build_s'], d.get('build_mode'), d['comm']['path'])"
tail -2 /tmp/reh_$n.err | cut -c1-200
done
# the headline workload on 2 ranks sharing the GPU (57 GB each)
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu 2> /tmp/reh_h.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline n', d['n_gpus'], 'it/s', d['value'], 'build', d['build_s'], d.get('build_mode'), d['comm']['path'])"
tail -2 /tmp/reh_h.err | cut -c1-200
