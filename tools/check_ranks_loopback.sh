#!/bin/bash
# Plumbing check of bench.py's N > 1 path on a ONE-GPU box: two ranks on cuda:0 over gloo (RCCL refuses two ranks on one device), in the
# default (work sharded over the ranks), --frames-total 8 (BASELINE configs[3]), --replicate, --config 5 (split stage) and --config f3c variants.  Everything but RCCL itself runs: parameter broadcast, the
# gradient bucket, barriers, max-over-ranks timing, rank-0 report.  The rates mean nothing.   gpurun -- 'bash tools/check_ranks_loopback.sh'
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out
export D3H_DIST_BACKEND=gloo D3H_SHARE_GPU=1
P=29511
for V in "" "--frames-total 8" "--replicate" "--config 5" "--config f3c"; do
  P=$((P + 1))
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 6 --warmup 3 \
      --no-cpu-baseline --no-extras $V 2> gpurun_out/loopback.err | grep '^{"metric"' | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('variant [$V]: n_gpus', d['n_gpus'], d['scaling'], d['config']['parallelism'], '| bucket bytes', d['config']['collective']['bytes'], 'calls', d['config']['collective']['calls'])"
done
