#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export NSAMD_BENCH_SAME_RAYS=1
for i in 1 2 3; do
  one=$(python bench.py --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 1 --no-graph 2>/dev/null | tail -1)
  echo "N=1 eager : $(echo $one | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["final_loss"], d["ms_per_step"])')"
done
for i in 1 2 3; do
  two=$(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 2 --steps 40 --warmup 10 --dist-backend gloo --share-gpu --profile-steps 1 --no-graph 2>/tmp/dp_err.log | tail -1)
  echo "N=2 eager : $(echo $two | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["final_loss"], d["ms_per_step"])' 2>/dev/null || (echo FAILED; tail -20 /tmp/dp_err.log))"
done
