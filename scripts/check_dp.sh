#!/bin/bash
# GPU box (one GPU): functional check of the data-parallel schedule. Two ranks share cuda:0 over gloo and train on
# IDENTICAL rays, so the mean gradient equals the single-process gradient and the loss after K steps must equal the
# N = 1 run's — through the pipelined all-reduce schedule, eagerly (the default for N > 1) and with captured segments
# (--dp-graph; pathologically slow in this gloo / shared-GPU setup, kept as a functional check).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export NSAMD_BENCH_SAME_RAYS=1
for mode in ""; do
  one=$(python bench.py --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 1 $mode 2>/dev/null | tail -1)
  echo "N=1 $mode : $(echo $one | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["final_loss"], d["ms_per_step"], d["config"]["launch"])')"
done
for mode in "" "--dp-graph"; do
  two=$(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 10 --dist-backend gloo --share-gpu --profile-steps 1 $mode 2>/tmp/dp_err.log | tail -1)
  echo "N=2 $mode : $(echo $two | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["final_loss"], d["ms_per_step"], d["config"]["launch"])' 2>/dev/null || (echo FAILED; tail -20 /tmp/dp_err.log))"
done
