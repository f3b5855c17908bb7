#!/bin/bash
# GPU box: the data-parallel schedule (pipelined exchange, compact table prefix, asynchronous all-reduce on the communication
# stream) over a ONE-rank RCCL communicator: eager segments against captured hipGraph segments (--dp-graph).
# Results: profiles/r02_schedule_ab.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/dp1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29555
for f in "--force-dp" "--force-dp --dp-graph" "--force-dp" "--force-dp --dp-graph"; do
  echo "== $f" | tee -a gpurun_out/dp1/dp.log
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $f 2>>gpurun_out/dp1/err.log | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:50])" | tee -a gpurun_out/dp1/dp.log
done
