#!/bin/bash
# GPU box: the training step driven through the nn.Module / autograd interface (what the nerfacto-hip plugin runs inside
# nerfstudio's trainer) against the explicit kernel schedule, eager and replayed from hipGraphs.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/${1:-module_path}; mkdir -p $O
for f in "--autograd --no-graph" "--autograd" "--no-graph" ""; do
  echo "== bench.py $f" | tee -a $O/module_path.log
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline $f 2>>$O/err.log | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:40], '|', d['config']['driver'])" | tee -a $O/module_path.log
done
