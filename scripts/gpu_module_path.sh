#!/bin/bash
# GPU box: the training step driven through the nn.Module / autograd interface (what the nerfacto-hip plugin runs inside
# nerfstudio's trainer), the same Model API over the explicit kernel schedule (config.fused_train_step), and the runner
# itself; eager and replayed from hipGraphs. Results: profiles/r02_module_path.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/${1:-module_path}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fused_train_step or runner_matches_autograd or camera_optimizer_gradients or runner_random_background" 2>&1 | tee $O/pytest.log | tail -15
for f in "--autograd --no-graph" "--fused-model-api --no-graph" "--no-graph" "--autograd" "--fused-model-api" ""; do
  echo "== bench.py $f" | tee -a $O/module_path.log
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline $f 2>>$O/err.log | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:40], '|', d['config']['driver'])" | tee -a $O/module_path.log
done
