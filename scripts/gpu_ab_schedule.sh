#!/bin/bash
# GPU box: A/B of the N = 1 schedule switches (NSAMD_DEFER_MAIN_ADAM, NSAMD_SPLIT_REDUCE): training tests first (bit
# equality of the deferred / split schedules with the in-order one), then bench lines and per-variant graph times.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-ab_schedule}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tee $O/pytest_training.log | tail -4
for d in 1 0; do for s in 1 0; do
  echo "== defer=$d split=$s" | tee -a $O/ab.log
  NSAMD_DEFER_MAIN_ADAM=$d NSAMD_SPLIT_REDUCE=$s timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>>$O/ab.err | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:40])" | tee -a $O/ab.log
done; done
for d in 1 0; do
  NSAMD_DEFER_MAIN_ADAM=$d NSAMD_SPLIT_REDUCE=$d timeout 300 python scripts/probe_graph_variants.py 2>>$O/ab.err | tee -a $O/variants.log
done
