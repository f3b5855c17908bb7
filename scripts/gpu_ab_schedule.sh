#!/bin/bash
# GPU box: A/B of the N = 1 schedule switches (NSAMD_DEFER_MAIN_ADAM, NSAMD_SPLIT_REDUCE; graph replay and eager launches),
# per-variant graph times, then the training tests (bit equality of the schedules). Results: profiles/r02_schedule_ab.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-ab_schedule}; mkdir -p $O
run() {  # label, env...
  echo "== $*" | tee -a $O/ab.log
  env "${@:2}" timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $FLAGS 2>>$O/ab.err | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:60])" | tee -a $O/ab.log
}
for FLAGS in "" "--no-graph"; do for d in 1 0; do for s in 0 1; do
  run "${FLAGS:-graph}" NSAMD_DEFER_MAIN_ADAM=$d NSAMD_SPLIT_REDUCE=$s
done; done; done
for d in 1 0; do NSAMD_DEFER_MAIN_ADAM=$d timeout 300 python scripts/probe_graph_variants.py 2>>$O/ab.err | tee -a $O/variants.log; done
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tee $O/pytest_training.log | tail -4
