#!/bin/bash
# GPU box: hipGraph replay with one graph per backward chain (default) against one graph per iteration and eager launches.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-ab_schedule2}; mkdir -p $O
run() {  # label, env...
  echo "== $*" | tee -a $O/ab.log
  env "${@:2}" timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $FLAGS 2>>$O/ab.err | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:90])" | tee -a $O/ab.log
}
FLAGS=
run graph-branched NSAMD_BRANCH_GRAPHS=1
run graph-single NSAMD_BRANCH_GRAPHS=0
run graph-branched NSAMD_BRANCH_GRAPHS=1
run graph-single NSAMD_BRANCH_GRAPHS=0
run graph-deferred NSAMD_DEFER_MAIN_ADAM=1
FLAGS=--no-graph
run eager NSAMD_BRANCH_GRAPHS=1
for b in 1 0; do NSAMD_BRANCH_GRAPHS=$b timeout 300 python scripts/probe_graph_variants.py 2>>$O/ab.err | tee -a $O/variants.log; done
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tee $O/pytest_training.log | tail -4
