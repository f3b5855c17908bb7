#!/bin/bash
# Round 6, GPU session 12: (1) new tests (two ranks on one GPU, eval render of a camera, merged-launch tests after pruning);
# (2) eval render throughput: rays generated per chunk vs a prebuilt bundle; (3) the 96-sample proposal level's table scatter
# through the run-merging route (NSAMD_SCATTER_COMBINE_RES=300) vs the plain route, windows from step 40 + long run.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s12
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py tests/test_gpu_fused_launches.py -m gpu -q -s -k "two_ranks or eval_render or loss_values or merged_launches" 2>&1 | grep -E "passed|failed|^E  |two RCCL" | cut -c1-300
echo "== eval render"
for i in 1 2; do
  timeout 200 python scripts/bench_render.py 2>/dev/null | tail -1 | cut -c1-400
  timeout 200 python scripts/bench_render.py --bundle 2>/dev/null | tail -1 | cut -c1-400
done
echo "== proposal scatter route"
for i in 1 2; do
  for res in 0 300; do
    export NSAMD_SCATTER_COMBINE_RES=$res
    echo "-- NSAMD_SCATTER_COMBINE_RES=$res"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --start-step 40 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j.get('long_run'), j['config']['final_loss'])"
  done
done
unset NSAMD_SCATTER_COMBINE_RES
} > $out/summary.txt 2>&1
cat $out/summary.txt
