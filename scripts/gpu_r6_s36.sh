#!/bin/bash
# Round 6, GPU session 36: ray terms on the Adam branch; the per-ray products (weight-gradient columns, appearance rows) inside the backward kernel — tests, then the same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s36
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ray_terms or nerfacto_field or ragged or fused_train_step or trajectory" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q -k "bit_repro or graph_replay or float64" 2>&1 | tail -4
for i in 1 2 3; do
  for arm in "ray_terms:" "terms_in_line:NSAMD_TERMS_ON_BRANCH=0" "plain:NSAMD_RAY_TERMS=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], 'bwd', j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
  done
done
echo "== per-kernel table (eager, live events)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --kernel-table 2>&1 | grep -v "^{" | grep "dw_reduce\|ray_terms\|gradients\|mlp_fwd"
} > $out/summary.txt 2>&1
cat $out/summary.txt
