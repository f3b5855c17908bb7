#!/bin/bash
# Round 6, GPU session 33: ray terms (head layer 0's 48 per-ray inputs once per ray: nsamd_field_ray_terms + the RAYC field kernels)
# — the new equality test against the plain kernels, the field / pipeline goldens and the bench-size float64 tests, then the
# same-box A/B of the bench line with NSAMD_RAY_TERMS=0 / 1 and the per-kernel table.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s33
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ray_terms or nerfacto_field or ragged or pipeline_golden or train_step_runner or eval_render" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q -s -k "float64 or emits_the_scatter or bench_configuration" 2>&1 | grep -v "^$" | tail -25
for i in 1 2; do
  for arm in "ray_terms:" "plain:NSAMD_RAY_TERMS=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], 'bwd', j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
  done
done
echo "== per-kernel table (eager, live events)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --kernel-table 2>&1 | grep -v "^{" | head -30
} > $out/summary.txt 2>&1
cat $out/summary.txt
