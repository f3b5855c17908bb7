#!/bin/bash
# Round 6, GPU session 34: ray blocks of the reduce on the matrix cores — equality test, bench A/B (NSAMD_RAY_TERMS=0 / 1), kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s34
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ray_terms or nerfacto_field or ragged" 2>&1 | tail -4
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
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --kernel-table 2>&1 | grep -v "^{" | head -12
echo "== per-kernel table, plain"
NSAMD_RAY_TERMS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --kernel-table 2>&1 | grep -v "^{" | head -8
} > $out/summary.txt 2>&1
cat $out/summary.txt
