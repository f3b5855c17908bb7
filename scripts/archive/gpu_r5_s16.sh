#!/bin/bash
# Round 5, GPU session 16: the proposal networks' fused forward with lane pairs on one line (NSAMD_DENSITY_LANE_PAIR) —
# golden / equality tests, per-kernel table, driver-window A/B (hash forward mode 7 in both arms)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s16
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fused_launches.py -k "density or hashgrid or sampler or field_golden or samplers_golden" -q -m gpu > $out/pytest.log 2>&1
el "pytest: rc $? $(tail -1 $out/pytest.log)"
grep -E "^E  |^FAILED" $out/pytest.log | head -20
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], 'long', (j.get('long_run') or {}).get('ms_per_step'), j['config'].get('param_checksum',{}).get('params','')[:12])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 100 --profile-steps 1 --param-checksum"
{
for i in 1 2 3; do
  for lp in 0 1; do
    echo "== NSAMD_DENSITY_LANE_PAIR=$lp"; NSAMD_DENSITY_LANE_PAIR=$lp timeout 150 $B 2>/dev/null | line
  done
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
for lp in 0 1; do
  NSAMD_DENSITY_LANE_PAIR=$lp timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --windows 1 --kernel-table > /dev/null 2> $out/kernel_table_lp$lp.log
  echo "== kernel table NSAMD_DENSITY_LANE_PAIR=$lp"; grep -v amdgpu $out/kernel_table_lp$lp.log | head -12
done
el end
