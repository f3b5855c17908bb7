#!/bin/bash
# Round 6, GPU session 45: the main table's apply pass deferred onto the next iteration's Adam branch (NSAMD_DEFER_APPLY=1) —
# same bits? (parameter checksums of a 20 + 300 step run), same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s45
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "defer_apply:NSAMD_DEFER_APPLY=1"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>&1 | grep '^{\|Error\|error' | python -c "
import json,sys
t=sys.stdin.read()
try:
    j=json.loads(t); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12], j['roofline']['avg_launch_ms'])
except Exception as e: print('FAILED', t[:2000])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
