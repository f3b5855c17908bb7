#!/bin/bash
# Round 6, GPU session 21: the captured iteration's MAIN branch on a high-priority stream (side branches normal).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s21
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "main_high:NSAMD_MAIN_PRIORITY=-1"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
