#!/bin/bash
# Round 6, GPU session 50: issue order of the captured iteration — the main branch's launches in front of the Adam branch's
# (NSAMD_ISSUE_MAIN_FIRST=1): same dependencies, which queue does the replay put the critical path on? bits + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s50
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "main_first:NSAMD_ISSUE_MAIN_FIRST=1"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
