#!/bin/bash
# Round 6, GPU session 41: what the main field's weight-gradient reduce costs the iteration now (rider / own launch / none at all —
# the last is wrong training, timing only: the bound of what moving it off the critical path could gain)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s41
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "rider:" "own_launch:NSAMD_REDUCE_RIDER=0" "no_reduce:NSAMD_DIAG_SKIP_DW_REDUCE=1"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
