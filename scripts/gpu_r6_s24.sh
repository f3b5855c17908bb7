#!/bin/bash
# Round 6, GPU session 24: what the proposal backward chains cost the driver's window today — default, 36 CUs left to them,
# and none of them at all (NSAMD_DIAG_SKIP_PROP_BWD=1: wrong training, timing only; the lower bound).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s24
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "reserve36:NSAMD_BWD_RESERVE_CUS=36" "no_chains:NSAMD_DIAG_SKIP_PROP_BWD=1" "in_line:NSAMD_SIDE_STREAM=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
