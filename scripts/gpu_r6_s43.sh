#!/bin/bash
# Round 6, GPU session 43: one-rank data-parallel rehearsal — the pending main-field update on its own stream (NSAMD_DP_UPDATE_STREAM=1),
# now that the eager path is no longer host-bound; per-segment host-synchronous times
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s43
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
  for arm in "graph:" "force_dp:--force-dp" "force_dp_update_stream:--force-dp:NSAMD_DP_UPDATE_STREAM=1"; do
    name=${arm%%:*}; rest=${arm#*:}; flags=${rest%%:*}; envs=""; [ "$rest" != "$flags" ] && envs=${rest#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 $flags 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['launch'][:40])"
  done
done
echo "== segment times (host-synchronous)"
NSAMD_DP_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --force-dp 2>&1 | grep "dp-timing"
} > $out/summary.txt 2>&1
cat $out/summary.txt
