#!/bin/bash
# Round 6, GPU session 19: the proposal backward chains beside the main backward (side streams) or in line, on the final tree —
# round 6 found that overlap of throughput-bound kernels buys nothing; does the fork / join still pay? Env-only arms.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s19
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "one_side_stream:NSAMD_LEVEL_STREAMS=0" "in_line:NSAMD_SIDE_STREAM=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
