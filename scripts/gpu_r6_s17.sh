#!/bin/bash
# Round 6, GPU session 17: where the deferred main-field Adam's branch is forked (NSAMD_FORK_AFTER_BINS) and its width
# (NSAMD_ADAM_BLOCKS_PER_CU) on the final tree — env-only arms, alternating, window + long run.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s17
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
  for arm in "default:" "late_fork:NSAMD_FORK_AFTER_BINS=1" "late_fork_8:NSAMD_FORK_AFTER_BINS=1 NSAMD_ADAM_BLOCKS_PER_CU=8" "adam2:NSAMD_ADAM_BLOCKS_PER_CU=2"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
