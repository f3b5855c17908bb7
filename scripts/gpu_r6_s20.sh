#!/bin/bash
# Round 6, GPU session 20: priority of the side stream the proposal backward chains run on (captured into the graph's branch), and
# the deferred Adam in order (NSAMD_DEFER_MAIN_ADAM=0), on the final tree. Env-only arms.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s20
mkdir -p $out
export TMPDIR=/tmp
cd $R
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > $out/summary.txt 2>&1
{
for i in 1 2 3; do
  for arm in "default:" "side_high:NSAMD_SIDE_PRIORITY=-1" "adam_in_order:NSAMD_DEFER_MAIN_ADAM=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['param_checksum']['params'][:12])"
  done
done
} >> $out/summary.txt 2>&1
cat $out/summary.txt
