#!/bin/bash
# Round 6, GPU session 22: the persistent main backward leaves N compute units free on the iterations that run the proposal
# backward chains (NSAMD_BWD_RESERVE_CUS; 36 -> 220 workgroups = 7 sweeps of 12 288 tiles instead of 6). Env-only arms,
# alternating, three repeats: driver window, 300-step run; then the late schedule and the per-kind iteration times.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s22
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "reserve36:NSAMD_BWD_RESERVE_CUS=36" "reserve64:NSAMD_BWD_RESERVE_CUS=64" "reserve16:NSAMD_BWD_RESERVE_CUS=16"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
  done
done
for arm in "default:" "reserve36:NSAMD_BWD_RESERVE_CUS=36"; do
  name=${arm%%:*}; envs=${arm#*:}
  echo "== $name: --start-step 5000"
  env $envs timeout 300 python bench.py --steps 120 --warmup 12 --start-step 5000 --no-cpu-baseline --no-secondary --long-steps 0 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('late', j['ms_per_step'], j['value'])"
  echo "== $name: per-kind iteration times"
  env $envs timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -3
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
