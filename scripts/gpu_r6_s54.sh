#!/bin/bash
# Round 6, GPU session 54: root-select schedule (NSAMD_SELECT_ROOT=1: the launch that selects the batch and writes the initial bins a
# root of the captured iteration beside the prologue) — same bits? same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s54
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "select_root:NSAMD_SELECT_ROOT=1"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>&1 | grep '^{\|Error' | python -c "
import json,sys
t=sys.stdin.read()
try:
    j=json.loads(t); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12], j['config']['launch'][:30])
except Exception as e: print('FAILED', t[-1500:])"
  done
done
echo "== eager arms (bits)"
for arm in "default:" "select_root:NSAMD_SELECT_ROOT=1"; do
  name=${arm%%:*}; envs=${arm#*:}
  env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --param-checksum --no-graph 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$name eager window', j['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
