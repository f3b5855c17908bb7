#!/bin/bash
# Round 6, GPU session 56: the proposal networks' Adam deferred to the head of the next iteration's Adam branch
# (NSAMD_DEFER_PROPS_ADAM=1) — same bits? same-box A/B; eager arm bits
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s56
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "default:" "defer_props:NSAMD_DEFER_PROPS_ADAM=1"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>&1 | grep '^{\|Error' | python -c "
import json,sys
t=sys.stdin.read()
try:
    j=json.loads(t); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12], j['config']['param_checksum'].get('exp_avg','')[:12])
except Exception as e: print('FAILED', t[-1500:])"
  done
done
echo "== eager (bits)"
for arm in "default:" "defer_props:NSAMD_DEFER_PROPS_ADAM=1"; do
  name=${arm%%:*}; envs=${arm#*:}
  env NSAMD_DEFER_MAIN_ADAM=1 $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --param-checksum --no-graph 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$name eager-deferred window', j['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
done
timeout 600 env NSAMD_DEFER_PROPS_ADAM=1 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_training.py -m gpu -x -q -k "graph_replay or bit_repro or trajectory or checkpoint" 2>&1 | tail -3
} > $out/summary.txt 2>&1
cat $out/summary.txt
