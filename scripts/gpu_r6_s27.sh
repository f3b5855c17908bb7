#!/bin/bash
# Round 6, GPU session 27: the proposal levels' backward chains merged stage by stage (nsamd_proposal_levels_bwd) — the new bit-
# equality test + the gated-chain tests, then same-box A/B: merged / level by level, each with and without 36 CUs left to the chains.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s27
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q -k "merged_proposal or gated_proposal or bit_repro or graph_replay" 2>&1 | tail -5
for i in 1 2 3; do
  for arm in "merged:" "level_by_level:NSAMD_MERGE_PROP_LEVELS=0" "merged_reserve36:NSAMD_BWD_RESERVE_CUS=36" "level_by_level_reserve36:NSAMD_MERGE_PROP_LEVELS=0 NSAMD_BWD_RESERVE_CUS=36" "merged_in_line:NSAMD_SIDE_STREAM=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
