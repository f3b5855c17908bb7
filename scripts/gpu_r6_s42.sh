#!/bin/bash
# Round 6, GPU session 42: current-stream helpers without the device-count query — graph / eager / one-rank data-parallel rehearsal
# (all-reduce and sharded) on one box, two repeats, and the default line's secondary lines (seam ratio)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s42
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -m gpu -x -q -k "one_rank or bit_repro or graph_replay" 2>&1 | tail -3
for i in 1 2; do
  for arm in "graph:" "eager:--no-graph" "force_dp:--force-dp" "force_dp_sharded:--force-dp --dp-mode sharded"; do
    name=${arm%%:*}; flags=${arm#*:}
    echo "== $name"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 $flags 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['launch'][:40])"
  done
done
echo "== default line"
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], 'long', j['long_run']['ms_per_step'], {k:(v.get('ms_per_step') or v.get('seam_over_direct')) for k,v in j.get('secondary',{}).items()})"
} > $out/summary.txt 2>&1
cat $out/summary.txt
