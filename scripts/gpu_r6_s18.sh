#!/bin/bash
# Round 6, GPU session 18: workgroups per CU of the deferred main-field Adam (NSAMD_ADAM_BLOCKS_PER_CU = 4 / 3 / 2 / 1), three
# alternating repeats: window + 300-step long run.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s18
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for n in 4 3 2 1; do
    echo "== NSAMD_ADAM_BLOCKS_PER_CU=$n"
    NSAMD_ADAM_BLOCKS_PER_CU=$n timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
