#!/bin/bash
# Round 6, GPU session 9: per-kind iteration times inside the loop (scripts/probe_iteration_times.py) and bench windows from step 40
# for the table's Adam inside the apply pass (1) against the deferred launch (0), alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s9
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in 0 1; do
    export NSAMD_FUSE_TABLE_ADAM=$arm
    echo "== NSAMD_FUSE_TABLE_ADAM=$arm"
    PROBE_STEPS=200 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --start-step 40 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window from step 40:', j['ms_per_step'], j['value'])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
