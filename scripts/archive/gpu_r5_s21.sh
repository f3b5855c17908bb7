#!/bin/bash
# Round 5, GPU session 21: the launch that selects the batch and writes the initial bins reads 45 us in the replayed trace
# (14 us alone) beside the HBM-saturating main-field Adam. Arms: fewer Adam workgroups per CU; the Adam branch forked BEHIND
# that launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s21
mkdir -p $out
export TMPDIR=/tmp
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], 'long', (j.get('long_run') or {}).get('ms_per_step'), j['config'].get('param_checksum',{}).get('params','')[:12])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 100 --profile-steps 1 --param-checksum"
{
for i in 1 2 3; do
  for arm in "X=0" "NSAMD_ADAM_BLOCKS_PER_CU=4" "NSAMD_ADAM_BLOCKS_PER_CU=2" "NSAMD_FORK_AFTER_BINS=1" "NSAMD_FORK_AFTER_BINS=1 NSAMD_ADAM_BLOCKS_PER_CU=4"; do
    echo "== $arm"; env $arm timeout 150 $B 2>/dev/null | line
  done
done
} > $out/ab_bench.txt 2>&1
cat $out/ab_bench.txt
