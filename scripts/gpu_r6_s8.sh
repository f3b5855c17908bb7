#!/bin/bash
# Round 6, GPU session 8: the main table's Adam inside the scatter's apply pass (NSAMD_FUSE_TABLE_ADAM=1) against the separate,
# deferred launch (=0): parameter checksums after the first window must be equal; driver window + long run, alternating arms.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s8
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
  for arm in 0 1; do
    export NSAMD_FUSE_TABLE_ADAM=$arm
    echo "== NSAMD_FUSE_TABLE_ADAM=$arm"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 300 --param-checksum --kernel-table --profile-steps 10 2> $out/table_${arm}_$i.log | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j.get('long_run'), j['config'].get('param_checksum'))"
    grep "apply\|adam" $out/table_${arm}_$i.log | cut -c1-150
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
