#!/bin/bash
# Round 6, GPU session 29: new defaults (merged proposal chain + one more sweep of the field backward on update iterations):
# the whole GPU suite, then the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s29
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench default"
timeout 600 python bench.py 2>/dev/null | grep '^{' | tee $out/bench_default.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['avg_launch_ms'], {k:(v.get('ms_per_step') or v.get('seam_over_direct')) for k,v in j.get('secondary',{}).items()})"
} > $out/summary.txt 2>&1
tail -30 $out/summary.txt
