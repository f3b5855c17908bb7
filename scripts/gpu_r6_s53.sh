#!/bin/bash
# Round 6, GPU session 53: the final tree once more — whole GPU suite, smoke, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s53
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
echo "== bench default"
timeout 600 python bench.py 2>/dev/null | grep '^{' | tee $out/bench_default.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'roofline', r['frac'], r['avg_launch_ms'], r.get('executed_frac'), r['traffic'], 'runner_up', r['runner_up']['frac'], {k:(v.get('ms_per_step') or v.get('seam_over_direct')) for k,v in j.get('secondary',{}).items()})"
} > $out/summary.txt 2>&1
cat $out/summary.txt
