#!/bin/bash
# Round 6, GPU session 38: ray terms as the default — the whole GPU suite, the default bench line, the eval render
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s38
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench default"
timeout 600 python bench.py 2>/dev/null | grep '^{' | tee $out/bench_default.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['roofline'].get('executed_frac'), {k:(v.get('ms_per_step') or v.get('seam_over_direct')) for k,v in j.get('secondary',{}).items()})"
echo "== eval render"
timeout 300 python scripts/bench_render.py --frames 5 2>/dev/null | grep '^{' | cut -c1-400
} > $out/summary.txt 2>&1
tail -40 $out/summary.txt
