#!/bin/bash
# Round 5, GPU session 11: the step prologue's three sources of the scalars (ring in pinned host memory / predicted table /
# per-iteration upload) — tests, same-box A/B on the driver window, the seam line; the instant-ngp seam test.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s11
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
timeout 420 python -m pytest tests/test_gpu_pipeline_seam.py "tests/test_gpu_fused_launches.py" -k "seam or camera or prologue or ngp" -q -m gpu > $out/pytest.log 2>&1
el "pytest: rc $? $(tail -1 $out/pytest.log)"
grep -E "^E  |Error|^FAILED|passed|failed" $out/pytest.log | head -30
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], 'long', (j.get('long_run') or {}).get('ms_per_step'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 100 --profile-steps 1"
{
for i in 1 2 3; do
  for mode in ring table 0; do
    echo "== NSAMD_STEP_PROLOGUE=$mode"; NSAMD_STEP_PROLOGUE=$mode timeout 150 $B 2>/dev/null | line
  done
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
for mode in ring 0; do
  NSAMD_STEP_PROLOGUE=$mode timeout 200 python scripts/bench_seam.py > $out/bench_seam_$mode.json 2> $out/bench_seam_$mode.err
  el "seam ($mode) rc $?"
  python -c "
import json; j=json.load(open('$out/bench_seam_$mode.json'))
print('$mode', {k: j.get(k) for k in ('direct_pool_ms','direct_set_batch_ms','seam_ms','seam_over_direct_pool','seam_over_direct_pool_per_window')})"
done
el end
