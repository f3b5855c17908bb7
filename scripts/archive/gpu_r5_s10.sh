#!/bin/bash
# Round 5, GPU session 10: the device-side head of the iteration (nsamd_step_prologue: step scalars from a table, Philox draws)
# and the parallel loss-value reduction — tests, same-box A/B of NSAMD_STEP_PROLOGUE, the seam line, the idle-gap trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s10
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
timeout 600 python -m pytest tests/test_gpu_fused_launches.py tests/test_gpu_bench_parity.py tests/test_gpu_pipeline_seam.py -x -q -m gpu > $out/pytest.log 2>&1
el "pytest: rc $? $(tail -1 $out/pytest.log)"
grep -E "^E  |Error|assert" $out/pytest.log | head -20
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], 'long', (j.get('long_run') or {}).get('ms_per_step'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 100 --profile-steps 1"
{
for i in 1 2 3 4; do
  echo "== default (device-side prologue)"; timeout 200 $B 2>/dev/null | line
  echo "== upload + torch generator";       NSAMD_STEP_PROLOGUE=0 timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
timeout 300 python scripts/bench_seam.py > $out/bench_seam.json 2> $out/bench_seam.err
el "seam rc $?"
python -c "
import json; j=json.load(open('$out/bench_seam.json'))
print({k: j[k] for k in ('direct_pool_ms','seam_ms','seam_over_direct_pool','seam_over_direct_pool_per_window')})"
cd /tmp
for arm in seam direct_pool; do
  timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_$arm -o t -- python $R/scripts/bench_seam.py --arms $arm --windows 3 > $out/trace_$arm.log 2>&1
  db=$(find /tmp/tr_$arm -name "*results.db" | head -1)
  echo "== $arm" | tee -a $out/gaps.txt
  python $R/scripts/trace_gaps.py $db --skip 20 --show 3 2>&1 | tee -a $out/gaps.txt
done
el end
