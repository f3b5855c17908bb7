#!/bin/bash
# A/B of the deferred main-table scatter (NSAMD_DEFER_SCATTER) on one box: driver window, 300 steps, late-training schedule;
# parameter checksums of the two arms must agree. -> gpurun_out/<tag>/
tag=${1:-defer_scatter}
out=gpurun_out/$tag
mkdir -p $out
{
for arm in 1 0 1 0; do
  echo "== NSAMD_DEFER_SCATTER=$arm  --steps 20 --warmup 5"
  NSAMD_DEFER_SCATTER=$arm timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'), j['config']['launch'][:70])"
done
for arm in 1 0; do
  echo "== NSAMD_DEFER_SCATTER=$arm  --steps 300"
  NSAMD_DEFER_SCATTER=$arm timeout 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'))"
  echo "== NSAMD_DEFER_SCATTER=$arm  --start-step 5000 --steps 120"
  NSAMD_DEFER_SCATTER=$arm timeout 300 python bench.py --start-step 5000 --steps 120 --warmup 6 --no-cpu-baseline --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'))"
done
echo "== eager N=1 (no graph) checksum, 20 steps"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum --no-graph 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'))"
echo "== graph-vs-eager bit tests"
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_bench_parity.py -m gpu -q -x 2>&1 | tail -5
} > $out/summary.txt 2>&1
cat $out/summary.txt
