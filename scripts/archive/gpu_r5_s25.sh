#!/bin/bash
# Round 5, GPU session 25: the zero-fills of an iteration as one launch (nsamd_zero_spans) — unit test, graph == eager parity,
# bit reproducibility, A/B against one fill kernel per span
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s25
mkdir -p $out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_fused_launches.py tests/test_gpu_bench_parity.py tests/test_gpu_training.py -k "zero_spans or parity or reproducible or merged_launches" -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc $? $(tail -1 $out/pytest.log)"; grep -E "^E  |^FAILED" $out/pytest.log | head
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], 'long', (j.get('long_run') or {}).get('ms_per_step'), j['config'].get('param_checksum',{}).get('params','')[:12])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 100 --profile-steps 1 --param-checksum"
{
for i in 1 2 3; do
  for z in 0 1; do
    echo "== NSAMD_ZERO_SPANS=$z"; NSAMD_ZERO_SPANS=$z timeout 150 $B 2>/dev/null | line
  done
done
} > $out/ab_bench.txt 2>&1
cat $out/ab_bench.txt
