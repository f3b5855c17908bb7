#!/bin/bash
# Round 6, GPU session 49: instant-ngp schedule — the field backward emits the table scatter's records (as the nerfacto schedule does):
# packed tests, then the same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s49
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline_seam.py -m gpu -x -q -s -k "ngp" 2>&1 | grep -v "^$" | tail -8
for i in 1 2 3; do
  for arm in "fused:" "two_launches:NSAMD_NGP_FUSE_ROUTE=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms', j['ms_per_step'], j['value'], 'loss', j['config'].get('final_loss'))"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
