#!/bin/bash
# Round 6, GPU session 44: camera optimiser on (SO3xR3, the reference's nerfacto default) — the rays' gradient through the main grid
# beside the scatter's apply pass instead of behind it: camera tests, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s44
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline_seam.py -m gpu -x -q -k "camera or SO3xR3" 2>&1 | tail -3
for i in 1 2 3; do
  for arm in "beside:" "behind:NSAMD_RAYS_BESIDE_APPLY=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --camera-optimizer SO3xR3 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
