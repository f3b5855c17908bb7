#!/bin/bash
# Round 6, GPU session 47: the rays' gradient with every lane busy (hash_encode_bwd_rays_flat_kernel) — camera tests, same-box A/B with
# the camera optimiser on; the two-ranks-on-one-GPU test after bench.py's fix of the profiling iterations' count on ranks > 0
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s47
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
timeout 900 python -m pytest tests -m gpu -x -q -k "camera or SO3xR3 or two_ranks or one_rank" 2>&1 | tail -4
for i in 1 2 3; do
  for arm in "flat:" "wave_per_ray:NSAMD_RAYS_FLAT=0"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --camera-optimizer SO3xR3 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'])"
  done
done
echo "== kernel table, camera on"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --camera-optimizer SO3xR3 --kernel-table 2>&1 | grep -v "^{" | head -8
} > $out/summary.txt 2>&1
cat $out/summary.txt
