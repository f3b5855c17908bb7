#!/bin/bash
# Round 6, GPU session 16: the machine scheduler's strategy for the whole library (max-ilp / max-memory-clause) against the
# default: the field backward alone on real buffers, then driver windows. Same-box, alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s16
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
  for arm in default ilp memclause; do
    if [ $arm = default ]; then unset NSAMD_LIB; else export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_$arm.so; fi
    timeout 120 python scripts/probe_field_bwd_real.py 30 2>&1 | grep "^lib" | cut -c1-130
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 100 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$arm window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
