#!/bin/bash
# Round 6, GPU session 52: Adam's gradient (NT=2) and parameters (NT=3) past the caches as well — does the proposal forward beside it
# keep its tables? same-box A/B through NSAMD_LIB
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s52
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in "shipped:" "nt2:NSAMD_LIB=$R/nerfstudio_amd/libnsamd_adamnt2.so" "nt3:NSAMD_LIB=$R/nerfstudio_amd/libnsamd_adamnt3.so"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], 'loss', j['config']['final_loss'], j['config']['param_checksum']['params'][:12])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
