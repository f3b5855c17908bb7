#!/bin/bash
# Round 6, GPU session 14: which of session 13's three changes costs the update iterations? Arms on one box: prev library; new with
# the rider off / the 96-sample merging off / both off (= only the folded finish pass differs from prev); new.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s14
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
  for arm in prev rider0_merge0 rider0 merge0 new; do
    unset NSAMD_LIB NSAMD_LIB_OLDER_ABI NSAMD_DENSITY_REDUCE_RIDER NSAMD_SCATTER_MERGE_96
    case $arm in
      prev) export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_prev.so NSAMD_LIB_OLDER_ABI=1 NSAMD_DENSITY_REDUCE_RIDER=0;;
      rider0_merge0) export NSAMD_DENSITY_REDUCE_RIDER=0 NSAMD_SCATTER_MERGE_96=0;;
      rider0) export NSAMD_DENSITY_REDUCE_RIDER=0;;
      merge0) export NSAMD_SCATTER_MERGE_96=0;;
    esac
    echo "== $arm"
    PROBE_STEPS=200 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['final_loss'])"
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
