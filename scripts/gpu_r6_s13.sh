#!/bin/bash
# Round 6, GPU session 13: the density weight-gradient reduce riding the level's scatter apply pass, the scatter finish pass folded
# into the apply pass's last arriver, run-merging for the 96-sample level: bit tests, then same-box A/B against the library of the
# commit before (libnsamd_prev.so; NSAMD_DENSITY_REDUCE_RIDER=0 for its host side), per-kind iteration times + windows.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s13
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2 3; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_prev.so NSAMD_LIB_OLDER_ABI=1 NSAMD_DENSITY_REDUCE_RIDER=0; else unset NSAMD_LIB NSAMD_LIB_OLDER_ABI NSAMD_DENSITY_REDUCE_RIDER; fi
    echo "== $arm"
    PROBE_STEPS=200 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 300 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window', j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config']['final_loss'])"
  done
done
unset NSAMD_LIB NSAMD_DENSITY_REDUCE_RIDER
} > $out/summary.txt 2>&1
cat $out/summary.txt
