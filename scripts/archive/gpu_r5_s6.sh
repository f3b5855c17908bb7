#!/bin/bash
# Round 5, GPU session 6: where the proposal networks' backward chains run on update iterations (the replayed trace of the
# evidence session shows them starving beside the main backward): their MLP stage in line ahead of the main backward, one side
# branch, everything in line — driver window + 300 steps, per-kind iteration times. Same parameter bits in every arm.
out=gpurun_out/r5_s6
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], 'long', j['long_run']['ms_per_step'], j['config'].get('param_checksum',{}).get('params'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 300 --param-checksum --profile-steps 1"
{
for i in 1 2 3; do
  echo "== default";                      timeout 200 $B 2>/dev/null | line
  echo "== proposal MLP stage in line";   NSAMD_PROP_MLP_INLINE=1 timeout 200 $B 2>/dev/null | line
  echo "== one side branch";              NSAMD_LEVEL_STREAMS=0 timeout 200 $B 2>/dev/null | line
  echo "== MLP in line + one branch";     NSAMD_PROP_MLP_INLINE=1 NSAMD_LEVEL_STREAMS=0 timeout 200 $B 2>/dev/null | line
  echo "== everything in line";           NSAMD_SIDE_STREAM=0 timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
for arm in "" "NSAMD_PROP_MLP_INLINE=1"; do
  echo "== $arm" >> $out/iteration_times.txt
  env $arm PROBE_STEPS=100 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1 >> $out/iteration_times.txt
done
cat $out/iteration_times.txt
el end
