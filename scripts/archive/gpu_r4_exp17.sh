#!/bin/bash
# round 4, experiment call 17: scatter apply pass with its first dependent loads (segment counts, first trip of records) issued
# around the LDS zero-fill instead of behind the barrier; same-box A/B against the library built just before (NSAMD_LIB)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp17; mkdir -p $OUT; cd $R
P=$R/nerfstudio_amd/libnsamd_prev.so
for i in 1 2 3; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$P; else unset NSAMD_LIB; fi
    echo "$arm: $(PROBE_STEPS=36 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v 'amdgpu.ids\|Warning' | tail -n 1)" | tee -a $OUT/summary.txt
  done
done
for i in 1 2; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$P; else unset NSAMD_LIB; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('window $arm', j['ms_per_step'], j['config']['window_ms']['min'], j['config'].get('param_checksum',{}).get('params'))" | tee -a $OUT/summary.txt
  done
done
unset NSAMD_LIB
