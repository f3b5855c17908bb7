#!/bin/bash
# round 4, experiment call 8: do the proposal chains share a hardware queue with the main chain? GPU_MAX_HW_QUEUES (ROCm runtime,
# default 4) against update / non-update iteration times, graph replay and eager streams
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp8; mkdir -p $OUT; cd $R
for q in default 8 2 16 default 8; do
  [ $q = default ] && unset GPU_MAX_HW_QUEUES || export GPU_MAX_HW_QUEUES=$q
  echo "== graph GPU_MAX_HW_QUEUES=$q"; PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
for q in default 8; do
  [ $q = default ] && unset GPU_MAX_HW_QUEUES || export GPU_MAX_HW_QUEUES=$q
  echo "== eager GPU_MAX_HW_QUEUES=$q"; PROBE_EAGER=1 PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
for q in default 8 default 8; do
  [ $q = default ] && unset GPU_MAX_HW_QUEUES || export GPU_MAX_HW_QUEUES=$q
  timeout 200 python bench.py --no-cpu-baseline --long-steps 300 > $OUT/bench_q$q.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_q$q.json')); print('bench queues=$q', d['ms_per_step'], d['config']['window_ms']['min'], 'long', d['long_run']['ms_per_step'])"
done
