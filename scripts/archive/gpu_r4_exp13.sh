#!/bin/bash
# round 4, experiment call 13: the per-iteration upload of the step-dependent scalars double-buffered (NSAMD_HYPER_PARITY=1) and
# the global depth clip beside the losses / backward (NSAMD_CLIP_BESIDE=1); per-kind iteration times, arms alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp13; mkdir -p $OUT; cd $R
run() { # name, env...
  local name=$1; shift
  echo "$name: $(env "$@" PROBE_STEPS=${PROBE_STEPS:-36} timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -n 1)" | tee -a $OUT/iteration_times.txt
}
for rep in 1 2 3; do
  run base A=0
  run parity NSAMD_HYPER_PARITY=1
  run clip NSAMD_CLIP_BESIDE=1
  run both NSAMD_HYPER_PARITY=1 NSAMD_CLIP_BESIDE=1
done
for a in base both base both; do
  [ $a = both ] && export NSAMD_HYPER_PARITY=1 NSAMD_CLIP_BESIDE=1 || unset NSAMD_HYPER_PARITY NSAMD_CLIP_BESIDE
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$a.json 2> $OUT/bench_$a.err
  python -c "
import json; d=json.load(open('$OUT/bench_$a.json')); print('window $a', d['ms_per_step'], d['config']['window_ms'], d['config'].get('final_loss'))" | tee -a $OUT/iteration_times.txt
done
unset NSAMD_HYPER_PARITY NSAMD_CLIP_BESIDE
timeout 300 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -k "same_bits" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
