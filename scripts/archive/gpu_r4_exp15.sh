#!/bin/bash
# round 4, experiment call 15: the proposal networks' Adam at the end of their side branch instead of behind a join on the main path
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp15; mkdir -p $OUT; cd $R
run() { # name, env...
  local name=$1; shift
  echo "$name: $(env "$@" PROBE_STEPS=${PROBE_STEPS:-36} timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -n 1)" | tee -a $OUT/iteration_times.txt
}
for rep in 1 2 3; do
  run base A=0
  run adam_on_branch NSAMD_PROPS_ADAM_ON_BRANCH=1
  run adam_on_one_branch NSAMD_PROPS_ADAM_ON_BRANCH=1 NSAMD_LEVEL_STREAMS=0
done
NSAMD_PROPS_ADAM_ON_BRANCH=1 timeout 300 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -k "same_bits" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
