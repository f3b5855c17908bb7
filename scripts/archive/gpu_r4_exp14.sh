#!/bin/bash
# round 4, experiment call 14: what do the parallel branches of the captured iteration cost? (exp13: ONE more fork / join pair
# around a 4 us launch cost 60 us per iteration.) Per-kind iteration times with fewer branches, arms alternating on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp14; mkdir -p $OUT; cd $R
run() { # name, env...
  local name=$1; shift
  echo "$name: $(env "$@" PROBE_STEPS=${PROBE_STEPS:-36} timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -n 1)" | tee -a $OUT/iteration_times.txt
}
for rep in 1 2; do
  run base A=0
  run one_side_branch NSAMD_LEVEL_STREAMS=0
  run no_side_branch NSAMD_SIDE_STREAM=0
  run adam_in_line NSAMD_DEFER_MAIN_ADAM=0
  run linear NSAMD_DEFER_MAIN_ADAM=0 NSAMD_SIDE_STREAM=0
  run skip_prop_bwd NSAMD_DIAG_SKIP_PROP_BWD=1
done
