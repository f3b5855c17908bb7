#!/bin/bash
# round 4, experiment call 12: launches on the critical path — the jitter draws on the Adam branch (NSAMD_SIDE_JITTER=1, real) and
# the upper bounds of two more (timing diagnostics: no depth-clip launch, no per-iteration hyper-parameter copy)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp12; mkdir -p $OUT; cd $R
NSAMD_SIDE_JITTER=1 timeout 400 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -k "graph" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
run() { # name, env...
  local name=$1; shift
  echo "$name: $(env "$@" PROBE_STEPS=200 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -n 1)" | tee -a $OUT/iteration_times.txt
}
run base A=0
run side_jitter NSAMD_SIDE_JITTER=1
run no_depth_clip NSAMD_DIAG_NO_DEPTH_CLIP=1
run no_hyper NSAMD_DIAG_NO_HYPER=1
run base A=0
run side_jitter NSAMD_SIDE_JITTER=1
run no_hyper NSAMD_DIAG_NO_HYPER=1
