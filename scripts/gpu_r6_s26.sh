#!/bin/bash
# Round 6, GPU session 26: clock stamps of the proposal levels' gated scatter in the sparse phase (step ~15) and past it (step ~150)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s26
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for steps in 14 150; do
  for lvl in 0 1; do
    echo "=== PROBE_STEPS=$steps level $lvl"
    PROBE_STEPS=$steps timeout 300 python scripts/probe_gated_scatter_clocks.py $lvl 2>&1 | grep -v amdgpu.ids
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
