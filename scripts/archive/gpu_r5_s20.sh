#!/bin/bash
# Round 5, GPU session 20: cost of the proposal networks' fused forward by grid resolution (scripts/probe_density_levels.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s20
mkdir -p $out
export TMPDIR=/tmp
timeout 200 python scripts/probe_density_levels.py > $out/density_levels.txt 2> $out/density_levels.err
echo "rc $?"; cat $out/density_levels.txt; grep -v amdgpu $out/density_levels.err | tail -5
