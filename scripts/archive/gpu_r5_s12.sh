#!/bin/bash
# Round 5, GPU session 12: per-level cost of the main hash forward by resolution (scripts/probe_hash_levels.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s12
mkdir -p $out
export TMPDIR=/tmp
timeout 200 python scripts/probe_hash_levels.py > $out/hash_levels.txt 2> $out/hash_levels.err
echo "rc $?"; cat $out/hash_levels.txt; tail -3 $out/hash_levels.err
