#!/bin/bash
# Round 5, GPU session 22: the Adam pass alone (eager per-launch events) with 8 / 4 / 3 / 2 workgroups per CU
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s22
mkdir -p $out
export TMPDIR=/tmp
for b in 8 4 3 2; do
  echo "== NSAMD_ADAM_BLOCKS_PER_CU=$b" | tee -a $out/adam_alone.txt
  NSAMD_ADAM_BLOCKS_PER_CU=$b timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --windows 1 --kernel-table 2>&1 >/dev/null | grep adam_step | tee -a $out/adam_alone.txt
done
