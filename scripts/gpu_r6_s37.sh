#!/bin/bash
# Round 6, GPU session 37: why the roofline leg's live timing of the field backward (0.17 - 0.24 ms) scatters above the kernel table's
# 0.16 ms: per-launch samples of the profiled iterations, default run
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s37
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for arm in "default:" "long0:--long-steps 0"; do
  name=${arm%%:*}; flags=${arm#*:}
  echo "== $name"
  NSAMD_ROOFLINE_SAMPLES=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary $flags 2>&1 | grep "samples\|^{" | cut -c1-600
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
