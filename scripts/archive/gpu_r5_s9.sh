#!/bin/bash
# Round 5, GPU session 9: kernel trace of the iteration through the seam and of the direct line: where the GPU idles.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s9
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for arm in seam direct_set_batch direct_pool; do
  timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_$arm -o t -- python $R/scripts/bench_seam.py --arms $arm --windows 3 > $out/trace_$arm.log 2>&1
  db=$(find /tmp/tr_$arm -name "*results.db" | head -1)
  echo "== $arm ($db)" | tee -a $out/gaps.txt
  grep '^{' $out/trace_$arm.log | cut -c1-200 | tee -a $out/gaps.txt
  python $R/scripts/trace_gaps.py $db --skip 20 2>&1 | tee -a $out/gaps.txt
done
