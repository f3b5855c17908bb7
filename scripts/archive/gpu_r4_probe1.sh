#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_p1; mkdir -p $OUT; cd $R
for skip in 0 8 16 24 32; do
  echo "=== NSAMD_FIELD_BWD_SKIP=$skip (8: no record stores, 16: no rank atomics, 32: no emission)"
  NSAMD_FIELD_BWD_SKIP=$skip timeout 120 python scripts/probe_field_clocks.py --no-build --route 2>&1 | grep -v amdgpu.ids | grep -E "bwd|it 2|it 5|loop end|emit partials" | tee -a $OUT/probe_route.log
done
echo "=== plain backward"; timeout 120 python scripts/probe_field_clocks.py --no-build 2>&1 | grep -E "bwd|it 2" | tee -a $OUT/probe_route.log
