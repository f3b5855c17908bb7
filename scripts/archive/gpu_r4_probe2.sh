#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_p2; mkdir -p $OUT; cd $R
timeout 300 python -m pytest "tests/test_gpu_bench_parity.py::test_backward_that_emits_the_scatter_records_equals_the_two_launches" -q -s 2>&1 | grep -E "passed|failed|fused route|Error|assert" | cut -c1-250
for skip in 0 8 32; do
  echo "=== NSAMD_FIELD_BWD_SKIP=$skip"
  NSAMD_FIELD_BWD_SKIP=$skip timeout 120 python scripts/probe_field_clocks.py --no-build --route 2>&1 | grep -v amdgpu.ids | grep -E "bwd|it 2|it 5|loop end|emit partials" | tee -a $OUT/probe_route.log
done
for arm in 1 0 1 0; do
  NSAMD_FUSE_ROUTE=$arm timeout 200 python bench.py --no-cpu-baseline --long-steps 0 --kernel-table > $OUT/bench_fuse$arm.json 2> $OUT/bench_fuse${arm}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_fuse$arm.json")); print("FUSE_ROUTE=$arm", d["ms_per_step"], d["config"]["window_ms"]["min"], d["config"]["final_loss"])
PY
  grep -v amdgpu.ids $OUT/bench_fuse${arm}_table.log | head -n 3
done
