#!/bin/bash
# round-4 checkpoint 2: fused field-backward + scatter-route: parity, then A/B timing on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_c2; mkdir -p $OUT; cd $R
timeout 500 python -m pytest "tests/test_gpu_bench_parity.py::test_backward_that_emits_the_scatter_records_equals_the_two_launches" tests/test_gpu_pipeline_seam.py "tests/test_gpu_packed.py::test_ngp_bench_size_parity_vs_oracle" -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|error|assert|parity|fused route" $OUT/pytest.log | tail -n 30 | cut -c1-300
for arm in 1 0 1 0; do
  NSAMD_FUSE_ROUTE=$arm timeout 200 python bench.py --no-cpu-baseline --long-steps 0 --kernel-table > $OUT/bench_fuse$arm.json 2> $OUT/bench_fuse${arm}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_fuse$arm.json")); print("FUSE_ROUTE=$arm", d["ms_per_step"], d["config"]["window_ms"], d["config"]["final_loss"])
PY
  grep -v amdgpu.ids $OUT/bench_fuse${arm}_table.log | head -n 4
done
