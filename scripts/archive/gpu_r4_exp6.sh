#!/bin/bash
# round 4, experiment call 6: compute units left out of the main field's backward on update iterations, so that the proposal
# networks' backward chains run DURING it (NSAMD_BWD_RESERVE_CUS = 0 / 8 / 16 / 32); floor = chains skipped
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp6; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_training.py -m gpu -q -x -k "bench_configuration or reproducible or emits" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
for r in 0 16 8 32 0 16; do
  echo "== probe_iteration_times reserve=$r"; NSAMD_BWD_RESERVE_CUS=$r PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
echo "== floor"; NSAMD_BWD_RESERVE_CUS=0 NSAMD_DIAG_SKIP_PROP_BWD=1 PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 300 --param-checksum > $OUT/bench_$name.json 2> /dev/null
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10], "long", d["long_run"]["ms_per_step"])
PY
}
arm r0_a NSAMD_BWD_RESERVE_CUS=0
arm r16_a NSAMD_BWD_RESERVE_CUS=16
arm r8_a NSAMD_BWD_RESERVE_CUS=8
arm r32_a NSAMD_BWD_RESERVE_CUS=32
arm r0_b NSAMD_BWD_RESERVE_CUS=0
arm r16_b NSAMD_BWD_RESERVE_CUS=16
