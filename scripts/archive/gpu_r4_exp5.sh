#!/bin/bash
# round 4, experiment call 5: density-MLP backward at 27.6 KB of LDS (co-resident with the main table's apply pass) — do the
# proposal chains of update iterations now overlap with the main backward? Floor: NSAMD_DIAG_SKIP_PROP_BWD=1.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp5; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "density or proposal or gated or pipeline_golden or fused_train or reproduc" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_training.py -m gpu -q -x -k "bench_configuration or reproducible" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
P=$R/nerfstudio_amd/libnsamd_prev2.so
for a in prev2 new floor prev2 new; do
  unset NSAMD_LIB NSAMD_DIAG_SKIP_PROP_BWD
  [ $a = prev2 ] && export NSAMD_LIB=$P
  [ $a = floor ] && export NSAMD_DIAG_SKIP_PROP_BWD=1
  echo "== probe_iteration_times $a"; PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
unset NSAMD_LIB NSAMD_DIAG_SKIP_PROP_BWD
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 300 --kernel-table --param-checksum > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10], "long", d["long_run"]["ms_per_step"])
PY
}
arm prev2_0 NSAMD_LIB=$P
arm new0 A=1
arm prev2_1 NSAMD_LIB=$P
arm new1 A=1
grep -v amdgpu.ids $OUT/bench_new1_table.log | head -n 22 | cut -c1-118
