#!/bin/bash
# round 4, experiment call 10: the scatter's finish pass folded into the apply pass (last workgroup) — whole GPU suite, then
# same-box A/B against the library of commit d597576 (libnsamd_prev2.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp10; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest_gpu.log | cut -c1-250
P=$R/nerfstudio_amd/libnsamd_prev2.so
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 300 --param-checksum --kernel-table > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10], "long", d["long_run"]["ms_per_step"])
PY
}
for i in 0 1 2; do
arm prev2_$i NSAMD_LIB=$P
arm new_$i A=1
done
for a in prev2 new prev2 new; do
  [ $a = prev2 ] && export NSAMD_LIB=$P || unset NSAMD_LIB
  echo "== probe $a"; PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1
done
unset NSAMD_LIB
grep -v amdgpu.ids $OUT/bench_new_2_table.log | head -n 8 | cut -c1-118
