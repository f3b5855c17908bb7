#!/bin/bash
# round 4, experiment call 9: the field backward's weight-gradient reduce beside the table scatter's apply pass (fused path)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp9; mkdir -p $OUT; cd $R
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 300 --param-checksum > $OUT/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10], "long", d["long_run"]["ms_per_step"])
PY
}
for i in 0 1 2; do
arm split0_$i NSAMD_SPLIT_REDUCE=0
arm split1_$i NSAMD_SPLIT_REDUCE=1
done
for v in 0 1 0 1; do
  echo "== probe split=$v"; NSAMD_SPLIT_REDUCE=$v PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1
done
