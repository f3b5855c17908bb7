#!/bin/bash
# round 4, experiment call 7: proposal chains made cheap for idle workgroups (density backward: contiguous chunk ranges, one mask
# burst, zero row + exit; route kernels: mask bytes first) + compute units reserved for them beside the main field's backward
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp7; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "density or proposal or gated or pipeline_golden or fused_train or reproduc or scatter or hashgrid or camera" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_training.py tests/test_gpu_packed.py -m gpu -q -x -k "bench_configuration or reproducible or ngp or emits" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
P=$R/nerfstudio_amd/libnsamd_prev2.so
echo "== probe prev2 reserve=0"; NSAMD_LIB=$P NSAMD_BWD_RESERVE_CUS=0 PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1
for r in 0 16 32 48 0 32; do
  echo "== probe new reserve=$r"; NSAMD_BWD_RESERVE_CUS=$r PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 300 --param-checksum --kernel-table > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10], "long", d["long_run"]["ms_per_step"])
PY
}
arm prev2_a NSAMD_LIB=$P NSAMD_BWD_RESERVE_CUS=0
arm new_r0_a NSAMD_BWD_RESERVE_CUS=0
arm new_r32_a NSAMD_BWD_RESERVE_CUS=32
arm new_r16_a NSAMD_BWD_RESERVE_CUS=16
arm prev2_b NSAMD_LIB=$P NSAMD_BWD_RESERVE_CUS=0
arm new_r0_b NSAMD_BWD_RESERVE_CUS=0
arm new_r32_b NSAMD_BWD_RESERVE_CUS=32
grep -v amdgpu.ids $OUT/bench_new_r0_b_table.log | grep "gated\|gate\|density_mlp" | cut -c1-118
