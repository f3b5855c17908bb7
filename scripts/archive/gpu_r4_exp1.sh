#!/bin/bash
# round 4, experiment call 1: the tests that failed in r4_base2, hash-forward variants (bit equality + timing), proposal-chain order
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp1; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -m gpu -q -k "camera_optimizer_gradients_golden or fused_train_step_behind_the_model_api or data_parallel_path_over_one_rank or hashgrid_golden" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 6 $OUT/pytest_a.log | cut -c1-250
for m in 7 11; do
  NSAMD_HASH_FWD_MODE=$m timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "hashgrid_golden or nerfacto_field_golden or pipeline_golden" > $OUT/pytest_hash$m.log 2>&1; echo "pytest hash mode $m rc=$?"; tail -n 2 $OUT/pytest_hash$m.log | cut -c1-200
done
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 0 --kernel-table > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"])
PY
  grep "encode_fwd\[L=16" $OUT/bench_${name}_table.log | cut -c1-140
}
arm base0 A=1
arm hash7 NSAMD_HASH_FWD_MODE=7
arm hash11 NSAMD_HASH_FWD_MODE=11
arm first NSAMD_PROP_ORDER=first
arm head NSAMD_PROP_ORDER=head
arm base1 A=1
arm hash7b NSAMD_HASH_FWD_MODE=7
arm firstb NSAMD_PROP_ORDER=first
arm headb NSAMD_PROP_ORDER=head
for o in beside first head; do
  echo "== probe_iteration_times NSAMD_PROP_ORDER=$o"
  NSAMD_PROP_ORDER=$o PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 2 | tee -a $OUT/probe_iter.log
done
