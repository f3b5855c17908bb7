#!/bin/bash
# round 4, experiment call 2: ISA clean-up (global record stores instead of flat, branch-free tile fetch, no spills in the fused
# backward, unconditional loads in the dW reduce / apply pass, position loads issued together) — parity tests, then same-box A/B
# against the library of the round's first commit (nerfstudio_amd/libnsamd_prev.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp2; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 4 $OUT/pytest_a.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q -x -k "reproducible or checkpoint" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
P=$R/nerfstudio_amd/libnsamd_prev.so
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 0 --kernel-table --param-checksum > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10])
PY
}
arm prev0 NSAMD_LIB=$P
arm new0 A=1
arm prev1 NSAMD_LIB=$P
arm new1 A=1
arm new_nofuse NSAMD_FUSE_ROUTE=0
arm prev_nofuse NSAMD_LIB=$P NSAMD_FUSE_ROUTE=0
grep -v amdgpu.ids $OUT/bench_prev1_table.log | head -n 12
grep -v amdgpu.ids $OUT/bench_new1_table.log | head -n 24
grep -v amdgpu.ids $OUT/bench_new_nofuse_table.log | head -n 8
for a in prev new; do
  [ $a = prev ] && export NSAMD_LIB=$P || unset NSAMD_LIB
  echo "== probe_iteration_times $a"; PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
unset NSAMD_LIB
timeout 200 python bench.py --no-cpu-baseline --camera-optimizer SO3xR3 --long-steps 0 > $OUT/bench_cam.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_cam.json')); print('camera on', d['ms_per_step'])"
timeout 200 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_table.log; python -c "
import json; d=json.load(open('$OUT/bench_ngp.json')); print('ngp', d['ms_per_step'], d['config'].get('ms_per_step_excluding_refresh'))"
grep -v "amdgpu.ids\|Warning" $OUT/bench_ngp_table.log | head -n 8
