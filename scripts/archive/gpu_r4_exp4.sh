#!/bin/bash
# round 4, experiment call 4: apply tile as feature planes, probe switches folded away, burst loads in the resample / loss /
# gated weights-backward kernels and in both route kernels — parity, then same-box A/B against libnsamd_prev.so
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp4; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_packed.py -m gpu -q -x -k "reproducible or ngp or marcher" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
P=$R/nerfstudio_amd/libnsamd_prev.so
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 0 --kernel-table --param-checksum > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10])
PY
}
for i in 0 1; do
arm prev$i NSAMD_LIB=$P
arm new$i A=1
done
grep -v amdgpu.ids $OUT/bench_prev1_table.log | head -n 26 | cut -c1-118
echo ---- new
grep -v amdgpu.ids $OUT/bench_new1_table.log | head -n 26 | cut -c1-118
for a in prev new prev new; do
  [ $a = prev ] && export NSAMD_LIB=$P || unset NSAMD_LIB
  echo "== probe_iteration_times $a"; PROBE_STEPS=30 timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/probe_iter.log
done
unset NSAMD_LIB
for a in prev new; do
  [ $a = prev ] && export NSAMD_LIB=$P || unset NSAMD_LIB
  timeout 200 python bench.py --no-cpu-baseline --windows 1 --long-steps 300 > $OUT/bench_long_$a.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_long_$a.json')); print('long $a', d['long_run'])"
  timeout 200 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline --kernel-table > $OUT/bench_ngp_$a.json 2> $OUT/bench_ngp_${a}_table.log; python -c "
import json; d=json.load(open('$OUT/bench_ngp_$a.json')); print('ngp $a', d['ms_per_step'], d['config'].get('ms_per_step_excluding_refresh'))"
  grep -v "amdgpu.ids\|Warning" $OUT/bench_ngp_${a}_table.log | head -n 8 | cut -c1-118
done
unset NSAMD_LIB
