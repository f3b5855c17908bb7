#!/bin/bash
# round 4, experiment call 3: four records of a level emitted together (route_level), burst position loads in the hash forward only,
# prefetching plain backward vs not (libnsamd_noahead.so), fused vs two launches — same box, alternating arms
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp3; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_kernels.py -m gpu -q -x -k "emits or bench_configuration or field or hashgrid or pipeline_golden or reproduc or proposal_density or packed or standalone" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_packed.py -m gpu -q -x -k "reproducible or ngp" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
P=$R/nerfstudio_amd/libnsamd_prev.so; NA=$R/nerfstudio_amd/libnsamd_noahead.so
arm() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --long-steps 0 --kernel-table --param-checksum > $OUT/bench_$name.json 2> $OUT/bench_${name}_table.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); w=d["config"]["window_ms"]; print("ARM $name", d["ms_per_step"], w["min"], w["max"], d["config"]["final_loss"], d["config"].get("param_checksum",{}).get("params","")[:10])
PY
  grep -v amdgpu.ids $OUT/bench_${name}_table.log | grep "field_mlp_bwd\|encode_bwd_set\|encode_fwd\[L=16\|density_field_fwd\[M=10" | cut -c1-118
}
for i in 0 1; do
arm prev$i NSAMD_LIB=$P
arm new$i A=1
arm new_nofuse$i NSAMD_FUSE_ROUTE=0
arm noahead_nofuse$i NSAMD_LIB=$NA NSAMD_FUSE_ROUTE=0
arm prev_nofuse$i NSAMD_LIB=$P NSAMD_FUSE_ROUTE=0
done
for a in new noahead; do
  [ $a = noahead ] && export NSAMD_LIB=$NA || unset NSAMD_LIB
  timeout 200 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline --kernel-table > $OUT/bench_ngp_$a.json 2> $OUT/bench_ngp_${a}_table.log; python -c "
import json; d=json.load(open('$OUT/bench_ngp_$a.json')); print('ngp $a', d['ms_per_step'], d['config'].get('ms_per_step_excluding_refresh'))"
  grep -v "amdgpu.ids\|Warning" $OUT/bench_ngp_${a}_table.log | head -n 3
done
unset NSAMD_LIB
