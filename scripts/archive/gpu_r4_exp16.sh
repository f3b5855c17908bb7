#!/bin/bash
# round 4, experiment call 16: the weight-gradient reduce of the main field's backward deferred with the main-field Adam
# (iteration k's reduce leads the Adam branch of iteration k + 1; NSAMD_DEFER_REDUCE=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp16; mkdir -p $OUT; cd $R
run() { # name, env...
  local name=$1; shift
  echo "$name: $(env "$@" PROBE_STEPS=${PROBE_STEPS:-36} timeout 200 python scripts/probe_iteration_times.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -n 1)" | tee -a $OUT/iteration_times.txt
}
for rep in 1 2 3; do
  run base NSAMD_DEFER_REDUCE=0
  run defer_reduce NSAMD_DEFER_REDUCE=1
done
for a in 0 1 0 1; do
  NSAMD_DEFER_REDUCE=$a timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$a.json 2> $OUT/bench_$a.err
  python -c "
import json; d=json.load(open('$OUT/bench_$a.json')); print('window defer_reduce=$a', d['ms_per_step'], d['config']['window_ms']['min'], d['config'].get('final_loss'))" | tee -a $OUT/iteration_times.txt
done
NSAMD_DEFER_REDUCE=1 timeout 300 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -k "same_bits" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 3 $OUT/pytest_a.log | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "field" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 3 $OUT/pytest_b.log | cut -c1-250
