#!/bin/bash
# round 4, experiment call 11: instant-ngp schedule — every ray marched once (stash), density-only head on the candidates
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_exp11; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_packed.py -m gpu -q > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -n 4 $OUT/pytest_a.log | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "field" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -n 2 $OUT/pytest_b.log | cut -c1-250
for a in two one two one; do
  [ $a = two ] && export NSAMD_NGP_TWO_PASS=1 || unset NSAMD_NGP_TWO_PASS
  timeout 200 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline --kernel-table > $OUT/bench_ngp_$a.json 2> $OUT/bench_ngp_${a}_table.log; python -c "
import json; d=json.load(open('$OUT/bench_ngp_$a.json')); print('ngp $a', d['ms_per_step'], d['config'].get('ms_per_step_excluding_refresh'), d['config']['final_loss'], d['roofline'].get('marching_passes_per_step'))"
done
unset NSAMD_NGP_TWO_PASS
grep -v "amdgpu.ids\|Warning" $OUT/bench_ngp_one_table.log | head -n 12 | cut -c1-118
