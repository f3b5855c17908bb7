#!/bin/bash
# Round-3 GPU session 6: fixed tests, eval renderer, bf16x2 weight gradients (accuracy + time), run-kernel levels per
# thread, apply early-out.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s6}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== tests (parity, packed, eval renderer, table scatter, training reproducibility)" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_packed.py tests/test_gpu_kernels.py tests/test_gpu_training.py -m gpu -q -s -k "bench or packed or occ or marcher or eval or scatter or reproducible or runner or gated or field_mlp_backward" > $OUT/pytest_a.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |Error|excluded" $OUT/pytest_a.log | grep -v "hash_table\[level" | cut -c1-300 | head -40 | tee -a $OUT/summary.txt
echo "== bf16x2 weight gradients: field tests + kernel-level float64 test" | tee -a $OUT/summary.txt
NSAMD_FIELD_BWD_BF16X2=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_parity.py -m gpu -q -s -k "nerfacto_field_golden or field_mlp_ragged or pipeline_golden or runner_matches or field_mlp_backward" > $OUT/pytest_bf2.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |Error|excluded|gpu-f64" $OUT/pytest_bf2.log | cut -c1-300 | head -40 | tee -a $OUT/summary.txt
echo "== bf16x2 timing (driver window, kernel table)" | tee -a $OUT/summary.txt
for b in 0 1; do NSAMD_FIELD_BWD_BF16X2=$b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table 2> $OUT/ktable_bf2_$b.log | cut -c1-200 | tee -a $OUT/summary.txt; grep -E "field_mlp" $OUT/ktable_bf2_$b.log | tee -a $OUT/summary.txt; done
echo "== run-kernel levels per thread (NSAMD_RUNS_LEVELS), driver window + 300 steps + kernel table" | tee -a $OUT/summary.txt
for l in 4 2 1; do NSAMD_RUNS_LEVELS=$l timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table 2> $OUT/ktable_runs_$l.log | cut -c1-180 | tee -a $OUT/summary.txt; grep -E "encode_bwd_gated" $OUT/ktable_runs_$l.log | tee -a $OUT/summary.txt; NSAMD_RUNS_LEVELS=$l timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-180 | tee -a $OUT/summary.txt; done
echo "== eval render, 800x800: device-side chunk loop / module loop" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_render.py 2>/dev/null | tail -n 1 | tee $OUT/bench_render_runner.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_render.py --module-loop 2>/dev/null | tail -n 1 | tee $OUT/bench_render_module.json | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
