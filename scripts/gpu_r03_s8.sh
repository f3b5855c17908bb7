#!/bin/bash
# Round-3 GPU session 8: full GPU suite after the fixes, ngp line with per-point appearance rows, eval render.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s8}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== pytest -m gpu (all)" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_gpu.log | cut -c1-250 | head -40 | tee -a $OUT/summary.txt
echo "== ngp workload" | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload ngp --steps 30 --warmup 5 --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_kernel_table.log
echo "rc=$?" | tee -a $OUT/summary.txt
cut -c1-2500 $OUT/bench_ngp.json | tee -a $OUT/summary.txt
grep -v "amdgpu.ids\|Warning\|detach\|final_loss" $OUT/bench_ngp_kernel_table.log | head -14 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
