#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02h}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== all GPU tests" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "mean PSNR|^   [0-9] \||GPU - oracle|passed|failed|^E  " $OUT/pytest_gpu.log | head -40 | tee -a $OUT/summary.txt
echo "== bench default" | tee -a $OUT/summary.txt
timeout 600 python bench.py --kernel-table > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
cat $OUT/bench_default.json | tee -a $OUT/summary.txt
head -n 24 $OUT/bench_default_kernel_table.log | tee -a $OUT/summary.txt
echo "== bench unbounded" | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload unbounded --no-cpu-baseline --kernel-table > $OUT/bench_unbounded.json 2> $OUT/bench_unbounded_kernel_table.log
cat $OUT/bench_unbounded.json | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
