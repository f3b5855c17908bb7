#!/bin/bash
# GPU-box session: all GPU tests (with the PSNR table), field-backward timing, bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02d}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -E "mean PSNR|^   [0-9] \||GPU - oracle|passed|failed|Error|^E  " $OUT/pytest_gpu.log | head -60 | tee -a $OUT/summary.txt
echo "== bench default" | tee -a $OUT/summary.txt
timeout 600 python bench.py --kernel-table --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
cat $OUT/bench_default.json | tee -a $OUT/summary.txt
head -n 14 $OUT/bench_default_kernel_table.log | tee -a $OUT/summary.txt
echo "== scatter main in isolation" | tee -a $OUT/summary.txt
timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
