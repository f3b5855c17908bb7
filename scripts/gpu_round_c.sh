#!/bin/bash
# GPU-box session: kernel tests only (training tests separately), scatter isolation, bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu (kernels + reproducibility)" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -s -k "not psnr" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|Error|assert|^E " $OUT/pytest_gpu.log | head -40 | tee -a $OUT/summary.txt
echo "== scatter main in isolation" | tee -a $OUT/summary.txt
for v in 114 524; do
  NSAMD_SCATTER_SHAPE=$v timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
done
echo "== bench default" | tee -a $OUT/summary.txt
timeout 600 python bench.py --kernel-table --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
cat $OUT/bench_default.json | tee -a $OUT/summary.txt
head -n 12 $OUT/bench_default_kernel_table.log | tee -a $OUT/summary.txt
NSAMD_SCATTER_SHAPE=524 timeout 600 python bench.py --kernel-table --no-cpu-baseline > $OUT/bench_524.json 2> $OUT/bench_524_kernel_table.log
cat $OUT/bench_524.json | cut -c1-220 | tee -a $OUT/summary.txt
head -n 4 $OUT/bench_524_kernel_table.log | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
