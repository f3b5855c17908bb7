#!/bin/bash
# GPU-box session: tests (no -x), field-backward attribution, scatter variants in isolation, bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -E "PSNR table|^   \(|mean delta|passed|failed|Error|assert" $OUT/pytest_gpu.log | head -60 | tee -a $OUT/summary.txt
echo "== scatter main in isolation" | tee -a $OUT/summary.txt
for v in 114 124 122 524 522 514 112; do
  NSAMD_SCATTER_SHAPE=$v timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
done
NSAMD_SCATTER_TILES=1024 timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
echo "== field backward attribution (NSAMD_FIELD_BWD_SKIP: 1 no dW, 2 no barriers, 4 no data-gradient GEMMs)" | tee -a $OUT/summary.txt
for k in 0 1 2 3 4 5 7; do
  echo "-- skip=$k" | tee -a $OUT/summary.txt
  NSAMD_FIELD_BWD_SKIP=$k timeout 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --kernel-table --profile-steps 5 --no-graph 2>&1 >/dev/null | grep -E "field_mlp" | tee -a $OUT/summary.txt
done
echo "== bench default" | tee -a $OUT/summary.txt
timeout 600 python bench.py --kernel-table > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
cat $OUT/bench_default.json | tee -a $OUT/summary.txt
head -n 24 $OUT/bench_default_kernel_table.log | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
