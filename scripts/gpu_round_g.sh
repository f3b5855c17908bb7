#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02g}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== packed-path tests" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_packed.py -m gpu -q -s > $OUT/pytest_packed.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |Error" $OUT/pytest_packed.log | head -40 | tee -a $OUT/summary.txt
echo "== all GPU tests" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "mean PSNR|^   [0-9] \||GPU - oracle|passed|failed|^E  " $OUT/pytest_gpu.log | head -40 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
