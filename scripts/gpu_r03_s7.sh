#!/bin/bash
# Round-3 GPU session 7: the whole GPU suite + smoke + driver-window bench at the current commit; per-kind iteration times.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s7}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== pytest -m gpu (all)" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |^FAILED|s call|s setup" $OUT/pytest_gpu.log | cut -c1-250 | head -50 | tee -a $OUT/summary.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee -a $OUT/summary.txt
echo "== bench, driver window (with cpu_baseline)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --kernel-table > $OUT/bench_driver_window.json 2> $OUT/bench_driver_window_kernel_table.log
cat $OUT/bench_driver_window.json | tee -a $OUT/summary.txt
head -n 28 $OUT/bench_driver_window_kernel_table.log | tee -a $OUT/summary.txt
echo "== per-kind iteration times (steps 12..111 / steps 5..24)" | tee -a $OUT/summary.txt
timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -n 1 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
