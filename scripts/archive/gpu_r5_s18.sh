#!/bin/bash
# Round 5, GPU session 18: the whole GPU suite + smoke + the driver's default bench command on the current tree
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s18
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
timeout 780 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.log 2>&1
el "pytest: rc $? $(tail -1 $out/pytest_gpu.log)"
grep -E "^E  |^FAILED|^ERROR" $out/pytest_gpu.log | head -30
grep -E "GPU - oracle" $out/pytest_gpu.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
el smoke
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
el "bench rc $?"
cat $out/bench_default.json
