#!/bin/bash
# GPU box: all GPU tests, then the default bench line with the per-kernel table
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-check}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee $O/pytest_gpu.log | tail -6
timeout 300 python bench.py --no-cpu-baseline --kernel-table > $O/bench.json 2> $O/bench_table.log
cut -c1-230 $O/bench.json; head -26 $O/bench_table.log
