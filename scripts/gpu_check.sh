#!/bin/bash
# GPU box: all GPU tests, smoke, then the default bench line with the per-kernel table
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-check}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee $O/pytest_gpu.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 | tee $O/smoke.log
timeout 300 python bench.py --kernel-table > $O/bench.json 2> $O/bench_table.log
cut -c1-230 $O/bench.json; head -8 $O/bench_table.log
