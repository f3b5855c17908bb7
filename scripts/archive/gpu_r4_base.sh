#!/bin/bash
# round-4 evidence on HEAD: all GPU tests, smoke, default bench line + kernel table, camera-optimiser line, ngp line
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${1:-r4_base}; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 | tee $OUT/smoke.log
timeout 300 python bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
grep -v amdgpu.ids $OUT/bench_table.log | head -n 40
timeout 200 python bench.py --no-cpu-baseline --camera-optimizer SO3xR3 --kernel-table > $OUT/bench_cam.json 2> $OUT/bench_cam.log; echo "bench cam rc=$?"; cut -c1-400 $OUT/bench_cam.json
grep -v amdgpu.ids $OUT/bench_cam.log | head -n 12
timeout 200 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_table.log; echo "bench ngp rc=$?"; cut -c1-300 $OUT/bench_ngp.json
