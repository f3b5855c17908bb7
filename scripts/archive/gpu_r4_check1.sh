#!/bin/bash
# round-4 checkpoint 1: the refactored trainer / seam / ngp parity on the GPU, then the new bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r4_c1; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_pipeline_seam.py "tests/test_gpu_packed.py::test_ngp_bench_size_parity_vs_oracle" "tests/test_gpu_kernels.py::test_samplers_golden_bit_exact" tests/test_gpu_bench_parity.py -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|error|assert|parity" $OUT/pytest.log | tail -n 40 | cut -c1-400
timeout 200 python bench.py --no-cpu-baseline --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log; echo "bench rc=$?"; cat $OUT/bench.json | cut -c1-3000
grep -v amdgpu.ids $OUT/bench_table.log | head -n 12
timeout 200 python bench.py --no-cpu-baseline --camera-optimizer SO3xR3 --kernel-table > $OUT/bench_cam.json 2> $OUT/bench_cam_table.log; echo "bench cam rc=$?"; cut -c1-1200 $OUT/bench_cam.json
grep -v amdgpu.ids $OUT/bench_cam_table.log | head -n 6
timeout 200 python bench.py --workload ngp --steps 32 --warmup 10 --no-cpu-baseline --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_table.log; echo "bench ngp rc=$?"; cut -c1-2500 $OUT/bench_ngp.json
grep -v "amdgpu.ids\|Warning" $OUT/bench_ngp_table.log | head -n 24
