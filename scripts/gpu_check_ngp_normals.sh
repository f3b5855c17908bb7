#!/bin/bash
# One short session: the GPU tests of the normals options and of the packed path (incl. the explicit instant-ngp schedule
# against the module path), then the instant-ngp bench line through both routes. -> gpurun_out/<tag>/
tag=${1:-ngp_normals}
out=gpurun_out/$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "== pytest (normals, packed)"
timeout 900 python -m pytest tests/test_normals.py tests/test_gpu_packed.py -m gpu -q 2>&1 | tail -25
echo "== bench --workload ngp (explicit schedule)"
timeout 600 python bench.py --workload ngp --steps 30 --warmup 10 --kernel-table 2> $out/bench_ngp_kernel_table.log | grep '^{' | tee $out/bench_ngp.json | cut -c1-1500
head -45 $out/bench_ngp_kernel_table.log | cut -c1-140
echo "== bench --workload ngp --ngp-module-path"
timeout 600 python bench.py --workload ngp --steps 30 --warmup 10 --no-cpu-baseline --ngp-module-path 2> $out/bench_ngp_module.err | grep '^{' | tee $out/bench_ngp_module.json | cut -c1-400
tail -3 $out/bench_ngp_module.err
} > $out/summary.txt 2>&1
tail -60 $out/summary.txt
