#!/bin/bash
# Round 6, GPU session 1: weight-gradient GEMMs of the main field's backward on the bf16 matrix cores (two-piece operands).
# Parity of the new default build (NSAMD_DW_BF16=1) and of the all-five-layers build (=2) at the benchmark's size against
# float64, then same-box A/B of the per-kernel table and the driver window: dw0 (f32, rounds 2-5) / new / dw2.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s1
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
echo "== field tests, default build"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "field or mlp or saved or fused_main" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -x -s -k "field_mlp_backward or emits_the_scatter or bench_configuration" 2>&1 | grep -v "^$" | tail -60
echo "== bench-size float64 test, all five layers bf16 (dw2)"
NSAMD_LIB=$R/nerfstudio_amd/libnsamd_dw2.so timeout 600 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -x -s -k "field_mlp_backward" 2>&1 | grep -v "^$" | tail -20
for i in 1 2; do
  for arm in dw0 new dw2; do
    if [ $arm = new ]; then unset NSAMD_LIB; else export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_$arm.so; fi
    echo "== $arm"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --param-checksum --kernel-table --profile-steps 10 2> $out/table_${arm}_$i.log | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['roofline'], j['config'].get('param_checksum',{}).get('params'))"
    grep "field_mlp_bwd\|field_mlp_fwd\|scatter_apply" $out/table_${arm}_$i.log | cut -c1-150
  done
done
unset NSAMD_LIB
} > $out/summary.txt 2>&1
cat $out/summary.txt
