#!/bin/bash
# Quick A/B session after a kernel change: the scatter / training bit tests, the gated chains alone (probe build), the driver
# window with the kernel table. -> gpurun_out/<tag>/summary.txt
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p $out
{
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -m gpu -q -x -k "scatter or gated or reproduc or bit or runner_matches or fused_training or golden" 2>&1 | tail -4
echo "== gated chains alone (sparse phase)"
(timeout 200 python scripts/probe_gated_scatter_clocks.py 0; timeout 200 python scripts/probe_gated_scatter_clocks.py 1) 2>&1 | grep -v amdgpu.ids | grep "marked\|chain of\|lifetime\|->"
for i in 1 2; do
echo "== driver window"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum --kernel-table 2> $out/table.log | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'))"
done
grep "encode_bwd\|density_mlp_bwd\|weights_bwd" $out/table.log | cut -c1-130
} > $out/summary.txt 2>&1
cat $out/summary.txt
