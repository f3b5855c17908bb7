#!/bin/bash
# Round 6, GPU session 15: DPP row sums instead of ds_bpermute butterflies in the field backward's appearance-gradient rows:
# bit tests (the rows' lane-0 sums are the butterfly's), then the launch alone on real buffers, prev / new alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s15
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_parity.py tests/test_gpu_fused_launches.py -m gpu -q -x -k "field or riding or emits or backward_at_bench" 2>&1 | tail -3
for i in 1 2 3; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_prev.so; else unset NSAMD_LIB; fi
    timeout 120 python scripts/probe_field_bwd_real.py 30 2>&1 | grep "^lib" | cut -c1-130
  done
done
unset NSAMD_LIB
for arm in prev new; do
  if [ $arm = prev ]; then export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_prev.so; else unset NSAMD_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$arm window', j['ms_per_step'], j['value'], j['config']['param_checksum']['params'])"
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
