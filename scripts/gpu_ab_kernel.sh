#!/bin/bash
# Same-box A/B of two library builds on the per-kernel table (eager, HIP events) + the loop: prev (libnsamd_prev.so) vs new.
tag=${1:-abk}
pat=${2:-encode_bwd_set}
out=gpurun_out/$tag
mkdir -p $out
P=$PWD/nerfstudio_amd/libnsamd_prev.so
{
echo "== bit tests (new)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -m gpu -q -x -k "scatter or gated or reproduc or bit or runner_matches or golden" 2>&1 | tail -2
for i in 1 2; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$P; else unset NSAMD_LIB; fi
    echo "== $arm"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum --kernel-table --profile-steps 10 2> $out/table_$arm.log | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'))"
    grep "$pat" $out/table_$arm.log | cut -c1-130
    PROBE_STEPS=20 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1
  done
done
unset NSAMD_LIB
} > $out/summary.txt 2>&1
cat $out/summary.txt
