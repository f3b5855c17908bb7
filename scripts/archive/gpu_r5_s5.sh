#!/bin/bash
# Round 5, GPU session 5: the per-ray kernels with the wave index as a scalar — tests, same-box A/B against -DNSAMD_SCALAR_RAY=0.
out=gpurun_out/r5_s5
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
timeout 600 python -m pytest tests/test_gpu_fused_launches.py tests/test_gpu_kernels.py -x -q -m gpu > $out/pytest_kernels.log 2>&1
el "pytest fused + kernels: rc $? $(tail -1 $out/pytest_kernels.log)"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], j['config'].get('param_checksum',{}).get('params'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --param-checksum --profile-steps 1"
{
for i in 1 2 3 4; do
  echo "== default (scalar wave index)"; timeout 200 $B 2>/dev/null | line
  echo "== vector wave index";           NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_vecray.so timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
K="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --windows 3 --kernel-table --profile-steps 4"
timeout 200 $K > /dev/null 2> $out/table_scalar.log
NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_vecray.so timeout 200 $K > /dev/null 2> $out/table_vector.log
for f in scalar vector; do echo "== $f"; grep -E "select_bins|proposal_resample|proposal_losses|render_train|weights_bwd" $out/table_$f.log; done
el end
