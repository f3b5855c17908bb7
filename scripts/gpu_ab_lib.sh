#!/bin/bash
# Same-box A/B of two builds of the library (NSAMD_LIB): nerfstudio_amd/libnsamd_prev.so (built from an older tree by hand)
# against the current libnsamd.so; alternating arms. Per-kind iteration times inside the loop + the driver window.
tag=${1:-ab_lib}
out=gpurun_out/$tag
mkdir -p $out
P=$PWD/nerfstudio_amd/libnsamd_prev.so
{
for i in 1 2; do
  echo "== prev"; NSAMD_LIB=$P PROBE_STEPS=20 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1
  echo "== new";  PROBE_STEPS=20 timeout 200 python scripts/probe_iteration_times.py 2>/dev/null | tail -1
done
for i in 1 2; do
  for arm in prev new; do
    if [ $arm = prev ]; then export NSAMD_LIB=$P; else unset NSAMD_LIB; fi
    echo "== driver window, $arm"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params'))"
  done
done
unset NSAMD_LIB
} > $out/summary.txt 2>&1
cat $out/summary.txt
