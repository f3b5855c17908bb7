#!/bin/bash
# Same-box A/B of environment switches of the library: each argument is one arm ("VAR=value VAR2=value" or "-" for none);
# alternating, twice: the driver window + the main-table scatter's launch time from the kernel table.
out=gpurun_out/ab_env
mkdir -p $out
{
for i in 1 2; do
  for arm in "$@"; do
    [ "$arm" = "-" ] && envs="" || envs="$arm"
    echo "== [$envs]"
    env $envs timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param-checksum --kernel-table --profile-steps 6 2> $out/t.log | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('param_checksum',{}).get('params')[:8])"
    grep "encode_bwd_set\|encode_fwd\[L=16" $out/t.log | cut -c1-130
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
