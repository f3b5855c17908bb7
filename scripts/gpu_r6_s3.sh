#!/bin/bash
# Round 6, GPU session 3: attribution of the record-emitting field backward on REAL buffers (scripts/probe_field_bwd_real.py)
# with compile-time switches in the product kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s3
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for v in default dw0 dw2 skip1 skip2 skip4 skip32 skip36 skip37 skip64 skip128 skip101 skip229 skip231 default; do
  if [ $v = default ]; then unset NSAMD_LIB; else export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_$v.so; fi
  timeout 120 python scripts/probe_field_bwd_real.py 30 2>&1 | grep "^lib\|Error\|error" | tail -2
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
