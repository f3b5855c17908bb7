#!/bin/bash
# Round 6, GPU session 4: depth of the weight-gradient loops (operand reads in flight) and static wave priority — same-box A/B on real
# buffers (scripts/probe_field_bwd_real.py), two rounds.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s4
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
for v in default u42 u82 u44 u84 prio dw2 u82dw2; do
  if [ $v = default ]; then unset NSAMD_LIB; else export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_$v.so; fi
  timeout 120 python scripts/probe_field_bwd_real.py 30 2>&1 | grep "^lib\|Error\|error" | tail -2
done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
