#!/bin/bash
# Round 6, GPU session 5: timing prototype of two half-workgroups with their own (LDS-counter) barriers in the field backward
# (NSAMD_BWD_GROUPS = barriers of the first group the second one waits for before it starts). Results of these builds are wrong
# (the weight-gradient share still reads all eight scratch areas); the question is what de-synchronised waves buy.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s5
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for i in 1 2; do
for v in default grp0 grp3 grp5 grp8; do
  if [ $v = default ]; then unset NSAMD_LIB; else export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_$v.so; fi
  timeout 120 python scripts/probe_field_bwd_real.py 30 2>&1 | grep "^lib\|Error\|error" | tail -2
done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
