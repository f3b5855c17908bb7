#!/bin/bash
# Round 6, GPU session 2: clock stamps of the record-emitting field backward after the bf16 weight-gradient change, and the
# attribution switches of the probe build (NSAMD_FIELD_BWD_SKIP: 1 no weight-gradient MFMAs, 2 no barriers, 4 no data-gradient
# GEMMs, 32 no record emission) — what is the kernel made of now?
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s2
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
python scripts/probe_field_clocks.py --route 2>&1 | grep -v amdgpu.ids
for skip in 1 2 4 32 5 7 39; do
  echo "== NSAMD_FIELD_BWD_SKIP=$skip"
  NSAMD_FIELD_BWD_SKIP=$skip python scripts/probe_field_clocks.py --route --no-build 2>&1 | grep "^fwd\|field_mlp_bwd"
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
