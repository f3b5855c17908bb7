#!/bin/bash
# Round 5, GPU session 17: lane-pair hash forward with two points per lane (NSAMD_HASH_FWD_MODE=15) against one (7)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s17
mkdir -p $out
export TMPDIR=/tmp
NSAMD_HASH_FWD_MODE=15 timeout 300 python -m pytest tests/test_gpu_kernels.py -k "hashgrid or nerfacto_field_golden" -q -m gpu > $out/pytest_mode15.log 2>&1
echo "mode 15 pytest: rc $? $(tail -1 $out/pytest_mode15.log)"
for mode in 7 15 7 15; do
  echo "== NSAMD_HASH_FWD_MODE=$mode" >> $out/hash_levels.txt
  NSAMD_HASH_FWD_MODE=$mode timeout 120 python scripts/probe_hash_levels.py --res 16,58,111,2048 >> $out/hash_levels.txt 2>/dev/null
done
cat $out/hash_levels.txt
