#!/bin/bash
# GPU box: hash-forward variants (NSAMD_HASH_FWD_MODE: 1 pair gathers, 2 XCD-aware level sweep, 3 both): parity + timing
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/probe_hash; mkdir -p $O
for m in 0 1 2 3; do
  echo "== mode $m"
  NSAMD_HASH_FWD_MODE=$m timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "hashgrid or pipeline_vs_oracle_full or field_golden" 2>&1 | tail -1
  NSAMD_HASH_FWD_MODE=$m timeout 300 python bench.py --no-cpu-baseline --kernel-table 2>&1 | grep -E "ms_per_step|hashgrid_encode_fwd|field_mlp_fwd" | cut -c1-170
done 2>&1 | tee $O/modes.log
