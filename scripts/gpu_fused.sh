#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/fused; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -q -x -k "fused_main or runner or reproducible or pipeline" 2>&1 | tail -3
for f in 0 1 0 1; do echo "== NSAMD_FUSE_MAIN_FWD=$f"; NSAMD_FUSE_MAIN_FWD=$f timeout 300 python bench.py --no-cpu-baseline --kernel-table 2>&1 | grep -E "ms_per_step|field_fused|field_mlp_fwd|hashgrid_encode_fwd" | cut -c1-200; done 2>&1 | tee $O/ab.log
