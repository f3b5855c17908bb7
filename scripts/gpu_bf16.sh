#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/bf16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "mfma or bf16x3" 2>&1 | tail -12
for f in 0 1 0 1; do echo "== NSAMD_FIELD_FWD_BF16X3=$f"; NSAMD_FIELD_FWD_BF16X3=$f timeout 300 python bench.py --no-cpu-baseline --kernel-table 2>&1 | grep -E "ms_per_step|field_mlp_fwd " | cut -c1-200; done 2>&1 | tee $O/ab.log
NSAMD_FIELD_FWD_BF16X3=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "field or pipeline or runner" 2>&1 | tail -8
