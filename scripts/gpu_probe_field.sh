#!/bin/bash
# GPU box: field backward after the dgrad/dW reorder: clocks, parity (incl. reproducibility), bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/probe_field; mkdir -p $O
timeout 200 python scripts/probe_field_clocks.py --no-build 2>&1 | grep -v "it [1-4]:\|tile" | tee $O/clocks_reorder.log | head -40
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -q -x -k "field or pipeline or reproducible or train_step" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --kernel-table 2>&1 | grep -E "ms_per_step|field_mlp" | cut -c1-190
