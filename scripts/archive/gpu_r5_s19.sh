#!/bin/bash
# Round 5, GPU session 19: the data-parallel schedule takes its draws from the step prologue — the one-rank RCCL run must train
# through the single-GPU bits again
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s19
mkdir -p $out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_training.py -k "data_parallel or reproducible" -q -m gpu > $out/pytest.log 2>&1
echo "pytest: rc $? $(tail -1 $out/pytest.log)"
grep -E "^E  |^FAILED" $out/pytest.log | head
