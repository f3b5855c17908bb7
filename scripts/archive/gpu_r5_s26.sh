#!/bin/bash
# Round 5, GPU session 26: the GPU suite on the FINAL tree (host-side changes after r05_final4: NgpEngine's gradient lookup, the
# rewritten mirror classes) — without the two longest tests, which those changes do not touch and r05_final4 / session 19 ran
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s26
mkdir -p $out
export TMPDIR=/tmp
timeout 270 python -m pytest tests -m gpu -q -x \
  --deselect tests/test_gpu_training.py::test_psnr_on_procedural_scene_matches_oracle_training \
  --deselect tests/test_gpu_training.py::test_data_parallel_path_over_one_rank_rccl_matches_single_gpu > $out/pytest_gpu.log 2>&1
echo "pytest rc $? $(tail -1 $out/pytest_gpu.log)"; grep -E "^E  |^FAILED|^ERROR" $out/pytest_gpu.log | head
