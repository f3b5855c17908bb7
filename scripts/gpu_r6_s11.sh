#!/bin/bash
# Round 6, GPU session 11: the GPU suite on the pruned tree (all tests, no -x), smoke(), the longest test apart.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s11
mkdir -p $out
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_training.py::test_psnr_on_procedural_scene_matches_oracle_training > $out/pytest_gpu.log 2>&1
echo "pytest rc $? $(tail -1 $out/pytest_gpu.log)" > $out/summary.txt
grep -E "^E  |^FAILED|^ERROR|two RCCL ranks" $out/pytest_gpu.log | cut -c1-300 | head -40 >> $out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $out/summary.txt 2>&1
cat $out/summary.txt
