#!/bin/bash
# Round 5, GPU session 24: NgpEngine without the per-iteration gradient binding (the schedule writes into the arena's views)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s24
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pipeline_seam.py tests/test_gpu_packed.py -k "ngp" -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc $? $(tail -1 $out/pytest.log)"; grep -E "^E  |^FAILED" $out/pytest.log | head
timeout 200 python scripts/bench_ngp_seam.py > $out/bench_ngp_seam.json 2>/dev/null
echo "ngp seam rc $?"; cut -c1-420 $out/bench_ngp_seam.json
