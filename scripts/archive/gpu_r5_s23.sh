#!/bin/bash
# Round 5, GPU session 23: the instant-ngp iteration through its pipeline seam against the direct trainer; the mirror classes
# rewritten in this session (vanilla NeRF field, proposal field) under their GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s23
mkdir -p $out
export TMPDIR=/tmp
timeout 200 python scripts/bench_ngp_seam.py > $out/bench_ngp_seam.json 2> $out/bench_ngp_seam.err
echo "ngp seam rc $?"; cat $out/bench_ngp_seam.json; grep -v amdgpu $out/bench_ngp_seam.err | tail -3
timeout 300 python -m pytest tests/test_gpu_vanilla.py tests/test_gpu_kernels.py -k "vanilla or density or proposal or embedding" -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc $? $(tail -1 $out/pytest.log)"; grep -E "^E  |^FAILED" $out/pytest.log | head
