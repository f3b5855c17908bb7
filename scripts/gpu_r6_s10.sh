#!/bin/bash
# Round 6, GPU session 10: (1) the default bench.py run with its new `secondary` object; (2) the level-1 table-gradient outlier of
# the N(0, 0.3) bench-size parity arm under three builds / routes (shipped, NSAMD_FUSE_ROUTE=0, round-to-nearest fixed point);
# (3) the GPU suite on the current tree (new: two ranks on one GPU).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s10
mkdir -p $out
export TMPDIR=/tmp
cd $R
echo "== default bench" > $out/summary.txt
( time timeout 600 python bench.py ) > $out/bench_default.json 2> $out/bench_default.err
tail -3 $out/bench_default.err >> $out/summary.txt
python - >> $out/summary.txt <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r6_s10/bench_default.json") if l.startswith("{")][-1])
print(j["ms_per_step"], j["value"], j.get("long_run"))
print(json.dumps(j.get("secondary"), indent=1))
print(j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["cpu_baseline"]["value"])
PY
for arm in shipped noroute round; do
  unset NSAMD_LIB NSAMD_FUSE_ROUTE
  [ $arm = noroute ] && export NSAMD_FUSE_ROUTE=0
  [ $arm = round ] && export NSAMD_LIB=$R/nerfstudio_amd/libnsamd_round.so
  echo "== level table, arm $arm" >> $out/summary.txt
  timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -x -s -k "bench_configuration and 0.3" > $out/parity_$arm.log 2>&1
  tail -1 $out/parity_$arm.log >> $out/summary.txt
  grep "hash_table\[level" $out/parity_$arm.log >> $out/summary.txt
done
unset NSAMD_LIB NSAMD_FUSE_ROUTE
echo "== pytest -m gpu" >> $out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x -s --deselect tests/test_gpu_training.py::test_psnr_on_procedural_scene_matches_oracle_training > $out/pytest_gpu.log 2>&1
tail -2 $out/pytest_gpu.log >> $out/summary.txt
grep -E "^E  |^FAILED|^ERROR|two RCCL ranks" $out/pytest_gpu.log | head -20 >> $out/summary.txt
cat $out/summary.txt
