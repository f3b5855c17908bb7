#!/bin/bash
# Round-3 GPU session 2: fixed parity tests, proposal-gradient sparsity in the bench's own run, gating A/B after the reduce
# revert, data-parallel rehearsal (one-rank RCCL): update stream on/off, captured segments, sharded mode equality.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s2}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== new tests" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -s > $OUT/pytest_new.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |bench-size parity|Error|gpu-f64" $OUT/pytest_new.log | cut -c1-400 | head -60 | tee -a $OUT/summary.txt
echo "== proposal-gradient sparsity in the bench's run" | tee -a $OUT/summary.txt
timeout 600 python scripts/probe_proposal_sparsity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/proposal_sparsity.txt | tee -a $OUT/summary.txt
echo "== per-kind iteration times: gated (default) / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -n 1 | tee -a $OUT/summary.txt; done
echo "== data-parallel rehearsal over a one-rank RCCL communicator (ms/step, checksum)" | tee -a $OUT/summary.txt
run() { # label, env..., -- flags
  label=$1; shift
  env "$@" 2>/dev/null
}
for cfg in "n1_graph::" "n1_eager::--no-graph" "dp_eager_updstream:NSAMD_DP_UPDATE_STREAM=1:--force-dp" "dp_eager_noupdstream:NSAMD_DP_UPDATE_STREAM=0:--force-dp" \
           "dp_graphsegs:NSAMD_DP_UPDATE_STREAM=1:--force-dp --dp-graph" "dp_sharded:NSAMD_DP_UPDATE_STREAM=1:--force-dp --dp-mode sharded"; do
  label=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; flags=${rest#*:}
  line=$(env $envs timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --profile-steps 1 --param-checksum $flags 2>/dev/null | tail -n 1)
  echo "$label: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["final_loss"], d["config"]["param_checksum"]["params"][:12], d["config"].get("launch"), d["config"].get("dp_mode"))')" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
