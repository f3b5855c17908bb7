#!/bin/bash
# Round-3 GPU session 3: parity tests (float64 arbiter + kernel-level field backward), gating A/B with the plain flag store,
# data-parallel rehearsal with errors logged.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s3}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== new tests" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -s > $OUT/pytest_new.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |bench-size parity|Error|gpu-f64|cpu32-f64" $OUT/pytest_new.log | cut -c1-300 | head -120 | tee -a $OUT/summary.txt
echo "== per-kind iteration times: gated (default) / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -n 1 | tee -a $OUT/summary.txt; done
echo "== bench driver window, gated / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table 2> $OUT/ktable_gate$g.log | cut -c1-200 | tee -a $OUT/summary.txt; grep -E "gate|density_mlp_bwd|encode_bwd" $OUT/ktable_gate$g.log | tee -a $OUT/summary.txt; done
echo "== data-parallel rehearsal over a one-rank RCCL communicator (ms/step, checksum)" | tee -a $OUT/summary.txt
for cfg in "n1_graph::" "n1_eager::--no-graph" "dp_eager_updstream:NSAMD_DP_UPDATE_STREAM=1:--force-dp" "dp_eager_noupdstream:NSAMD_DP_UPDATE_STREAM=0:--force-dp" \
           "dp_graphsegs:NSAMD_DP_UPDATE_STREAM=1:--force-dp --dp-graph" "dp_sharded:NSAMD_DP_UPDATE_STREAM=1:--force-dp --dp-mode sharded"; do
  label=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; flags=${rest#*:}
  env $envs timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --profile-steps 1 --param-checksum $flags > $OUT/dp_$label.json 2> $OUT/dp_$label.err
  echo "$label: rc=$? $(tail -n 1 $OUT/dp_$label.json | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["final_loss"], d["config"]["param_checksum"]["params"][:12], d["config"].get("launch"), d["config"].get("dp_mode"))
except Exception as e: print("no json", e)')" | tee -a $OUT/summary.txt
  grep -E "Error|error|Traceback" -A3 $OUT/dp_$label.err | tail -n 12 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
