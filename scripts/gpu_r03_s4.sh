#!/bin/bash
# Round-3 GPU session 4: parity tests, packed-path tests (wave-per-ray marcher), per-ray-mask gating A/B, DP fork A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s4}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== new tests + packed tests" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_packed.py -m gpu -q -s > $OUT/pytest_new.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |bench-size parity|Error|gpu-f64|cpu32-f64|per-sample stages|excluded" $OUT/pytest_new.log | grep -v "hash_table\[level" | cut -c1-400 | head -120 | tee -a $OUT/summary.txt
echo "== per-kind iteration times: gated (default) / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -n 1 | tee -a $OUT/summary.txt; done
echo "== bench driver window, gated / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table 2> $OUT/ktable_gate$g.log | cut -c1-200 | tee -a $OUT/summary.txt; grep -E "gate|density_mlp_bwd|encode_bwd" $OUT/ktable_gate$g.log | tee -a $OUT/summary.txt; done
echo "== 300 steps gated / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt; done
echo "== data-parallel rehearsal over a one-rank RCCL communicator" | tee -a $OUT/summary.txt
for cfg in "n1_graph::" "n1_eager::--no-graph" "dp_fork:NSAMD_DP_FORK=1:--force-dp" "dp_nofork:NSAMD_DP_FORK=0:--force-dp" \
           "dp_fork_sharded:NSAMD_DP_FORK=1:--force-dp --dp-mode sharded"; do
  label=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; flags=${rest#*:}
  env $envs timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --profile-steps 1 --param-checksum $flags > $OUT/dp_$label.json 2> $OUT/dp_$label.err
  echo "$label: rc=$? $(grep '^{' $OUT/dp_$label.json | tail -n 1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], c["final_loss"], c["param_checksum"]["params"][:12], c.get("launch"), c.get("dp_mode"))
except Exception as e: print("no json", e)')" | tee -a $OUT/summary.txt
  grep -E "Error|Traceback" -A3 $OUT/dp_$label.err | tail -n 12 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
