#!/bin/bash
# GPU-box session that produces the round's committed evidence: full GPU test log, smoke, PMC passes + traffic file, the
# default and driver-window bench lines (with cpu_baseline), per-kernel tables, the 300-step / steady-state / unbounded /
# instant-ngp lines, rocprofv3 kernel stats, per-kind iteration times, eval render, the one-rank data-parallel rehearsal.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_final}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "mean PSNR|^   [0-9] \||GPU - oracle|passed|failed|^E  |bench-size parity|excluded" $OUT/pytest_gpu.log | cut -c1-400 | head -40 | tee -a $OUT/summary.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee -a $OUT/summary.txt
echo "== PMC passes + rocprofv3 stats (scripts/collect_pmc.sh)" | tee -a $OUT/summary.txt
bash scripts/collect_pmc.sh $TAG > $OUT/collect_pmc.log 2>&1
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json 2>/dev/null
head -n 45 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
echo "== bench, driver window (traffic now stamped for these sources)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --kernel-table > $OUT/bench_driver_window.json 2> $OUT/bench_driver_window_kernel_table.log
cat $OUT/bench_driver_window.json | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/bench_driver_window_kernel_table.log | head -n 26 | tee -a $OUT/summary.txt
echo "== bench default (30 steps)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>/dev/null
cut -c1-260 $OUT/bench_default.json | tee -a $OUT/summary.txt
echo "== bench 300 steps" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --kernel-table --profile-steps 10 > $OUT/bench_300steps.json 2> $OUT/bench_300steps_kernel_table.log
cut -c1-260 $OUT/bench_300steps.json | tee -a $OUT/summary.txt
echo "== bench --start-step 5000 (steady state of the proposal update schedule; not the headline)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 120 --warmup 12 --start-step 5000 --no-cpu-baseline > $OUT/bench_start5000.json 2>/dev/null
cut -c1-260 $OUT/bench_start5000.json | tee -a $OUT/summary.txt
echo "== per-kind iteration times (scripts/probe_iteration_times.py, steps 12..111)" | tee -a $OUT/summary.txt
timeout 300 python scripts/probe_iteration_times.py 2>/dev/null | tail -n 1 | tee -a $OUT/summary.txt
echo "== bench unbounded" | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload unbounded --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_unbounded.json 2>/dev/null
cut -c1-260 $OUT/bench_unbounded.json | tee -a $OUT/summary.txt
echo "== bench ngp" | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload ngp --steps 30 --warmup 5 --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_kernel_table.log
cat $OUT/bench_ngp.json | tee -a $OUT/summary.txt
grep -v "amdgpu.ids\|Warning\|detach\|final_loss" $OUT/bench_ngp_kernel_table.log | head -n 16 | tee -a $OUT/summary.txt
echo "== eval render 800x800: device-side chunk loop / module loop" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_render.py 2>/dev/null | tail -n 1 | tee $OUT/bench_render_device_loop.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_render.py --module-loop 2>/dev/null | tail -n 1 | tee $OUT/bench_render_module_loop.json | tee -a $OUT/summary.txt
echo "== data-parallel rehearsal over a one-rank RCCL communicator" | tee -a $OUT/summary.txt
for cfg in "n1_graph::" "dp_allreduce::--force-dp" "dp_sharded::--force-dp --dp-mode sharded"; do
  label=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; flags=${rest#*:}
  env $envs timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --profile-steps 1 --param-checksum $flags > $OUT/dp_$label.json 2> $OUT/dp_$label.err
  echo "$label: $(grep '^{' $OUT/dp_$label.json | tail -n 1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], c["final_loss"], c["param_checksum"]["params"][:12], c.get("dp_mode"), c.get("rccl_ranks"))
except Exception as e: print("no json", e)')" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
