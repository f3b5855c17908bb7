#!/bin/bash
# GPU-box session that produces the round's committed evidence, most important first, every step under its own timeout and
# the optional tail under an elapsed-time guard (BUDGET_S, default 870 s: the steps after it are skipped, not cut off):
# full GPU test log, smoke, rocprofv3 kernel stats + PMC passes + traffic file, the driver-window bench line (with
# cpu_baseline and the per-kernel table), then the instant-ngp / 300-step / steady-state / unbounded lines, eval render, the
# one-rank data-parallel rehearsal.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_final}
OUT=$R/gpurun_out/$TAG
BUDGET_S=${BUDGET_S:-1700}
T0=$(date +%s)
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
left() { [ $(( $(date +%s) - T0 )) -lt $BUDGET_S ]; }
say() { echo "$@" | tee -a $OUT/summary.txt; }
say "== pytest -m gpu"
timeout 1300 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
say "rc=$?"
grep -E "mean PSNR|^   [0-9] \||GPU - oracle|passed|failed|^E  |bench-size parity|excluded" $OUT/pytest_gpu.log | cut -c1-400 | head -40 | tee -a $OUT/summary.txt
say "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee -a $OUT/summary.txt
say "== PMC passes + rocprofv3 stats (scripts/collect_pmc.sh)"
PMC_STATS_TIMEOUT=150 PMC_PASS_TIMEOUT=120 bash scripts/collect_pmc.sh $TAG > $OUT/collect_pmc.log 2>&1
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json 2>/dev/null
head -n 45 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
say "== bench, driver window (traffic now stamped for these sources)"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench_driver_line.err
cat $OUT/bench_driver_line.json | tee -a $OUT/summary.txt
timeout 200 python bench.py --steps 20 --warmup 5 --kernel-table --no-cpu-baseline > $OUT/bench_driver_window.json 2> $OUT/bench_driver_window_kernel_table.log
cat $OUT/bench_driver_window.json | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/bench_driver_window_kernel_table.log | head -n 26 | tee -a $OUT/summary.txt
say "elapsed $(( $(date +%s) - T0 )) s"
if left; then
say "== the iteration through the Trainer / Pipeline seam next to the direct line (scripts/bench_seam.py)"
timeout 200 python scripts/bench_seam.py > $OUT/bench_seam.json 2> $OUT/bench_seam.err
python -c "
import json; j=json.load(open('$OUT/bench_seam.json'))
print({k: j[k] for k in ('direct_pool_ms','direct_set_batch_ms','seam_ms','seam_over_direct_pool','seam_over_direct_pool_per_window')})" | tee -a $OUT/summary.txt
fi
if left; then
say "== bench, camera optimiser ON (SO3xR3: the reference's nerfacto default, models/nerfacto.py:131)"
timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --camera-optimizer SO3xR3 --kernel-table > $OUT/bench_camera_SO3xR3.json 2> $OUT/bench_camera_SO3xR3_kernel_table.log
cut -c1-400 $OUT/bench_camera_SO3xR3.json | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/bench_camera_SO3xR3_kernel_table.log | head -n 8 | tee -a $OUT/summary.txt
fi
if left; then
say "== bench ngp (explicit schedule)"
timeout 150 python bench.py --workload ngp --steps 30 --warmup 10 --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_kernel_table.log
cat $OUT/bench_ngp.json | tee -a $OUT/summary.txt
grep -v "amdgpu.ids\|Warning\|detach\|final_loss" $OUT/bench_ngp_kernel_table.log | head -n 12 | tee -a $OUT/summary.txt
fi
if left; then
say "== per-kind iteration times (scripts/probe_iteration_times.py, steps 12..111)"
timeout 100 python scripts/probe_iteration_times.py 2>/dev/null | tail -n 1 | tee -a $OUT/summary.txt
fi
if left; then
say "== bench 300 steps"
timeout 120 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --kernel-table --profile-steps 10 > $OUT/bench_300steps.json 2> $OUT/bench_300steps_kernel_table.log
cut -c1-260 $OUT/bench_300steps.json | tee -a $OUT/summary.txt
fi
if left; then
say "== bench --start-step 5000 (steady state of the proposal update schedule; not the headline)"
timeout 100 python bench.py --steps 120 --warmup 12 --start-step 5000 --no-cpu-baseline > $OUT/bench_start5000.json 2>/dev/null
cut -c1-260 $OUT/bench_start5000.json | tee -a $OUT/summary.txt
fi
if left; then
say "== data-parallel rehearsal over a one-rank RCCL communicator"
for cfg in "n1_graph::" "dp_allreduce::--force-dp" "dp_sharded::--force-dp --dp-mode sharded"; do
  label=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; flags=${rest#*:}
  left || break
  env $envs timeout 100 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --profile-steps 1 --param-checksum $flags > $OUT/dp_$label.json 2> $OUT/dp_$label.err
  say "$label: $(grep '^{' $OUT/dp_$label.json | tail -n 1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], c["final_loss"], c["param_checksum"]["params"][:12], c.get("dp_mode"), c.get("rccl_ranks"))
except Exception as e: print("no json", e)')"
done
fi
if left; then
say "== bench unbounded"
timeout 100 python bench.py --workload unbounded --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_unbounded.json 2>/dev/null
cut -c1-260 $OUT/bench_unbounded.json | tee -a $OUT/summary.txt
fi
if left; then
say "== eval render 800x800: device-side chunk loop / module loop"
timeout 100 python scripts/bench_render.py 2>/dev/null | tail -n 1 | tee $OUT/bench_render_device_loop.json | tee -a $OUT/summary.txt
left && timeout 100 python scripts/bench_render.py --module-loop 2>/dev/null | tail -n 1 | tee $OUT/bench_render_module_loop.json | tee -a $OUT/summary.txt
fi
say "== done after $(( $(date +%s) - T0 )) s"
