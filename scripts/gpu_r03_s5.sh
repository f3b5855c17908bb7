#!/bin/bash
# Round-3 GPU session 5: tests (parity with conditioning bound, packed, ngp sampler contract), rocprofv3 kernel trace of the
# SPARSE regime (eager, one stream), ngp workload, gating + DP A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s5}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== tests" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_packed.py tests/test_ngp_sampler_contract.py -m gpu -q -s > $OUT/pytest_new.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |Error|per-sample stages|excluded" $OUT/pytest_new.log | grep -v "hash_table\[level" | cut -c1-300 | head -60 | tee -a $OUT/summary.txt
echo "== ngp workload" | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload ngp --steps 30 --warmup 5 --kernel-table > $OUT/bench_ngp.json 2> $OUT/bench_ngp_kernel_table.log
echo "rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench_ngp.json | cut -c1-3000 | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/bench_ngp_kernel_table.log | head -40 | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel stats, eager, one stream, steps 8..27 (sparse proposal gradients), gated" | tee -a $OUT/summary.txt
cd /tmp
NSAMD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ksparse -o k -- python $R/bench.py --no-graph --steps 20 --warmup 8 --no-cpu-baseline --profile-steps 1 > $OUT/rocprof_sparse.log 2>&1
cd $R
python - <<'PY' | tee -a $OUT/summary.txt
import glob, sqlite3
dbs = glob.glob("/tmp/ksparse/**/*results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select name, grid_x*grid_y*grid_z, workgroup_x, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0, sum(end-start)/1000.0 "
                      "from kernels group by name, grid_x, grid_y, workgroup_x order by 8 desc").fetchall()
    print("kernel,grid_threads,workgroup,calls,avg_us,min_us,max_us,total_us")
    for r in rows[:45]:
        print(f"\"{r[0][:80]}\",{r[1]},{r[2]},{r[3]},{r[4]:.2f},{r[5]:.2f},{r[6]:.2f},{r[7]:.1f}")
PY
echo "== per-kind iteration times: gated (default) / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -n 1 | tee -a $OUT/summary.txt; done
echo "== bench driver window + 300 steps, gated / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-180 | tee -a $OUT/summary.txt; NSAMD_GATE_PROPOSALS=$g timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-180 | tee -a $OUT/summary.txt; done
echo "== data-parallel rehearsal over a one-rank RCCL communicator" | tee -a $OUT/summary.txt
for cfg in "n1_graph::" "n1_eager::--no-graph" "dp_coalesced:NSAMD_COALESCE_ALLREDUCE=1:--force-dp" "dp_separate:NSAMD_COALESCE_ALLREDUCE=0:--force-dp" \
           "dp_sharded::--force-dp --dp-mode sharded" "dp_coalesced_updstream:NSAMD_DP_UPDATE_STREAM=1:--force-dp"; do
  label=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; flags=${rest#*:}
  env $envs timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --profile-steps 1 --param-checksum $flags > $OUT/dp_$label.json 2> $OUT/dp_$label.err
  echo "$label: rc=$? $(grep '^{' $OUT/dp_$label.json | tail -n 1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], c["final_loss"], c["param_checksum"]["params"][:12], c.get("launch"), c.get("dp_mode"))
except Exception as e: print("no json", e)')" | tee -a $OUT/summary.txt
  grep -E "Error|Traceback" -A3 $OUT/dp_$label.err | tail -n 12 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
