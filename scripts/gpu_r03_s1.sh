#!/bin/bash
# Round-3 GPU session 1: new parity tests + full GPU suite, driver-window bench, gating A/B, saved-vs-recompute A/B,
# rocprofv3 kernel stats of the replayed graphs. Everything lands under gpurun_out/r03_s1/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_s1}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== new tests" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -m gpu -q -s > $OUT/pytest_new.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |bench-size parity|Error" $OUT/pytest_new.log | head -40 | tee -a $OUT/summary.txt
echo "== pytest -m gpu (rest)" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_parity.py > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_gpu.log | head -30 | tee -a $OUT/summary.txt
echo "== bench, driver window" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --kernel-table > $OUT/bench_driver_window.json 2> $OUT/bench_driver_window_kernel_table.log
cat $OUT/bench_driver_window.json | tee -a $OUT/summary.txt
head -n 30 $OUT/bench_driver_window_kernel_table.log | tee -a $OUT/summary.txt
echo "== per-kind iteration times: gated (default) / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python scripts/probe_iteration_times.py 2>&1 | tail -n 1 | tee -a $OUT/summary.txt; done
echo "== bench default window, gated / ungated" | tee -a $OUT/summary.txt
for g in 1 0; do NSAMD_GATE_PROPOSALS=$g timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt; done
echo "== field backward: recompute (default) vs saved activations, eager kernel table" | tee -a $OUT/summary.txt
for s in 0 1; do
  NSAMD_FIELD_SAVE_ACTS=$s timeout 300 python bench.py --no-cpu-baseline --kernel-table --steps 30 > $OUT/bench_saveacts$s.json 2> $OUT/bench_saveacts${s}_kernel_table.log
  cut -c1-160 $OUT/bench_saveacts$s.json | tee -a $OUT/summary.txt
  grep -E "field_mlp|hashgrid_encode_fwd\[L=16" $OUT/bench_saveacts${s}_kernel_table.log | tee -a $OUT/summary.txt
done
echo "== rocprofv3 kernel stats (graph replay, 20 steps)" | tee -a $OUT/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kstats -o k -- python $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --profile-steps 1 > $OUT/rocprof_bench.log 2>&1
cd $R
OUT=$OUT python - <<'PY' | tee -a $OUT/summary.txt
import glob, os, sqlite3
out = os.environ["OUT"]
dbs = glob.glob("/tmp/kstats/**/*results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t]
    try:
        rows = db.execute("select name, grid_x*grid_y*grid_z, workgroup_x, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 "
                          "from kernels group by name, grid_x, grid_y, workgroup_x order by 6 desc").fetchall()
    except Exception as e:
        print("kernels view missing:", e, tabs[:20]); rows = []
    tot = sum(r[5] for r in rows) or 1.0
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        f.write("kernel,grid_threads,workgroup,calls,avg_us,total_us,percent\n")
        for r in rows:
            f.write(f"\"{r[0][:90]}\",{r[1]},{r[2]},{r[3]},{r[4]:.2f},{r[5]:.1f},{100*r[5]/tot:.2f}\n")
    print(open(os.path.join(out, "kernel_stats.csv")).read()[:4000])
PY
echo "== done" | tee -a $OUT/summary.txt
