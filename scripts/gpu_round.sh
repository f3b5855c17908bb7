#!/bin/bash
# One GPU-box session (run through gpurun): tests, the bench line, per-kernel table, scatter variants, rocprofv3 stats.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh <tag> [quick]'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -n 25 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
echo "== bench default" | tee -a $OUT/summary.txt
timeout 600 python bench.py --kernel-table > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench_default.json | tee -a $OUT/summary.txt
head -n 30 $OUT/bench_default_kernel_table.log | tee -a $OUT/summary.txt
if [ "$2" != "quick" ]; then
  for v in 124 122 524 522 514 112; do
    echo "== scatter shape $v" | tee -a $OUT/summary.txt
    NSAMD_SCATTER_SHAPE=$v timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --kernel-table --profile-steps 5 \
      > $OUT/bench_shape$v.json 2> $OUT/bench_shape$v.log
    python -c "import json;d=json.load(open('$OUT/bench_shape$v.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])" | tee -a $OUT/summary.txt
    grep -E "hashgrid_encode_bwd" $OUT/bench_shape$v.log | tee -a $OUT/summary.txt
  done
  echo "== no run levels (NSAMD_SCATTER_COMBINE_RES=1)" | tee -a $OUT/summary.txt
  NSAMD_SCATTER_COMBINE_RES=1 timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --kernel-table --profile-steps 5 \
    > $OUT/bench_norun.json 2> $OUT/bench_norun.log
  grep -E "hashgrid_encode_bwd" $OUT/bench_norun.log | tee -a $OUT/summary.txt
  echo "== unbounded workload" | tee -a $OUT/summary.txt
  timeout 300 python bench.py --workload unbounded --no-cpu-baseline --kernel-table > $OUT/bench_unbounded.json 2> $OUT/bench_unbounded_kernel_table.log
  cat $OUT/bench_unbounded.json | tee -a $OUT/summary.txt
fi
echo "== rocprofv3 kernel stats (graph replay)" | tee -a $OUT/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kstats_$TAG -o k -- python $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --profile-steps 1 > $OUT/rocprof_bench.log 2>&1
cd $R
python - <<PY | tee -a $OUT/summary.txt
import glob, os, sqlite3
out = "$OUT"
dbs = glob.glob("/tmp/kstats_$TAG/**/*results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    try:
        rows = db.execute("select name, grid_x*grid_y*grid_z, workgroup_x, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 "
                          "from kernels group by name, grid_x, grid_y, workgroup_x order by 6 desc").fetchall()
    except Exception as e:
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        print("schema differs:", e, tabs[:40]); rows = []
    tot = sum(r[5] for r in rows) or 1.0
    with open(os.path.join(out, "rocprofv3_kernel_stats.csv"), "w") as f:
        f.write("kernel,grid_threads,workgroup,calls,avg_us,total_us,percent\n")
        for r in rows:
            f.write(f"\"{r[0][:90]}\",{r[1]},{r[2]},{r[3]},{r[4]:.2f},{r[5]:.1f},{100*r[5]/tot:.2f}\n")
    print(open(os.path.join(out, "rocprofv3_kernel_stats.csv")).read()[:4000])
else:
    print("no rocprofv3 database found")
PY
echo "== done" | tee -a $OUT/summary.txt
