#!/bin/bash
# GPU-box session: staged pass-1 variants of the scatter: correctness (scatter + reproducibility tests) and timing.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02e}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
for v in 822 814 824; do
  echo "== shape $v: tests" | tee -a $OUT/summary.txt
  NSAMD_SCATTER_SHAPE=$v timeout 900 python -m pytest tests -m gpu -q -k "scatter or reproducible or hashgrid or pipeline_golden or camera" > $OUT/pytest_$v.log 2>&1
  echo "rc=$?" | tee -a $OUT/summary.txt
  grep -E "passed|failed|^E  " $OUT/pytest_$v.log | head -12 | tee -a $OUT/summary.txt
done
echo "== scatter main in isolation" | tee -a $OUT/summary.txt
for v in 114 822 814 824 114 822; do
  NSAMD_SCATTER_SHAPE=$v timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
done
echo "== bench with 822 / 814" | tee -a $OUT/summary.txt
for v in 822 814; do
NSAMD_SCATTER_SHAPE=$v timeout 600 python bench.py --kernel-table --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_${v}_kernel_table.log
cut -c1-200 $OUT/bench_$v.json | tee -a $OUT/summary.txt
head -n 4 $OUT/bench_${v}_kernel_table.log | tee -a $OUT/summary.txt
done
echo "== rocprofv3 kernel stats of the isolated scatter (shape 822)" | tee -a $OUT/summary.txt
cd /tmp
NSAMD_SCATTER_SHAPE=822 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kst -o k -- python $R/scripts/probe_scatter_main.py > $OUT/rocprof_probe.log 2>&1
cd $R
python - <<PY | tee -a $OUT/summary.txt
import glob, sqlite3
dbs = glob.glob("/tmp/kst/**/*results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        rows = db.execute("select name, count(*), avg(end-start)/1000.0 from kernels where name like '%scatter%' group by name order by 3 desc").fetchall()
        for r in rows: print(r[0][:80], r[1], round(r[2], 2))
    except Exception as e:
        print("query failed", e, tabs[:30])
PY
echo "== done" | tee -a $OUT/summary.txt
