#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02f}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== tests (scatter-related)" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -k "scatter or reproducible or hashgrid or pipeline_golden" > $OUT/pytest.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "passed|failed|^E  " $OUT/pytest.log | head -12 | tee -a $OUT/summary.txt
echo "== scatter main in isolation: default, 2048 tiles, 514" | tee -a $OUT/summary.txt
timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
NSAMD_SCATTER_TILES=2048 timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
NSAMD_SCATTER_SHAPE=514 timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
timeout 200 python scripts/probe_scatter_main.py 2>&1 | grep "scatter main" | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel stats of the isolated scatter" | tee -a $OUT/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kst -o k -- python $R/scripts/probe_scatter_main.py > $OUT/rocprof_probe.log 2>&1
cd $R
python - <<PY | tee -a $OUT/summary.txt
import glob, sqlite3
dbs = glob.glob("/tmp/kst/**/*results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select name, count(*), avg(end-start)/1000.0 from kernels where name like '%scatter%' group by name order by 3 desc").fetchall()
    for r in rows: print(r[0][:80], r[1], round(r[2], 2))
PY
echo "== done" | tee -a $OUT/summary.txt
