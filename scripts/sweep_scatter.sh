#!/bin/bash
# GPU box: per-level split of the scatter into pass 1 / pass 2 (rocprofv3 kernel trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for l in -1 0 2 4; do
  NSAMD_SCATTER_ONLY_LEVEL=$l timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l$l -o s -- python $R/scripts/probe_scatter3.py --dense > /tmp/rp.log 2>&1
  echo "== level $l"
  python - <<PY
import sqlite3
db=sqlite3.connect("/tmp/prof_l$l/s_results.db")
for r in db.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0, max(end-start)/1000.0 from kernels where name like '%hash_bwd%' group by name, grid_x, grid_y, workgroup_x order by grid_x, name"): print(r[0][7:36], r[1:5], round(r[5],1), "us avg", round(r[6],1), "max")
PY
done
