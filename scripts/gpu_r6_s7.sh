#!/bin/bash
# Round 6, GPU session 7: kernel timeline (rocprofv3 --kernel-trace) of replayed iterations of the long run's regime (start step 40:
# every second iteration updates the proposal networks): what is on the critical path of an update iteration?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_s7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl
timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 12 --warmup 10 --start-step 40 --long-steps 0 --windows 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1
cd $R
TAG=r6_s7 python - <<'PY'
import glob, os, sqlite3
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", os.environ.get("TAG", "timeline"))
dbs = glob.glob("/tmp/ktl/**/*results.db", recursive=True)
db = sqlite3.connect(dbs[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'").fetchall()]
print([t for t in tabs if "kernel" in t.lower()][:10])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, queue_id, stream_id from kernels order by start").fetchall()
lo, hi = 0, len(rows)
t0 = rows[lo][1]
with open(os.path.join(out, "timeline.csv"), "w") as f:
    f.write("kernel,start_us,end_us,dur_us,grid,wg,queue,stream\n")
    for r in rows[lo:hi]:
        f.write(f"\"{r[0][:60]}\",{(r[1]-t0)/1e3:.2f},{(r[2]-t0)/1e3:.2f},{(r[2]-r[1])/1e3:.2f},{r[3]},{r[4]},{r[5]},{r[6]}\n")
PY
tail -3 $OUT/rocprof.log
