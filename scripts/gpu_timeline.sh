#!/bin/bash
# GPU box: kernel timeline of two replayed training steps (rocprofv3 --kernel-trace): name, start, end relative to the step,
# so that gaps and the critical path of the captured graph can be read off. Writes gpurun_out/<tag>/timeline.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-timeline}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl
timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 12 --warmup 10 --no-cpu-baseline > $OUT/rocprof.log 2>&1
cd $R
python - <<'PY'
import glob, os, sqlite3
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", os.environ.get("TAG", "timeline"))
dbs = glob.glob("/tmp/ktl/**/*results.db", recursive=True)
db = sqlite3.connect(dbs[0])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, queue_id, stream_id from kernels order by start").fetchall()
# a step starts at select_batch; take the last 3 complete steps
starts = [i for i, r in enumerate(rows) if "select_batch" in r[0]]
lo, hi = starts[-5], starts[-1]
t0 = rows[lo][1]
with open(os.path.join(out, "timeline.csv"), "w") as f:
    f.write("kernel,start_us,end_us,dur_us,grid,wg,queue,stream\n")
    for r in rows[lo:hi]:
        f.write(f"\"{r[0][:60]}\",{(r[1]-t0)/1e3:.2f},{(r[2]-t0)/1e3:.2f},{(r[2]-r[1])/1e3:.2f},{r[3]},{r[4]},{r[5]},{r[6]}\n")
print(open(os.path.join(out, "timeline.csv")).read()[:12000])
PY
