#!/bin/bash
# Round 6, GPU session 31: kernel timeline of iterations with the camera optimiser ON (SO3xR3, the reference's nerfacto default)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s31
mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ktl
timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 12 --warmup 150 --windows 1 --long-steps 0 --no-cpu-baseline --no-secondary --camera-optimizer SO3xR3 > $out/rocprof.log 2>&1
cd $R
OUT=$out python - <<'PY'
import glob, os, sqlite3
out = os.environ["OUT"]
dbs = glob.glob("/tmp/ktl/**/*results.db", recursive=True)
db = sqlite3.connect(dbs[0])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, queue_id, stream_id from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "step_prologue" in r[0]]
lo, hi = starts[-7], starts[-1]
t0 = rows[lo][1]
with open(os.path.join(out, "timeline_camera.csv"), "w") as f:
    f.write("kernel,start_us,end_us,dur_us,grid,wg,queue,stream\n")
    for r in rows[lo:hi]:
        f.write(f"\"{r[0][:70]}\",{(r[1]-t0)/1e3:.2f},{(r[2]-t0)/1e3:.2f},{(r[2]-r[1])/1e3:.2f},{r[3]},{r[4]},{r[5]},{r[6]}\n")
PY
grep '^{' $out/rocprof.log | cut -c1-300
