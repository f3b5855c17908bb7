#!/bin/bash
# Round 6, GPU session 25: per-kernel durations of UPDATE iterations past the sparse phase (steps 150+; rocprofv3 --kernel-trace),
# proposal chains in line (NSAMD_SIDE_STREAM=0: every kernel alone) and on the side stream (default).
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s25
mkdir -p $out
export TMPDIR=/tmp
for arm in "in_line:NSAMD_SIDE_STREAM=0" "default:"; do
  name=${arm%%:*}; envs=${arm#*:}
  cd /tmp; rm -rf /tmp/ktl
  env $envs timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 12 --warmup 150 --windows 1 --long-steps 0 --no-cpu-baseline --no-secondary > $out/rocprof_$name.log 2>&1
  cd $R
  NAME=$name OUT=$out python - <<'PY'
import glob, os, sqlite3
out, name = os.environ["OUT"], os.environ["NAME"]
dbs = glob.glob("/tmp/ktl/**/*results.db", recursive=True)
db = sqlite3.connect(dbs[0])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, queue_id, stream_id from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "step_prologue" in r[0]]
lo, hi = starts[-7], starts[-1]
t0 = rows[lo][1]
with open(os.path.join(out, f"timeline_{name}.csv"), "w") as f:
    f.write("kernel,start_us,end_us,dur_us,grid,wg,queue,stream\n")
    for r in rows[lo:hi]:
        f.write(f"\"{r[0][:70]}\",{(r[1]-t0)/1e3:.2f},{(r[2]-t0)/1e3:.2f},{(r[2]-r[1])/1e3:.2f},{r[3]},{r[4]},{r[5]},{r[6]}\n")
PY
done
ls -la $out
