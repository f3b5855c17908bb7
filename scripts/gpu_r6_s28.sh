#!/bin/bash
# Round 6, GPU session 28: per-kernel durations of update iterations with the merged proposal chain, in line (every kernel alone),
# in the dense phase (steps 60+) and past it (steps 150+); rocprofv3 --kernel-trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s28
mkdir -p $out
export TMPDIR=/tmp
for arm in "in_line_60:NSAMD_SIDE_STREAM=0:60" "in_line_150:NSAMD_SIDE_STREAM=0:150" "default_60::60"; do
  name=${arm%%:*}; rest=${arm#*:}; envs=${rest%%:*}; warm=${rest#*:}
  cd /tmp; rm -rf /tmp/ktl
  env $envs timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 12 --warmup $warm --windows 1 --long-steps 0 --no-cpu-baseline --no-secondary > $out/rocprof_$name.log 2>&1
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
