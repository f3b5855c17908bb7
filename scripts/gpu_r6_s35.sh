#!/bin/bash
# Round 6, GPU session 35: kernel timelines of replayed iterations, ray terms on / off (rocprofv3 --kernel-trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s35
mkdir -p $out
export TMPDIR=/tmp
for arm in "ray_terms:NSAMD_RAY_TERMS=1" "plain:NSAMD_RAY_TERMS=0"; do
  name=${arm%%:*}; envs=${arm#*:}
  cd /tmp; rm -rf /tmp/ktl
  env $envs timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 8 --warmup 40 --windows 1 --long-steps 0 --no-cpu-baseline --no-secondary > $out/rocprof_$name.log 2>&1
  cd $R
  OUT=$out NAME=$name python - <<'PY'
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
# iteration lengths
for a, b in zip(starts[-7:-1], starts[-6:]):
    ks = rows[a:b]
    print(name, "iteration", f"{(rows[b][1]-rows[a][1])/1e3:8.1f} us", len(ks), "launches", "update" if any("density_mlp_bwd" in k[0] or "proposal_levels" in k[0] or "weights_bwd" in k[0] for k in ks) else "")
PY
done
