#!/bin/bash
# Round 6, GPU session 46: kernel timeline of REPLAYED iterations of the current tree (rocprofv3 --kernel-trace): one non-update and
# one update iteration of the timed window, gaps and branch overlap read off
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${TAG:-r6_s46}
mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ktl
NSAMD_ISSUE_MAIN_FIRST=${MAIN_FIRST:-0} timeout 600 rocprofv3 --kernel-trace -d /tmp/ktl -o k -- python $R/bench.py --steps 12 --warmup 60 --windows 1 --long-steps 0 --no-cpu-baseline --no-secondary > $out/rocprof.log 2>&1
cd $R
OUT=$out python - <<'PY'
import glob, os, sqlite3
out = os.environ["OUT"]
dbs = glob.glob("/tmp/ktl/**/*results.db", recursive=True)
db = sqlite3.connect(dbs[0])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, queue_id, stream_id from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "step_prologue" in r[0]]
its = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    qs = sorted(set(k[5] for k in ks))
    upd = any("density_mlp_bwd" in k[0] for k in ks)
    its.append((a, b, (rows[b][1] - rows[a][1]) / 1e3, len(ks), qs, upd))
with open(os.path.join(out, "iterations.txt"), "w") as f:
    for i, (a, b, d, n, qs, upd) in enumerate(its):
        f.write(f"{i:4d} {d:10.1f} us {n:3d} launches queues {qs} {'update' if upd else ''}\n")
# replayed iterations: more than one queue; take the last two of each kind inside the window (steps 60..71 = iterations 60+)
rep = [t for t in its if len(t[4]) > 1 and t[2] < 5000]
for kind, name in ((False, "non_update"), (True, "update")):
    sel = [t for t in rep if t[5] == kind][-2:]
    for n, (a, b, d, cnt, qs, upd) in enumerate(sel):
        t0 = rows[a][1]
        with open(os.path.join(out, f"timeline_{name}_{n}.csv"), "w") as f:
            f.write("kernel,start_us,end_us,dur_us,grid,wg,queue\n")
            for r in rows[a:b]:
                f.write(f"\"{r[0][:70]}\",{(r[1]-t0)/1e3:.2f},{(r[2]-t0)/1e3:.2f},{(r[2]-r[1])/1e3:.2f},{r[3]},{r[4]},{r[5]}\n")
print(open(os.path.join(out, "iterations.txt")).read()[-3000:])
PY
