#!/bin/bash
# Round 6, GPU session 32: per-kernel totals of an 800 x 800 eval render (device-side chunk loop), rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s32
mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ktl
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ktl -o k -- python $R/scripts/bench_render.py --frames 5 > $out/rocprof.log 2>&1
cd $R
OUT=$out python - <<'PY'
import glob, os, sqlite3, collections
out = os.environ["OUT"]
dbs = glob.glob("/tmp/ktl/**/*results.db", recursive=True)
db = sqlite3.connect(dbs[0])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z from kernels order by start").fetchall()
fr = rows
agg = collections.OrderedDict()
for r in fr:
    k = r[0][:64]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (r[2]-r[1])/1e3
span = (fr[-1][2]-fr[0][1])/1e3
with open(os.path.join(out, "eval_frame_kernels.txt"), "w") as f:
    f.write(f"whole run (warm-up frame + 5 timed frames): {len(fr)} launches, span {span:.0f} us, kernel sum {sum(a[1] for a in agg.values()):.0f} us\n")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{a[1]:10.1f} us {a[0]:5d} x {a[1]/a[0]:8.1f}  {k}\n")
print(open(os.path.join(out, "eval_frame_kernels.txt")).read())
PY
grep '^{' $out/rocprof.log | cut -c1-400
