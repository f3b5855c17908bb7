#!/bin/bash
# Round 6, GPU session 30: where the EAGER iteration's host time goes (the N > 1 path launches eagerly): cProfile of bench.py
# --no-graph and of the one-rank data-parallel rehearsal (--force-dp), top functions by own and cumulative time.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6_s30
mkdir -p $out
export TMPDIR=/tmp
cd $R
{
for arm in "eager:--no-graph" "force_dp:--force-dp"; do
  name=${arm%%:*}; flags=${arm#*:}
  echo "== $name"
  timeout 600 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--steps', '200', '--warmup', '20', '--windows', '1', '--long-steps', '0', '--no-cpu-baseline', '--no-secondary'] + '$flags'.split()
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s)
ps.sort_stats('tottime').print_stats(28)
ps.sort_stats('cumulative').print_stats(45)
print(s.getvalue())
" 2>&1 | grep -v amdgpu.ids | cut -c1-180
done
} > $out/summary.txt 2>&1
head -150 $out/summary.txt
