#!/bin/bash
# Round 5, GPU session 8: the data-parallel schedule over a one-rank RCCL communicator with eager segments (the N > 1 default) and
# with captured segments (--dp-graph), against the N = 1 graph line on the same box; parameter checksums must agree between the
# two data-parallel arms.
out=gpurun_out/r5_s8
mkdir -p $out
export TMPDIR=/tmp
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(d['ms_per_step'], c['window_ms']['min'], c['window_ms']['max'], c['launch'][:40], c['param_checksum']['params'][:12], c.get('dp_mode'), c.get('rccl_ranks'))"; }
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --long-steps 0 --profile-steps 1 --param-checksum"
{
for i in 1 2; do
  echo "== N = 1 graph";                         timeout 120 $B 2>/dev/null | line
  echo "== one-rank RCCL, eager segments";        timeout 120 $B --force-dp 2>/dev/null | line
  echo "== one-rank RCCL, captured segments";     timeout 120 $B --force-dp --dp-graph 2>/dev/null | line
  echo "== one-rank RCCL sharded, captured";      timeout 120 $B --force-dp --dp-mode sharded --dp-graph 2>/dev/null | line
done
} > $out/dp_arms.txt 2>&1
cat $out/dp_arms.txt
