#!/bin/bash
# Round 5, GPU session 3: defaults after session 2 (one-launch sampler off) against the round-4 launches once more, the seam line
# with the host's issue time and a profile of its host side, and the apply pass on 4096-entry tiles with two workgroups per CU.
out=gpurun_out/r5_s3
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], j['config'].get('param_checksum',{}).get('params'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --param-checksum --profile-steps 1"
{
for i in 1 2 3 4; do
  echo "== default (rays + select merged, reduce rides)"; timeout 200 $B 2>/dev/null | line
  echo "== round-4 launches";  NSAMD_REDUCE_RIDER=0 NSAMD_FUSE_RAYS=0 NSAMD_FUSE_SELECT=0 timeout 200 $B 2>/dev/null | line
  echo "== only the rider off"; NSAMD_REDUCE_RIDER=0 timeout 200 $B 2>/dev/null | line
  echo "== only rays off";      NSAMD_FUSE_RAYS=0 timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
timeout 300 python scripts/bench_seam.py --profile > $out/bench_seam.json 2> $out/bench_seam.err
el "seam: rc $?"
python -c "
import json; j=json.load(open('$out/bench_seam.json'))
print({k: j[k] for k in ('direct_pool_ms','seam_ms','seam_over_direct_pool','seam_over_direct_pool_per_window','host_issue_ms_per_step')})"
grep -A60 "cumulative" $out/bench_seam.err | head -75
K="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --windows 3 --kernel-table --profile-steps 2"
for arm in "" "NSAMD_SCATTER_TILES=2048" "NSAMD_SCATTER_TILES=2048 NSAMD_APPLY_THREADS_12=512" "NSAMD_SCATTER_TILES=2048 NSAMD_APPLY_THREADS_12=256"; do
  echo "== two-launch backward, $arm" >> $out/apply_tiles.txt
  env NSAMD_FUSE_ROUTE=0 $arm timeout 200 $K 2> $out/tmp_table.log | line >> $out/apply_tiles.txt
  grep -E "hashgrid_encode_bwd_set|field_mlp_bwd " $out/tmp_table.log >> $out/apply_tiles.txt
done
el "apply tiles done"
cat $out/apply_tiles.txt
el end
