#!/bin/bash
# Round 5, GPU session 4: the VALU diet (32-bit ray division, packed feature blends) — kernel tests, same-box A/B against the
# library built with -DNSAMD_VALU_DIET=0 —, the seam line on the new defaults, rocprofv3 kernel stats of the replayed graphs
# with and without the merged per-ray launch.
out=gpurun_out/r5_s4
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
timeout 600 python -m pytest tests/test_gpu_fused_launches.py tests/test_gpu_kernels.py -x -q -m gpu > $out/pytest_kernels.log 2>&1
el "pytest fused + kernels: rc $? $(tail -1 $out/pytest_kernels.log)"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], j['config'].get('param_checksum',{}).get('params'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --param-checksum --profile-steps 1"
{
for i in 1 2 3 4; do
  echo "== default";                 timeout 200 $B 2>/dev/null | line
  echo "== library without the diet"; NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_nodiet.so timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 300 --kernel-table > $out/bench_kernel_table.json 2> $out/bench_kernel_table.log
el "kernel table + 300 steps: $(python -c "import json;j=json.load(open('$out/bench_kernel_table.json'));print(j['ms_per_step'], j['long_run']['ms_per_step'])")"
head -22 $out/bench_kernel_table.log
NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_nodiet.so timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --windows 3 --kernel-table > /dev/null 2> $out/bench_kernel_table_nodiet.log
grep -E "density_field_fwd|hashgrid_encode_fwd" $out/bench_kernel_table_nodiet.log
timeout 300 python scripts/bench_seam.py > $out/bench_seam.json 2> $out/bench_seam.err
el "seam: rc $?"
python -c "
import json; j=json.load(open('$out/bench_seam.json'))
print({k: j[k] for k in ('direct_pool_ms','seam_ms','seam_over_direct_pool','seam_over_direct_pool_per_window')})"
cd /tmp
for arm in default fuse_rays; do
  if [ $arm = fuse_rays ]; then export NSAMD_FUSE_RAYS=1; else unset NSAMD_FUSE_RAYS; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_$arm -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --windows 3 --profile-steps 1 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/prof_$arm -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/kernel_stats_$arm.csv
  rm -rf $GRAFT_REPO_ROOT/$out/prof_$arm
done
unset NSAMD_FUSE_RAYS
cd $GRAFT_REPO_ROOT
el "rocprof stats done"
for arm in default fuse_rays; do echo "== $arm"; head -16 $out/kernel_stats_$arm.csv | cut -c1-150; done
el end
