#!/bin/bash
# Round 5, GPU session 1: the merged per-ray launches (tests + same-box A/B), the apply pass through count-sized buffer
# descriptors (library variant), the PSNR stand-in on 8 seeds with the fused-route / rounding arms.
out=gpurun_out/r5_s1
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
timeout 420 python -m pytest tests/test_gpu_fused_launches.py tests/test_gpu_bench_parity.py -x -q -m gpu > $out/pytest_new.log 2>&1
el "pytest new: rc $? $(tail -1 $out/pytest_new.log)"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], j['config'].get('param_checksum',{}).get('params'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --param-checksum --profile-steps 1"
{
for i in 1 2; do
  echo "== merged launches (default)";            timeout 200 $B 2>/dev/null | line
  echo "== separate launches";                    NSAMD_FUSE_RAYS=0 NSAMD_FUSE_SELECT=0 timeout 200 $B 2>/dev/null | line
  echo "== merged, weights backward not folded";  NSAMD_FOLD_WEIGHTS_BWD=0 timeout 200 $B 2>/dev/null | line
  echo "== merged + apply through buffer loads";  NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_applybuf.so timeout 200 $B 2>/dev/null | line
  echo "== merged + round-to-nearest fixed point"; NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_round.so timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --kernel-table > $out/bench_kernel_table.json 2> $out/bench_kernel_table.log
el "kernel table done"
timeout 600 python scripts/psnr_ab.py --seeds 0,1,2,3,4,5,6,7 --twins 1 --arms fuse1,fuse0 > $out/psnr_trunc.txt 2>&1
el "psnr default lib: rc $?"
tail -12 $out/psnr_trunc.txt
NSAMD_LIB=$PWD/nerfstudio_amd/libnsamd_round.so timeout 400 python scripts/psnr_ab.py --seeds 0,1,2,3,4,5,6,7 --twins 1 --arms fuse1 > $out/psnr_round.txt 2>&1
el "psnr round lib: rc $?"
tail -8 $out/psnr_round.txt
el end
