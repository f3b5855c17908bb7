#!/bin/bash
# Round 5, GPU session 2: the one-launch proposal sampler, the reduce riding the apply pass, loss values from the finishing
# pass — tests, then same-box A/B of each switch, the seam line, per-kind iteration times.
out=gpurun_out/r5_s2
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
el start
timeout 500 python -m pytest tests/test_gpu_fused_launches.py -x -q -m gpu > $out/pytest_fused.log 2>&1
el "pytest fused launches: rc $? $(tail -1 $out/pytest_fused.log)"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], j['config'].get('param_checksum',{}).get('params'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --param-checksum --profile-steps 1"
{
for i in 1 2 3; do
  echo "== all merged (default)";        timeout 200 $B 2>/dev/null | line
  echo "== sampler per level";           NSAMD_FUSE_SAMPLER=0 timeout 200 $B 2>/dev/null | line
  echo "== reduce as its own launch";    NSAMD_REDUCE_RIDER=0 timeout 200 $B 2>/dev/null | line
  echo "== round-4 launches";            NSAMD_FUSE_SAMPLER=0 NSAMD_REDUCE_RIDER=0 NSAMD_FUSE_RAYS=0 NSAMD_FUSE_SELECT=0 timeout 200 $B 2>/dev/null | line
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 300 --kernel-table > $out/bench_kernel_table.json 2> $out/bench_kernel_table.log
el "kernel table + 300 steps done: $(python -c "import json;j=json.load(open('$out/bench_kernel_table.json'));print(j['ms_per_step'], j['long_run']['ms_per_step'])")"
head -24 $out/bench_kernel_table.log
timeout 300 python scripts/bench_seam.py > $out/bench_seam.json 2> $out/bench_seam.err
el "seam: rc $? $(cut -c1-400 $out/bench_seam.json)"
tail -3 $out/bench_seam.err
PROBE_STEPS=40 timeout 200 python scripts/probe_iteration_times.py > $out/iteration_times.txt 2>&1
el "iteration times: $(tail -2 $out/iteration_times.txt)"
timeout 400 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_pipeline_seam.py -x -q -m gpu > $out/pytest_parity.log 2>&1
el "pytest parity + seam: rc $? $(tail -1 $out/pytest_parity.log)"
el end
