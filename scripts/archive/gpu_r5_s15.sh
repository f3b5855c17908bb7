#!/bin/bash
# Round 5, GPU session 15: the hash forward with lane pairs sharing a line (NSAMD_HASH_FWD_MODE=7) — bit equality with the
# oracle, per-level cost by resolution, driver-window A/B against mode 3
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s15
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/progress.txt; }
for mode in 7 3; do
  NSAMD_HASH_FWD_MODE=$mode timeout 300 python -m pytest tests/test_gpu_kernels.py -k "hashgrid or nerfacto_field_golden" -q -m gpu > $out/pytest_mode$mode.log 2>&1
  el "mode $mode pytest: rc $? $(tail -1 $out/pytest_mode$mode.log)"
done
NSAMD_HASH_FWD_LEVELS=1 NSAMD_HASH_FWD_MODE=7 timeout 300 python -m pytest tests/test_gpu_kernels.py -k "hashgrid or nerfacto_field_golden or proposal_density_golden" -q -m gpu > $out/pytest_mode7_small_tables.log 2>&1
el "mode 7, every table through it: rc $? $(tail -1 $out/pytest_mode7_small_tables.log)"
grep -E "^E  |^FAILED" $out/pytest_mode*.log | head -20
for mode in 3 7; do
  echo "== NSAMD_HASH_FWD_MODE=$mode" >> $out/hash_levels.txt
  NSAMD_HASH_FWD_MODE=$mode timeout 120 python scripts/probe_hash_levels.py --res 16,58,111,212,2048 >> $out/hash_levels.txt 2>/dev/null
done
cat $out/hash_levels.txt
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['config']['window_ms']['min'], j['config']['window_ms']['max'], 'long', (j.get('long_run') or {}).get('ms_per_step'))"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 100 --profile-steps 1 --param-checksum"
{
for i in 1 2 3; do
  for mode in 3 7; do
    echo "== NSAMD_HASH_FWD_MODE=$mode"; NSAMD_HASH_FWD_MODE=$mode timeout 150 $B 2>/dev/null | tee $out/bench_mode${mode}_$i.json | line
  done
done
} > $out/ab_bench.txt 2>&1
el "bench A/B done"
cat $out/ab_bench.txt
python - <<PY
import json
for m in (3, 7):
    j = json.loads([l for l in open("$out/bench_mode%d_1.json" % m) if l.startswith("{")][0])
    print(m, j["config"].get("param_checksum"))
PY
el end
