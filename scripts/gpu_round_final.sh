#!/bin/bash
# GPU-box session that produces the round's committed evidence: full GPU test log, smoke, the default bench line (with
# cpu_baseline), per-kernel table, the unbounded-workload line, rocprofv3 kernel stats, PMC passes + traffic file.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_final}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
: > $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "mean PSNR|^   [0-9] \||GPU - oracle|passed|failed|^E  " $OUT/pytest_gpu.log | head -30 | tee -a $OUT/summary.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee -a $OUT/summary.txt
echo "== PMC passes + rocprofv3 stats (scripts/collect_pmc.sh)" | tee -a $OUT/summary.txt
bash scripts/collect_pmc.sh $TAG > $OUT/collect_pmc.log 2>&1
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json 2>/dev/null
head -n 45 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
echo "== bench default (traffic now stamped for these sources)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --kernel-table > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
cat $OUT/bench_default.json | tee -a $OUT/summary.txt
head -n 24 $OUT/bench_default_kernel_table.log | tee -a $OUT/summary.txt
echo "== bench 300 steps" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --kernel-table --profile-steps 10 > $OUT/bench_300steps.json 2> $OUT/bench_300steps_kernel_table.log
cut -c1-260 $OUT/bench_300steps.json | tee -a $OUT/summary.txt
head -n 24 $OUT/bench_300steps_kernel_table.log | tee -a $OUT/summary.txt
echo "== bench --start-step 5000 (steady state of the proposal update schedule; not the headline)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 120 --warmup 12 --start-step 5000 --no-cpu-baseline > $OUT/bench_start5000.json 2>/dev/null
cut -c1-260 $OUT/bench_start5000.json | tee -a $OUT/summary.txt
echo "== per-kind iteration times (scripts/probe_iteration_times.py): Adam deferred (default) / in order" | tee -a $OUT/summary.txt
for d in 1 0; do NSAMD_DEFER_MAIN_ADAM=$d timeout 300 python scripts/probe_iteration_times.py 2>/dev/null | tee -a $OUT/summary.txt; done
echo "== bench unbounded" | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload unbounded --no-cpu-baseline --kernel-table > $OUT/bench_unbounded.json 2> $OUT/bench_unbounded_kernel_table.log
cat $OUT/bench_unbounded.json | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
