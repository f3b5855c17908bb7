#!/bin/bash
# GPU box: PMC passes + rocprofv3 kernel stats, then the default bench line with the traffic file stamped for these sources.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_pmc}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
bash scripts/collect_pmc.sh $TAG > $OUT/collect_pmc.log 2>&1
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json 2>/dev/null
head -n 45 $OUT/kernel_stats.csv 2>/dev/null | cut -c1-200
timeout 600 python bench.py --kernel-table > $OUT/bench_default.json 2> $OUT/bench_default_kernel_table.log
cat $OUT/bench_default.json
