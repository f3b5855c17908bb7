#!/bin/bash
# GPU box: 16-wave forward variant (clock probe, parity, bench line) + L1/L2 request counters of the hash forward
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/probe_field; mkdir -p $O
NSAMD_FIELD_FWD_WAVES=16 timeout 200 python scripts/probe_field_clocks.py --no-build 2>&1 | grep -v "it [0-5]:" | tee $O/clocks_w16.log | head -24
NSAMD_FIELD_FWD_WAVES=16 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "field" 2>&1 | tail -2
for w in 4 8 16; do NSAMD_FIELD_FWD_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --kernel-table 2>&1 | grep -E "ms_per_step|field_mlp_fwd" | cut -c1-190; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC)_[A-Z0-9_]+(_sum)?\b" | sort -u > $GRAFT_REPO_ROOT/$O/counters_tcp_tcc.txt
wc -l $GRAFT_REPO_ROOT/$O/counters_tcp_tcc.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 1 --fixed-batch"
export NSAMD_SIDE_STREAM=0
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l1 -o l1 -- $CMD > $GRAFT_REPO_ROOT/$O/pmc_l1.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_l1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "nsamd" in n:
            agg[n.split("(")[0].replace("void ", "") + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k[:70].ljust(70), {c: round(sum(v) / len(v)) for c, v in agg[k].items()})
PY
