#!/bin/bash
# GPU box: hardware counters for the hot kernels (run via gpurun). Three passes: SQ (MFMA/LDS/wait), TCC fetch, TCC write.
# PMC passes use --kernel-trace only (no sys/hip/hsa tracing), as required on this pool.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 3 --no-graph --no-cpu-baseline --profile-steps 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "pmc")
def load(sub):
    rows = []
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "fetch", "write"):
    for r in load(sub):
        name = r.get("Kernel_Name", "")
        if "nsamd" not in name:
            continue
        key = name.split("(")[0].replace("void ", "") + f" grid={r.get('Grid_Size','?')}"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.csv"), "w") as f:
    names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
             "SQ_INSTS_VALU", "SQ_WAIT_INST_LDS", "FETCH_SIZE", "WRITE_SIZE"]
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(agg):
        n = max(len(v) for v in agg[k].values())
        f.write(k + f",{n}," + ",".join(f"{sum(agg[k][c])/len(agg[k][c]):.1f}" if agg[k][c] else "" for c in names) + "\n")
print(open(os.path.join(out, "summary.csv")).read())
PY
find $OUT -name "*.csv" ! -name summary.csv -size +2M -delete
