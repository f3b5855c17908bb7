#!/bin/bash
# Round 5, GPU session 13: L1 / L2 counters of the main hash forward at one coarse and one fine resolution (what bounds a fine
# level: L1 tag lookups or the 64 B/clk L2 -> L1 fill path at 128 B per 16-B gather)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s13
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "TCP_|TCC_|TA_|TD_" | grep -i -E "name|counter" | head -150 > $out/avail.txt
wc -l $out/avail.txt
CMD="python $R/scripts/probe_hash_levels.py --res 16,2048 --reps 3"
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_h$i -o h -- $CMD > $out/pmc_$i.log 2>&1
  echo "set $i ($set): rc $?"
done
cd $R
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5_s13"
rows = []
for f in glob.glob("/tmp/pmc_h*/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
seq = collections.defaultdict(list)
for r in rows:
    if "hash_encode_fwd" in r.get("Kernel_Name", ""):
        seq[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
with open(out + "/hash_counters.txt", "w") as f:
    for c, v in sorted(seq.items()):
        v.sort()
        line = f"{c:36s} " + " ".join(f"{x:.4g}" for _, x in v)
        print(line); f.write(line + "\n")
PY
