#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc1 -o p -- python $R/scripts/probe_scatter3.py --dense > /tmp/rp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN --output-format csv -d /tmp/pmc2 -o p -- python $R/scripts/probe_scatter3.py --dense >> /tmp/rp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -- python $R/scripts/probe_scatter3.py --dense >> /tmp/rp.log 2>&1
python - <<'PY'
import csv, glob, collections, sqlite3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "hash_bwd" not in n: continue
            key = n.split("(")[0].replace("void nsamd::", "") + " grid=" + r["Grid_Size"]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k); print("   ", {c: round(sum(v) / len(v) / 1e6, 3) for c, v in agg[k].items()}, "(millions)")
db=sqlite3.connect(glob.glob("/tmp/ks/**/*results.db", recursive=True)[0])
for r in db.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0 from kernels where name like '%hash_bwd%' group by name, grid_x, grid_y, workgroup_x order by grid_x, name"): print(r[0][7:44], r[1:5], round(r[5],1), "us avg")
PY
