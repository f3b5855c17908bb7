#!/bin/bash
# GPU box: rocprofv3 evidence for the bench command (run via gpurun). Writes under gpurun_out/<tag = $1, default final>/:
#   kernel_stats.csv     per-kernel time (rocprofv3 --kernel-trace --stats of `bench.py --steps 20`, graph replay)
#   pmc_summary.csv      SQ counters (MFMA busy, LDS waits, ...), FETCH_SIZE, WRITE_SIZE per kernel — three separate
#                        passes with --kernel-trace only (no sys/hip/hsa tracing), as required on this pool
#   pmc_traffic.json     HBM-side bytes per launch for the bench's kernel keys (MI355X_MICROARCH.md "HBM": FETCH_SIZE and
#                        WRITE_SIZE are reported in KiB; FETCH_SIZE counts 128-B requests at 64 B for wide coalesced
#                        streaming reads -> doubled for the kernels that stream 16 B per lane, raw value kept too)
R=${GRAFT_REPO_ROOT:-/root/repo}
export OUT=$R/gpurun_out/${1:-final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout ${PMC_STATS_TIMEOUT:-600} rocprofv3 --kernel-trace --stats -d /tmp/kstats -o k -- python $R/bench.py --steps 20 --warmup 5 --windows 1 --long-steps 0 --no-cpu-baseline --profile-steps 1 > $OUT/rocprof_bench.log 2>&1
CMD="python $R/bench.py --steps 3 --warmup 3 --windows 1 --long-steps 0 --no-graph --no-cpu-baseline --profile-steps 1 --fixed-batch"
export NSAMD_SIDE_STREAM=0
timeout ${PMC_PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_sq -o sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout ${PMC_PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o f -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout ${PMC_PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o w -- $CMD > $OUT/pmc_write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json, os, sqlite3
out = os.environ["OUT"]
# ---- kernel stats from the graph-replay run
dbs = glob.glob("/tmp/kstats/**/*results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select name, grid_x*grid_y*grid_z, workgroup_x, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 "
                      "from kernels group by name, grid_x, grid_y, workgroup_x order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows)
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        f.write("kernel,grid_threads,workgroup,calls,avg_us,total_us,percent\n")
        for r in rows:
            f.write(f"\"{r[0][:90]}\",{r[1]},{r[2]},{r[3]},{r[4]:.2f},{r[5]:.1f},{100*r[5]/tot:.2f}\n")
    print(open(os.path.join(out, "kernel_stats.csv")).read()[:3500])
# ---- PMC
def load(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pmc_sq", "/tmp/pmc_fetch", "/tmp/pmc_write"):
    for r in load(d):
        n = r.get("Kernel_Name", "")
        if "nsamd" not in n:
            continue
        key = n.split("(")[0].replace("void ", "").replace("nsamd::", "") + " grid=" + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
         "SQ_INSTS_VALU", "SQ_WAIT_INST_LDS", "FETCH_SIZE", "WRITE_SIZE"]
mean = lambda v: sum(v) / len(v) if v else None
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(agg):
        n = max(len(v) for v in agg[k].values())
        f.write(k + f",{n}," + ",".join(f"{mean(agg[k][c]):.1f}" if agg[k][c] else "" for c in names) + "\n")
print(open(os.path.join(out, "pmc_summary.csv")).read())
# ---- HBM-side bytes per launch of the bench's kernel keys (config 2: N = 4096, S = (256, 96, 48))
def kib(k, c):
    return (mean(agg[k][c]) or 0.0) * 1024.0
groups = {  # bench key -> (kernel substring, grid, streaming) parts; streaming = 16 B/lane coalesced reads dominate
    "nsamd_hashgrid_encode_bwd_set[L=16,M=196608]": [("scatter_route_fine_kernel<1024, 1, 4>", "786432", False),
                                                      ("scatter_apply_kernel<false>", "1048576", True),
                                                      ("scatter_finish_kernel", "8192", False)],
    "nsamd_field_mlp_bwd": [("field_mlp_bwd_kernel<false", None, False), ("field_dw_reduce_kernel", None, False)],
    # the fused launch group of the training step, one bench key per phase (nsamd_field_mlp_bwd_scatter_phase)
    "nsamd_field_mlp_bwd_scatter_phase[gradients+records]": [("field_mlp_bwd_kernel<true", None, False)],
    "nsamd_field_mlp_bwd_scatter_phase[dw_reduce]": [("field_dw_reduce_kernel", None, False)],
    # (round 5: in the training step the weight-gradient reduce RIDES the apply pass — 5 more rows of 64 workgroups, grid
    #  1376256: the eager PMC passes see that launch, listed under its own key; the bench's "[apply]" key is the phase launch
    #  of its per-kernel table, the apply pass alone)
    "nsamd_field_mlp_bwd_scatter_phase[apply]": [("scatter_apply_kernel<false>", "1048576", True),
                                                 ("scatter_finish_kernel", "8192", False)],
    "nsamd_field_mlp_bwd_scatter[apply + riding reduce]": [("scatter_apply_kernel<true>", "1376256", True),
                                                           ("scatter_finish_kernel", "8192", False)],
    "nsamd_field_mlp_fwd": [("field_mlp_fwd_kernel", None, False)],
    "nsamd_adam_step[n=16826880]": [("adam_kernel", "524288", True)],
    "nsamd_hashgrid_encode_fwd[L=16,M=196608]": [("hash_encode_fwd", "3145728", False)],
}
traffic = {}
for key, parts in groups.items():
    fetch_raw = fetch_cal = write = 0.0
    found = []
    for sub, grid, streaming in parts:
        for k in agg:
            if sub in k and (grid is None or k.endswith("grid=" + grid)):
                fr = kib(k, "FETCH_SIZE")
                fetch_raw += fr
                fetch_cal += fr * (2.0 if streaming else 1.0)
                write += kib(k, "WRITE_SIZE")
                found.append(k)
    traffic[key] = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_calibrated": fetch_cal, "write_bytes": write,
                    "hbm_bytes": fetch_cal + write, "kernels": found}
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nerfstudio_amd.utils import roofline  # bench.py reports a traffic figure only when it was measured on these very sources
traffic["_kernel_sources_sha256_16"] = roofline.kernel_sources_hash()
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
PY
