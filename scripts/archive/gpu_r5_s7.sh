#!/bin/bash
# Round 5, GPU session 7: where the record emission's extra write traffic goes: memory-side write requests of
# the fused backward (records leave four at a time between GEMM phases) and of the stand-alone route pass, split by request size.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r5_s7
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "WRREQ|WRITE_SIZE|EA0_WR|EA_WR" | head -40 > $out/counters.txt
cat $out/counters.txt
CMD="python $R/bench.py --steps 3 --warmup 3 --windows 1 --long-steps 0 --no-graph --no-cpu-baseline --profile-steps 1 --fixed-batch"
export NSAMD_SIDE_STREAM=0
for arm in fused two_launch; do
  if [ $arm = two_launch ]; then export NSAMD_FUSE_ROUTE=0; else unset NSAMD_FUSE_ROUTE; fi
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d /tmp/pmc_wr_$arm -o w -- $CMD > $out/pmc_wr_$arm.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_ws_$arm -o w -- $CMD > $out/pmc_ws_$arm.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r5_s7")
lines = []
for arm in ("fused", "two_launch"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (f"/tmp/pmc_wr_{arm}", f"/tmp/pmc_ws_{arm}"):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r.get("Kernel_Name", "")
                if not any(k in n for k in ("field_mlp_bwd_kernel", "scatter_route_fine", "scatter_apply_kernel", "adam_kernel")):
                    continue
                key = n.split("(")[0].replace("void ", "").replace("nsamd::", "") + " grid=" + r.get("Grid_Size", "?")
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        wr, wr64, ws = m.get("TCC_EA0_WRREQ_sum"), m.get("TCC_EA0_WRREQ_64B_sum"), m.get("WRITE_SIZE")
        if wr is None:
            lines.append(f"{arm:10s} {k:60s} counters: {m}")
            continue
        b32 = (wr - (wr64 or 0)) * 32 / 1e6
        b64 = (wr64 or 0) * 64 / 1e6
        lines.append(f"{arm:10s} {k:60s} WRREQ {wr:12.0f} of them 64B {wr64 or 0:12.0f} -> {b32:8.1f} MB as 32-B requests + {b64:8.1f} MB as 64-B requests"
                     f" = {b32 + b64:8.1f} MB; WRITE_SIZE {((ws or 0) * 1024) / 1e6:8.1f} MB")
open(os.path.join(out, "write_requests.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
tail -3 $out/pmc_wr_fused.log
