#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy report of one HIP source (cross-compiles for gfx950, no GPU needed):
    python scripts/kernel_resources.py nerfstudio_amd/csrc/scatter.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{name[:72]:72s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>3s} SGPR {g('TotalSGPRs'):>4s} scratch {g('ScratchSize [bytes/lane]'):>4s} "
          f"occ {g('Occupancy [waves/SIMD]'):>2s} LDS {g('LDS Size [bytes/block]')}")
