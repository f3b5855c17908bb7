#!/usr/bin/env python3
"""What one LEVEL of the main hash forward costs as a function of its resolution, on the benchmark's own sample points (the
48 final samples of 4096 rays after a few training iterations): nsamd_hashgrid_encode_fwd with all 16 levels of the grid set to
one resolution R, per-launch HIP-event time / 16. Also the proposal levels' fused density field the same way (5 levels).
Answers where the 85 us of the forward go — coarse levels whose lanes share cells, or fine hashed levels — before staging
coarse levels in LDS."""
import argparse
import os
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="16,22,30,42,58,80,111,153,212,293,405,1072,2048", help="resolutions to run (all 16 levels at each)")
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfstudio_amd import _native as N  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402
from nerfstudio_amd.trainer import HipTrainer  # noqa: E402

dev = torch.device("cuda")
F.DIRECT_GRAD = True
model = bench.build_model(dev, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=False, use_runner=True, pool=pool)
for _ in range(12):
    tr.train_iteration()
tr.finish()
torch.cuda.synchronize()
r = tr.runner
fld = model.field
enc = fld.mlp_base.encoding
L = len(r.counts) - 1
mm = r.n * r.counts[L]
lib, st = N.load(), N.stream()
pts = r._points(L)


def timed(fn, reps=args.reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def fwd(spec):
    N.check(lib.nsamd_hashgrid_encode_fwd(pts, mm, fld._transform, fld._box, N.ptr(enc.hash_table), spec.native(), N.ptr(r.f_enc), 1, mm,
                                          N.ptr(r.f_sel), st), "hashgrid_encode_fwd")


base = enc.spec
print(f"main grid as shipped (L=16, {base.min_res}..{base.max_res}, T=2^{base.log2_hashmap_size}), M={mm}: {timed(lambda: fwd(base)):.1f} us")
print("scalings:", [int(s) for s in base.scalings().tolist()])
for R in [int(r) for r in args.res.split(",") if r]:
    spec = F.HashGridSpec(16, R, R, base.log2_hashmap_size)
    t = timed(lambda: fwd(spec))
    print(f"all 16 levels at res {R:5d} ((R+1)^3 = {(R + 1) ** 3:>11d} entries): {t:7.1f} us per launch = {t / 16:5.2f} us per level")
