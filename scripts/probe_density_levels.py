#!/usr/bin/env python3
"""What the proposal networks' fused forward (nsamd_density_field_fwd) costs as a function of its grid's resolution, on the
benchmark's own points (256 / 96 samples of 4096 rays after a few training iterations): all 5 levels set to one resolution R,
per-launch HIP-event time. Companion of probe_hash_levels.py: is the launch bound by lines (grows with R) or by its own
instructions / latencies (flat)?"""
import argparse
import os
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="16,32,64,128,256,1024")
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
lib, st = N.load(), N.stream()


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


for lvl in range(len(r.counts) - 1):
    net = r.props[lvl]
    m = r.n * r.counts[lvl]
    spec0 = net.encoding.spec
    W0, b0, W1, b1 = net.mlp_base[1].param_tensors()
    dm = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0], float(net.average_init_density))

    def fwd(spec):
        N.check(lib.nsamd_density_field_fwd(r._points(lvl), m, net._transform, net._box, N.ptr(net.encoding.hash_table), spec.native(), dm,
                                            None, None, N.ptr(r.p_dens[lvl]), None, st), "density_field_fwd")

    print(f"level {lvl}: M = {m}, shipped grid {spec0.num_levels} x ({spec0.min_res}..{spec0.max_res}), T = 2^{spec0.log2_hashmap_size}: "
          f"{timed(lambda: fwd(spec0)):.1f} us   scalings {[int(s) for s in spec0.scalings().tolist()]}")
    for R in [int(x) for x in args.res.split(",") if x]:
        spec = F.HashGridSpec(spec0.num_levels, R, R, spec0.log2_hashmap_size)
        print(f"   all levels at res {R:5d}: {timed(lambda: fwd(spec)):7.1f} us")
