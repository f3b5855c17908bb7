#!/usr/bin/env python3
"""GPU box: the main-table gradient scatter (nsamd_hashgrid_encode_bwd_set, L=16, T=2^19, M=196608) in isolation, on the
gradients of a real training state: 20 eager training steps of the bench workload, then the scatter call alone, timed with
HIP events (median of 50). Environment switches (NSAMD_SCATTER_SHAPE, NSAMD_SCATTER_COMBINE_RES, ...) are read once per
process by the library, so run one process per variant."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfstudio_amd import _native as N  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402

dev = torch.device("cuda", 0)
F.DIRECT_GRAD = True
model = bench.build_model(dev, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
tr = bench.Trainer(model, arena, rb, batch, world=1, use_graph=False, use_runner=True, pool=pool)
for _ in range(int(os.environ.get("PROBE_TRAIN_STEPS", "20"))):
    tr.train_iteration()
torch.cuda.synchronize()
r = tr.runner
fld = model.field
enc = fld.mlp_base.encoding
L, mm = r.n_prop, r.m_main
lib = N.load()
ws, ws_n = F._scatter_workspace(enc.spec, dev, mm, write_only=True)
grad = enc.hash_table.grad


def call():
    N.check(lib.nsamd_hashgrid_encode_bwd_set(r._points(L), mm, fld._transform, fld._box, N.ptr(enc.hash_table), enc.spec.native(),
                                              N.ptr(r.f_denc), 1, mm, N.ptr(grad), None, N.ptr(ws), ws_n, N.stream()), "scatter")


for _ in range(5):
    call()
torch.cuda.synchronize()
ref = grad.clone()
ts = []
for _ in range(50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
assert torch.equal(grad, ref), "the scatter is not bit-reproducible"
nz = float((r.f_denc != 0).float().mean())
print(f"scatter main: median {np.median(ts):7.1f} us  min {np.min(ts):7.1f} us  (nonzero gradient fraction {nz:.2f}, "
      f"events {F.scatter_events(ws)}, env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NSAMD_")) + ")")
