#!/usr/bin/env python3
"""Time the main field's backward launch (phase 1 of nsamd_field_mlp_bwd_scatter_phase: gradients + the scatter's records) ALONE
on the buffers of a real training iteration at the benchmark's size — the bench's model and rays, one forward + losses, then the
launch repeated with HIP events around each repetition. The library comes from NSAMD_LIB, so attribution builds
(scripts/build_variant.sh x "-DNSAMD_FIELD_BWD_SKIP_CONST=n") can be timed on the same inputs: their results are wrong, their
time is what the product kernel costs without the part they leave out. GPU box only:
    [NSAMD_LIB=...] python scripts/probe_field_bwd_real.py [reps]"""
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
from nerfstudio_amd.train_step import NerfactoTrainStep  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
model = bench.build_model(dev, 0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
n = bench.RAYS_PER_GPU
o, d, cam, tgt = (torch.from_numpy(a).to(dev) for a in bench.synthetic_rays(1003))
r = NerfactoTrainStep(model, n, dev)
r.side_stream = None
r.set_batch(o, d, cam[:, 0], tgt)
rs = np.random.RandomState(4)
r.jitter.copy_(torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)))
arena.zero_grad(skip=r.written_params())
r.forward_backward(updated=False, draw_jitter=False)  # fills f_enc, f_sel, d_dens_main, d_rgb_s (library under test: timing only)
torch.cuda.synchronize()

lib, st = N.load(), N.stream()
fld = model.field
L = r.n_prop
S, mm = r.counts[L], r.m_main
enc = fld.mlp_base.encoding
params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
emb = fld.embedding_appearance.embedding.weight
fm = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(emb), emb.shape[0], float(fld.average_init_density))
grads = N.FieldMlpGrads(*(N.ptr(r._grad(p)) for p in params), N.ptr(r._grad(emb)))
sws, sws_n = F._producer_scatter_workspace(enc.spec, r.f_enc.device, mm)
args = (r._points(L), fld._transform, fld._box, enc.spec.native(), N.ptr(r.f_enc), N.ptr(r.f_sel), N.ptr(r.directions),
        N.ptr(r.camera_indices), None, S, mm, fm, N.ptr(r.d_dens_main), N.ptr(r.d_rgb_s), None, grads, N.ptr(r.field_ws),
        r.field_ws.numel(), N.ptr(r._grad(enc.hash_table)), N.ptr(sws), sws_n)


def timed(phase):
    ts = []
    for i in range(reps + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        N.check(lib.nsamd_field_mlp_bwd_scatter_phase(*args, phase, st), "phase")
        b.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b) * 1e3)
        if phase == 1:  # the records of a launch are consumed by the apply pass before the next one is emitted
            N.check(lib.nsamd_field_mlp_bwd_scatter_phase(*args, 4, st), "apply")
    ts = np.array(ts)
    return float(np.median(ts)), float(ts.min())


nz = float((r.d_dens_main != 0).float().mean())
m1, lo1 = timed(1)
m4, lo4 = timed(4)
print(f"lib {os.environ.get('NSAMD_LIB', 'default'):40s} gradients+records median {m1:7.1f} us (min {lo1:7.1f})   apply median {m4:6.1f} us   "
      f"[samples with dL/d density != 0: {100 * nz:.1f} %]")
