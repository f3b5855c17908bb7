#!/usr/bin/env python3
"""GPU box: how sparse is the proposal networks' upstream gradient in the BENCHMARK's own training run? Per proposal level
and update iteration: the gating flag, the fraction of rays whose interlevel-loss gradient dw is non-zero, the fraction of
samples / of 256-sample chunks with a non-zero density gradient (what nsamd_weights_bwd_gate / density_mlp_bwd's chunk skip
/ the route kernels' workgroup skip can exploit). Eager launches of bench.Trainer, same rays / schedule as bench.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402

dev = torch.device("cuda", 0)
F.DIRECT_GRAD = True
model = bench.build_model(dev, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
tr = bench.Trainer(model, arena, rb, batch, world=1, use_graph=False, use_runner=True, pool=pool)
r = tr.runner
steps = int(os.environ.get("PROBE_STEPS", "400"))
report_at = {0, 1, 2, 5, 9, 12, 20, 30, 50, 75, 100, 150, 200, 300, 399}
print("step | per level: flag, rays with dw != 0, samples with ddens != 0, 256-chunks with any ddens != 0, max|dw|")
for k in range(steps):
    upd = model.proposal_sampler.updated_this_step()
    tr.train_iteration()
    if upd and (k in report_at or (k + 1) in report_at):
        torch.cuda.synchronize()
        row = [f"{k:4d}"]
        for lvl in range(r.n_prop):
            S = r.counts[lvl]
            dw = r.dw_prop[lvl]
            dd = r.p_ddens[lvl]
            flag = int(r.prop_gates[4 * lvl])
            rays = float((dw != 0).any(dim=1).float().mean())
            smp = float((dd != 0).float().mean()) if flag else 0.0
            ch = dd.reshape(-1)
            pad = (-ch.numel()) % 256
            chunks = float((torch.nn.functional.pad(ch, (0, pad)).reshape(-1, 256) != 0).any(dim=1).float().mean()) if flag else 0.0
            row.append(f"L{lvl}(S={S}): flag {flag} rays {rays:6.3f} samples {smp:6.3f} chunks {chunks:6.3f} max|dw| {float(dw.abs().max()):.2e}")
        print(" | ".join(row))
tr.finish()
print("final loss", float(tr.last_loss()))
