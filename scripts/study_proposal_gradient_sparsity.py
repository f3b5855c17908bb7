#!/usr/bin/env python3
"""CPU study (no GPU): how sparse are the gradients the proposal networks' backward kernels process? The oracle runs
training iterations at the benchmark configuration; for each proposal level the gradient of the loss with respect to the
level's WEIGHTS (input of weights_bwd) and with respect to its DENSITIES (input of density_mlp_bwd, and what decides
which samples the table scatter skips) is captured. Reported: the fraction of exactly-zero entries, and the fraction of
aligned groups of 64 consecutive samples (one wavefront of density_mlp_bwd) / 256 (one of its chunks) that are zero
throughout. STUDY_STEPS iterations (default 3) with Adam in between."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import nerfacto_oracle as orc  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 16))
grads = {}
_orig_density, _orig_weights = orc.proposal_density, orc.weights_from_density
_level = {"next": 0}


def recording_density(pos, params, lvl, cfg, aabb=None):
    dens = _orig_density(pos, params, lvl, cfg, aabb)
    if dens.requires_grad:
        dens.register_hook(lambda g, k=lvl: grads.__setitem__(("density", k), g.detach()))
    return dens


def recording_weights(t_bins, density):
    w = _orig_weights(t_bins, density)
    k = _level["next"]
    _level["next"] += 1
    if w.requires_grad and k < 2:
        w.register_hook(lambda g, k=k: grads.__setitem__(("weights", k), g.detach()))
    return w


orc.proposal_density, orc.weights_from_density = recording_density, recording_weights


def groups_zero(g, size):
    flat = g.reshape(-1)
    m = flat.numel() // size * size
    return float((flat[:m].reshape(-1, size) == 0).all(dim=1).float().mean())


def main():
    steps = int(os.environ.get("STUDY_STEPS", "3"))
    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0)
    plist = list(params.values())
    for p in plist:
        p.requires_grad_(True)
    opt = torch.optim.Adam(plist, lr=1e-2, eps=1e-15)
    n = bench.RAYS_PER_GPU
    rs = np.random.RandomState(1)
    print("# step level  samples/ray | dL/dweights: zero   | dL/ddensity: zero   64-groups zero   256-chunks zero")
    for it in range(steps):
        o, d, cam, tgt = (torch.from_numpy(a) for a in bench.synthetic_rays(1000 + it))
        jit = [torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)) for _ in range(3)]
        grads.clear()
        _level["next"] = 0
        opt.zero_grad(set_to_none=True)
        out = orc.nerfacto_forward(params, cfg, o, d, cam[:, 0], jit, training=True)
        sum(orc.nerfacto_losses(out, tgt, cfg).values()).backward()
        for lvl in range(2):
            gw, gd = grads[("weights", lvl)], grads[("density", lvl)]
            print(f"  {it:3d}   {lvl}      {gw.shape[1]:4d}       | {float((gw == 0).float().mean()):18.3f} | "
                  f"{float((gd == 0).float().mean()):17.3f} {groups_zero(gd, 64):16.3f} {groups_zero(gd, 256):17.3f}")
        opt.step()


if __name__ == "__main__":
    main()
