#!/usr/bin/env python3
"""Authoring container only (/root/reference is not on the GPU box): time one nerfacto training step on 4096 rays
(BASELINE configs[1]) with (a) the REFERENCE's own torch-path modules and (b) the CPU oracle port that bench.py's
`cpu_baseline` leg runs, same parameters, same rays, same thread count — the port must reproduce the reference's step time
(within +-20 %).

    python scripts/cpu_reference_vs_port.py [threads]  >  profiles/r02_cpu_reference_vs_port.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402  (sets up the import stubs and sys.path for /root/reference)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import nerfacto_oracle as orc  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
torch.set_num_threads(threads)
N, STEPS = bench.RAYS_PER_GPU, 3
cfg = orc.NerfactoCfg()
o, d, cam, tgt = (torch.from_numpy(a) for a in bench.synthetic_rays(1000, "bounded"))
cam = cam[:, 0]
rs = np.random.RandomState(1)
jits = [[torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32)) for _ in range(3)] for _ in range(STEPS + 1)]


def time_reference():
    params = orc.init_params(cfg, seed=0)
    fld, props = MG.build_reference(cfg, params)
    sampler = MG.ProposalNetworkSampler(num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96),
                                        num_proposal_network_iterations=2, single_jitter=True)
    collider = MG.NearFarCollider(0.05, 1000.0)
    rgb_r = MG.RGBRenderer("last_sample")
    for m in (fld, props, sampler, collider, rgb_r):
        m.train(True)
    plist = list(fld.parameters()) + list(props.parameters())
    opt = torch.optim.Adam(plist, lr=1e-2, eps=1e-15)
    times = []
    for it in range(STEPS + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        rb = MG.RayBundle(origins=o, directions=d, pixel_area=torch.full((N, 1), 1e-6), camera_indices=cam[:, None])
        rb = collider(rb)
        with MG.replay_rand(jits[it]):
            rsamp, wl, rsl = sampler(rb, density_fns=[p.density_fn for p in props])
        fo = fld(rsamp)
        w = rsamp.get_weights(fo[MG.FieldHeadNames.DENSITY])
        wl.append(w)
        rsl.append(rsamp)
        rgb = rgb_r(rgb=fo[MG.FieldHeadNames.RGB], weights=w)
        loss = torch.nn.functional.mse_loss(tgt, rgb) + MG.interlevel_loss(wl, rsl) + 0.002 * MG.distortion_loss(wl, rsl)
        loss.backward()
        opt.step()
        if it > 0:
            times.append(time.perf_counter() - t0)
    return float(np.median(times)), float(loss.detach())


def time_port():
    params = orc.init_params(cfg, seed=0)
    plist = list(params.values())
    for p in plist:
        p.requires_grad_(True)
    opt = torch.optim.Adam(plist, lr=1e-2, eps=1e-15)
    times = []
    for it in range(STEPS + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = orc.nerfacto_forward(params, cfg, o, d, cam, jits[it], training=True)
        loss = sum(orc.nerfacto_losses(out, tgt, cfg).values())
        loss.backward()
        opt.step()
        if it > 0:
            times.append(time.perf_counter() - t0)
    return float(np.median(times)), float(loss.detach())


t_ref, l_ref = time_reference()
t_port, l_port = time_port()
print(f"nerfacto training step, {N} rays x (256, 96, 48) samples, full tables (2^19 / 2^17), fwd + losses + bwd + Adam,")
print(f"CPU, {threads} threads, torch {torch.__version__}; median of {STEPS} steps after 1 warm-up, same parameters / rays / jitter")
print(f"  reference modules (/root/reference, implementation='torch'): {t_ref:7.3f} s/step = {N / t_ref:8.1f} rays/s   loss after {STEPS + 1} steps {l_ref:.6f}")
print(f"  CPU oracle port (oracle/nerfacto_oracle.py)                 : {t_port:7.3f} s/step = {N / t_port:8.1f} rays/s   loss after {STEPS + 1} steps {l_port:.6f}")
print(f"  port / reference step time = {t_port / t_ref:.3f}")
