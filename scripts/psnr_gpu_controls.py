#!/usr/bin/env python3
"""GPU-side chaos controls for the PSNR stand-in (tests/test_gpu_training.py::test_psnr_on_procedural_scene...): the SAME
GPU training repeated from initial hash tables perturbed by 1e-6 relative noise (k = 0: unperturbed). Shows how far two
correct runs of the MI355X path itself end apart, per seed: mean PSNR over the 120 training / 20 held-out views and the
delta to the CPU-oracle fixture. GPU box only:  python scripts/psnr_gpu_controls.py [twins per seed, default 4]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import psnr_scene as S  # noqa: E402
import test_gpu_training as T  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.cameras.rays import RayBundle  # noqa: E402
from oracle import nerfacto_oracle as orc  # noqa: E402

twins = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("seed twin | training views: GPU, oracle fixture, delta | held-out: GPU, fixture, delta")
only = os.environ.get("PSNR_SEEDS")
for seed in ([int(x) for x in only.split(",")] if only else S.SEEDS):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"psnr_scene_s{seed}.npz")))
    main_log2, prop_log2, init_seed = (int(v) for v in g["cfg"])
    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, main_log2),
                          prop_grids=(orc.HashGridCfg(5, 16, 128, prop_log2), orc.HashGridCfg(5, 16, 256, prop_log2)),
                          num_images=S.N_TRAIN, appearance_embed_dim=0)
    for k in range(twins + 1):
        params = orc.init_params(cfg, seed=init_seed)
        if k > 0:
            gen = torch.Generator().manual_seed(1000 * seed + k)
            for name, p in params.items():
                if "hash_table" in name:
                    p.mul_(1.0 + 1e-6 * torch.randn(p.shape, generator=gen))
        F._SCATTER_WS.clear()
        model, arena, losses = T._train(F, cfg, params, S.RAYS_PER_STEP, S.STEPS, seed=0, batches=S.batches(seed=9 + seed))
        model.eval()
        psnr = []
        for cam_id in S.ALL_CAMERAS:
            o, d, gt = S.full_view(cam_id)
            rb = RayBundle(origins=torch.from_numpy(o).cuda(), directions=torch.from_numpy(d).cuda(),
                           pixel_area=torch.full((len(o), 1), 1e-6, device="cuda"),
                           camera_indices=torch.zeros((len(o), 1), dtype=torch.int64, device="cuda"))
            with torch.no_grad():
                out = model.get_outputs_for_camera_ray_bundle(rb._map(lambda t: t.view(S.H, S.W, -1)))
            psnr.append(S.psnr(out["rgb"].reshape(-1, 3).cpu().numpy(), gt))
        psnr = np.array(psnr)
        tr, ho = slice(0, S.N_TRAIN), slice(S.N_TRAIN, None)
        a, b = psnr[tr].mean(), g["psnr_views"][tr].mean()
        c, e = psnr[ho].mean(), g["psnr_views"][ho].mean()
        print(f"  {seed}   {k}   | {a:7.3f} {b:7.3f} {a - b:+6.3f} | {c:7.3f} {e:7.3f} {c - e:+6.3f}", flush=True)
        del model, arena
