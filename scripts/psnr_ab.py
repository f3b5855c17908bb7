#!/usr/bin/env python3
"""PSNR stand-in, many seeds, several arms on ONE box: is the GPU path's PSNR after 300 steps biased
against the CPU oracle's, and does the fused record emission's fixed-point scale (NSAMD_FUSE_ROUTE) or the conversion's
rounding (library built with -DNSAMD_FIXED_ROUND=1, through NSAMD_LIB) move it?

Per seed and arm: the training of tests/test_gpu_training.py::test_psnr_on_procedural_scene_matches_oracle_training from the
fixture's initial state (twin 0) and from `twins` copies whose hash tables are perturbed by 1e-6 relative (the GPU path's own
chaos spread), PSNR of the 120 training / 20 held-out views, difference to the CPU-oracle fixture (mean of its base and twin
run). Then per arm: mean +- standard error over seeds of (GPU mean over twins - oracle mean over twins), sign test.
GPU box only:  [NSAMD_LIB=...] python scripts/psnr_ab.py --seeds 0,1,2,3,4,5,6,7 --twins 2 --arms fuse1,fuse0"""
import argparse
import math
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

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", default=",".join(str(s) for s in S.SEEDS))
ap.add_argument("--twins", type=int, default=0)
ap.add_argument("--arms", default="fuse1")
args = ap.parse_args()
seeds = [int(x) for x in args.seeds.split(",")]
ARMS = {"fuse1": {"NSAMD_FUSE_ROUTE": "1"}, "fuse0": {"NSAMD_FUSE_ROUTE": "0"}}
tr, ho = slice(0, S.N_TRAIN), slice(S.N_TRAIN, None)


def run(cfg, params, seed):
    F._SCATTER_WS.clear()
    model, arena, _ = T._train(F, cfg, params, S.RAYS_PER_STEP, S.STEPS, seed=0, batches=S.batches(seed=9 + seed))
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
    del model, arena
    psnr = np.array(psnr)
    return psnr[tr].mean(), psnr[ho].mean()


print(f"library: {os.environ.get('NSAMD_LIB', 'in-tree libnsamd.so')}; seeds {seeds}; twins per seed {args.twins}; arms {args.arms}")
print("arm seed twin | training views: GPU, oracle (base, twin) | held-out: GPU, oracle (base, twin)")
table = {}
for arm in args.arms.split(","):
    os.environ.update(ARMS[arm])  # read by NerfactoTrainStep.__init__
    for seed in seeds:
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"psnr_scene_s{seed}.npz")))
        main_log2, prop_log2, init_seed = (int(v) for v in g["cfg"])
        cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, main_log2),
                              prop_grids=(orc.HashGridCfg(5, 16, 128, prop_log2), orc.HashGridCfg(5, 16, 256, prop_log2)),
                              num_images=S.N_TRAIN, appearance_embed_dim=0)
        ot = (g["psnr_views"][tr].mean(), g["psnr_views_twin"][tr].mean())
        oh = (g["psnr_views"][ho].mean(), g["psnr_views_twin"][ho].mean())
        for k in range(args.twins + 1):
            params = orc.init_params(cfg, seed=init_seed)
            if k > 0:
                gen = torch.Generator().manual_seed(1000 * seed + k)
                for name, p in params.items():
                    if "hash_table" in name:
                        p.mul_(1.0 + 1e-6 * torch.randn(p.shape, generator=gen))
            a, c = run(cfg, params, seed)
            table.setdefault(arm, {}).setdefault(seed, []).append((a, c))
            print(f"{arm} {seed} {k} | {a:7.3f} {ot[0]:7.3f} {ot[1]:7.3f} | {c:7.3f} {oh[0]:7.3f} {oh[1]:7.3f}", flush=True)
        table[arm][seed] = (np.array(table[arm][seed]), np.mean(ot), np.mean(oh))

print("\narm | GPU - oracle [dB], training views: mean +- s.e. over seeds (seeds with GPU below oracle) | held-out | "
      "first run only (what the test of round 4 compared): training, held-out")
for arm, rows in table.items():
    dt = np.array([rows[s][0][:, 0].mean() - rows[s][1] for s in seeds])
    dh = np.array([rows[s][0][:, 1].mean() - rows[s][2] for s in seeds])
    d0t = np.array([rows[s][0][0, 0] - float(np.load(os.path.join(ROOT, "tests", "golden", f"psnr_scene_s{s}.npz"))["psnr_views"][tr].mean()) for s in seeds])
    d0h = np.array([rows[s][0][0, 1] - float(np.load(os.path.join(ROOT, "tests", "golden", f"psnr_scene_s{s}.npz"))["psnr_views"][ho].mean()) for s in seeds])
    n = len(seeds)
    se = lambda x: x.std(ddof=1) / math.sqrt(n) if n > 1 else float("nan")  # noqa: E731
    print(f"{arm} | {dt.mean():+.3f} +- {se(dt):.3f} ({int((dt < 0).sum())}/{n} negative) | {dh.mean():+.3f} +- {se(dh):.3f} "
          f"({int((dh < 0).sum())}/{n}) | {d0t.mean():+.3f} +- {se(d0t):.3f} ({int((d0t < 0).sum())}/{n}), {d0h.mean():+.3f} +- {se(d0h):.3f}")
    print(f"    per seed, training: {np.round(dt, 3).tolist()}")
    print(f"    per seed, held-out: {np.round(dh, 3).tolist()}")
# the GPU path's own chaos spread: standard deviation over twins, averaged over seeds
for arm, rows in table.items():
    if args.twins > 0:
        sd = np.mean([rows[s][0][:, 0].std(ddof=1) for s in seeds])
        print(f"{arm}: spread of the GPU path between twins (s.d. of the training-view mean, averaged over seeds): {sd:.3f} dB")
