#!/usr/bin/env python3
"""Eval-mode render throughput of the nerfacto path (the "render" half of north_star): a full image of
synthetic pinhole rays through RayGenerator -> NerfactoModel.get_outputs_for_camera_ray_bundle (models/base_model.py:178-205:
chunks of eval_num_rays_per_chunk = 32768 rays; no jitter, near plane 0, mean appearance embedding, nan_to_num + clamp),
random-init weights. GPU box only:  python scripts/bench_render.py [--height 800 --width 800 --frames 5]
Prints one JSON line: rays/s (= pixels/s) and ms per frame."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfstudio_amd.model_components.ray_generators import RayGenerator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=800)
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--module-loop", action="store_true", help="the reference-shaped Python chunk loop + torch.cat (NSAMD_EVAL_RUNNER=0)")
ap.add_argument("--bundle", action="store_true",
                help="build the camera's [H, W] ray bundle first (nsamd_raygen_pinhole over an index list) and render it, as round 5 "
                     "did; default: Model.get_outputs_for_camera with the rays generated inside the chunk loop")
args = ap.parse_args()
if args.module_loop:
    os.environ["NSAMD_EVAL_RUNNER"] = "0"
dev = torch.device("cuda", 0)
model = bench.build_model(dev, seed=0).eval()


class Cams:  # the tensors nerfstudio's `Cameras` exposes to RayGenerator
    pass


H, W = args.height, args.width
cams = Cams()
c2w = np.eye(4, dtype=np.float32)[:3]
c2w[:, 3] = (0.0, 0.0, 0.9)
cams.camera_to_worlds = torch.from_numpy(c2w)[None]
cams.fx = torch.tensor([[0.9 * W]])
cams.fy = torch.tensor([[0.9 * W]])
cams.cx = torch.tensor([[W / 2.0]])
cams.cy = torch.tensor([[H / 2.0]])
cams.height, cams.width = torch.tensor([[H]]), torch.tensor([[W]])
cams.camera_type = torch.tensor([[1]])  # CameraType.PERSPECTIVE
cams.distortion_params = None
gen = RayGenerator(cams).to(dev)
cams.camera_to_worlds = cams.camera_to_worlds.to(dev)
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
idx = torch.stack([torch.zeros_like(yy), yy, xx], dim=-1).reshape(-1, 3).to(dev)


def frame():
    if not (args.bundle or args.module_loop):
        return model.get_outputs_for_camera(cams)  # rays generated chunk by chunk inside the device-side loop
    rb = gen(idx)  # rays of the whole image on the device (nsamd_raygen_pinhole)
    out = model.get_outputs_for_camera_ray_bundle(rb.reshape((H, W)) if hasattr(rb, "reshape") else rb)
    return out


for _ in range(2):
    out = frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.frames):
    out = frame()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.frames
assert out["rgb"].shape[:2] == (H, W) and bool(torch.isfinite(out["rgb"]).all())
# forward-only HBM ceiling = 8 TB/s / 161.9 kB of hash gathers + ray I/O per ray = 49.4 M rays/s
CEILING = 8.0e12 / 161.9e3
print(json.dumps({"metric": "eval render rays/sec (nerfacto, 256 -> 96 -> 48 samples per ray)", "value": round(H * W / dt, 1),
                  "unit": "rays/s", "ms_per_frame": round(dt * 1e3, 2), "image": [H, W],
                  "chunk": model.config.eval_num_rays_per_chunk,
                  "launch": "Python loop over forward + torch.cat (eager)" if args.module_loop else
                            ("device-side chunk loop: one captured kernel schedule per chunk (eval_render.py)"
                             + (", rays from a prebuilt [H,W] bundle" if args.bundle else ", rays generated per chunk (no bundle)")),
                  "forward_ceiling_rays_per_s": round(CEILING, 1), "frac_of_forward_ceiling": round(H * W / dt / CEILING, 4),
                  "data": "synthetic", "dtype": "f32"}))
