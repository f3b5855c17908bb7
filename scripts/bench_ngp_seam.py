#!/usr/bin/env python3
"""The instant-ngp training iteration a nerfstudio user gets — `Trainer.train_iteration` (engine/trainer.py:487-531, restated in
tests/trainer_restatement.py) -> `pipeline.DynamicBatchSeam.get_train_loss_dict` (what HipDynamicBatchPipeline adds to the
reference's DynamicBatchPipeline, pipelines/dynamic_batch.py:40-108: the ray batch resized after every iteration from the
samples it kept) -> pipeline.NgpEngine -> ngp_trainer.NgpTrainer — TIMED next to the direct NgpTrainer on the same batches of the
same sizes, on the benchmark's model and synthetic grid (scripts/bench_ngp.py), occupancy refresh every 16th step in both arms.

    python scripts/bench_ngp_seam.py [--steps 64] [--warmup 16]
prints one JSON line."""
import argparse
import collections
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import trainer_restatement as R  # noqa: E402
from scripts.bench_ngp import build_ngp_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--warmup", type=int, default=16)
ap.add_argument("--target-samples", type=int, default=1 << 17, help="DynamicBatchPipelineConfig.target_num_samples (reference: 1 << 18)")
args = ap.parse_args()

from nerfstudio_amd import _native, functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402
from nerfstudio_amd.cameras.rays import RayBundle  # noqa: E402
from nerfstudio_amd.ngp_trainer import NgpTrainer  # noqa: E402
from nerfstudio_amd.pipeline import DynamicBatchSeam  # noqa: E402

_native.load()
F.DIRECT_GRAD = True
dev = torch.device("cuda")
STEP0, TARGET, MAX_PER_RAY = 513, args.target_samples, 1 << 5
parts = [bench.synthetic_rays(1000 + i) for i in range(4)]
pool = [torch.from_numpy(np.concatenate([p[j] for p in parts])).to(dev) for j in range(4)]
total = pool[0].shape[0]
area = torch.full((total, 1), 1e-6, device=dev)


def batch_of_size(step, n):
    n = min(int(n), total)
    lo = (step * 997) % (total - n + 1)
    o, d, cam, tgt = (x[lo:lo + n] for x in pool)
    return RayBundle(origins=o, directions=d, pixel_area=area[:n], camera_indices=cam), {"image": tgt}


def seam_arm():
    model, keep_grid = build_ngp_model(dev)
    groups = {"fields": list(model.field.parameters())}
    opts = R.Optimizers({"fields": {"optimizer": {"lr": 1e-2, "eps": 1e-15}, "scheduler": {"lr_final": 1e-4, "max_steps": 200000}}}, groups)
    sizes = []

    class SeamPipeline(DynamicBatchSeam):
        def __init__(self):
            self.config = SimpleNamespace(target_num_samples=TARGET, max_num_samples_per_ray=MAX_PER_RAY)
            self.dynamic_num_rays_per_batch = TARGET // MAX_PER_RAY
            self.sampler = SimpleNamespace(num_rays_per_batch=self.dynamic_num_rays_per_batch)
            self.datamanager = SimpleNamespace(next_train=self.next_train, train_pixel_sampler=self.sampler)
            self.model = self._model = model
            self.world_size = 1

        def next_train(self, step):
            sizes.append(min(self.sampler.num_rays_per_batch, total))
            return batch_of_size(step, sizes[-1])

        def _update_dynamic_num_rays_per_batch(self, kept):  # pipelines/dynamic_batch.py:71-76
            self.dynamic_num_rays_per_batch = int(self.dynamic_num_rays_per_batch * (self.config.target_num_samples / kept))

        def _update_pixel_samplers(self):
            self.sampler.num_rays_per_batch = self.dynamic_num_rays_per_batch

    pipeline = SeamPipeline()
    trainer = SimpleNamespace(pipeline=pipeline, optimizers=opts, device="cuda:0", mixed_precision=False,
                              gradient_accumulation_steps=collections.defaultdict(lambda: 1),
                              grad_scaler=torch.amp.GradScaler("cuda", enabled=False), config=SimpleNamespace(log_gradients=False))
    pipeline.attach_optimizers(opts, trainer)
    refresher = NgpTrainer(model, None, 1, dev, module_path=True, after_refresh=keep_grid)  # the model's training callback

    def run(step):
        refresher.update_occupancy_grid(step)  # BEFORE_TRAIN_ITERATION (models/instant_ngp.py:150-163), run by the trainer
        R.train_iteration(trainer, step)

    torch.manual_seed(5)
    for i in range(args.warmup):
        run(STEP0 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        run(STEP0 + i)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    eng = pipeline._engine
    assert eng.reason is None and eng.trainer.runner is not None
    return sec, sizes, list(eng.trainer.samples)


def direct_arm(sizes):
    model, keep_grid = build_ngp_model(dev)
    arena = ParamArena({"fields": list(model.field.parameters())}, lr=1e-2, eps=1e-15)
    tr = NgpTrainer(model, arena, sizes[0], dev, after_refresh=keep_grid)
    torch.manual_seed(5)
    for i in range(args.warmup):
        tr.set_batch(*batch_of_size(STEP0 + i, sizes[i]))
        tr.train_iteration(STEP0 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        tr.set_batch(*batch_of_size(STEP0 + i, sizes[i]))
        tr.train_iteration(STEP0 + i)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, list(tr.samples)


F._SCATTER_WS.clear()
seam_s, sizes, kept = seam_arm()
F._SCATTER_WS.clear()
direct_s, kept_direct = direct_arm(sizes)
timed = slice(args.warmup, args.warmup + args.steps)
rays = float(np.sum(sizes[timed]))
print(json.dumps({
    "metric": "ms per instant-ngp training iteration through the DynamicBatchPipeline seam vs the direct trainer (same batches, same sizes)",
    "steps": args.steps, "warmup": args.warmup, "target_num_samples": TARGET,
    "seam_ms": round(seam_s / args.steps * 1e3, 4), "direct_ms": round(direct_s / args.steps * 1e3, 4),
    "seam_over_direct": round(seam_s / direct_s, 4),
    "rays_per_batch": {"first": sizes[0], "mean_timed": round(rays / args.steps, 1), "min": int(min(sizes[timed])), "max": int(max(sizes[timed]))},
    "kept_samples_per_batch_mean": round(float(np.mean(kept[timed])), 1),
    "rays_per_s": {"seam": round(rays / seam_s, 1), "direct": round(rays / direct_s, 1)},
    "same_kept_counts": kept == kept_direct,
    "note": "occupancy refresh every 16th step inside both arms (synthetic 5 % grid kept stationary); the batch size follows the "
            "reference's rule from the kept-sample count the schedule reads anyway"}))
