#!/usr/bin/env python3
"""The training iteration a nerfstudio user gets — `Trainer.train_iteration` (engine/trainer.py:487-531) ->
`pipeline.get_train_loss_dict(step)` (pipelines/base_pipeline.py:290-303) -> the captured kernel schedule behind the
`nerfacto-hip` seam (nerfstudio_amd/pipeline.py) -> `loss.backward()` -> `Optimizers` -> schedulers — TIMED next to the
direct `HipTrainer` line of bench.py on the same box, same batches, same window.

The reference is absent on the GPU box: tests/trainer_restatement.py (pinned to the reference's trainer code by
tests/test_reference_trainer_drive.py, CPU tier) is the trainer; the datamanager stand-in hands out bench.py's HBM-resident
batches as fresh tensors every step (what a device-side datamanager does: base_datamanager.py:506-515).

    python scripts/bench_seam.py [--steps 20] [--warmup 5] [--windows 7]
prints one JSON line: ms_per_step of the three arms (direct over the pool, direct with set_batch, through the seam) and the ratios."""
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

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--windows", type=int, default=7)
ap.add_argument("--profile", action="store_true", help="cProfile of the seam arm's host side (top entries to stderr)")
ap.add_argument("--arms", default="direct_pool,direct_set_batch,seam", help="which arms to run (a kernel trace wants one)")
args = ap.parse_args()

from nerfstudio_amd import _native, functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402
from nerfstudio_amd.cameras.rays import RayBundle  # noqa: E402
from nerfstudio_amd.pipeline import EngineSeam  # noqa: E402
from nerfstudio_amd.trainer import HipTrainer  # noqa: E402

_native.load()
F.DIRECT_GRAD = True
dev = torch.device("cuda")
n = bench.RAYS_PER_GPU
_, _, pool = bench.synthetic_batch(dev, seed=1000)
area = torch.full((n, 1), 1e-6, device=dev)


def batch_of(step):
    s = step % bench.BATCH_SLOTS
    rb = RayBundle(origins=pool["origins"][s], directions=pool["directions"][s], pixel_area=area,
                   camera_indices=pool["cameras"][s][:, None])
    return rb, {"image": pool["target"][s]}


def windows(run_step, finish, first_step):
    """`--windows` timed windows of `--steps` iterations each (consecutive iterations: a trainer's step counter only moves
    forward), device sync on both sides; -> list of ms per step."""
    out, step = [], first_step
    issue = []
    for _ in range(args.windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_step(step)
            step += 1
        finish()
        issue.append((time.perf_counter() - t0) / args.steps * 1e3)  # host time to ISSUE the window (no device sync yet)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / args.steps * 1e3)
    HOST_ISSUE.append(issue)
    return out


HOST_ISSUE = []


def opt_config(groups):
    return {k: {"optimizer": {"lr": 1e-2, "eps": 1e-15}, "scheduler": {"lr_final": 1e-4, "max_steps": 200000}} for k in groups}


results = {}
# ---- arm 1 / 2: the trainer driven directly (bench.py's line); over its own pool, or with set_batch every step
ARMS = args.arms.split(",")
for arm in [a for a in ("direct_pool", "direct_set_batch") if a in ARMS]:
    F._SCATTER_WS.clear()
    model = bench.build_model(dev, seed=0)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    rb, batch = batch_of(0)
    tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=True, use_runner=True, pool=pool if arm == "direct_pool" else None)

    def step_direct(step, tr=tr, arm=arm):
        if arm == "direct_set_batch":
            tr.set_batch(*batch_of(step))
        tr.train_iteration()

    for s in range(2):
        step_direct(s)
    tr.finish()
    assert tr.try_capture()
    for s in range(2, args.warmup):
        step_direct(s)
    results[arm] = windows(step_direct, tr.finish, args.warmup)
    del tr, arena, model

if "seam" not in ARMS:
    med = {k: float(np.median(v)) for k, v in results.items()}
    print(json.dumps({"arms": ARMS, "ms_per_step": med, "windows_ms": results}))
    sys.exit(0)
# ---- arm 3: the restated reference trainer -> seam -> engine
F._SCATTER_WS.clear()
model = bench.build_model(dev, seed=0)
groups = model.get_param_groups()
opts = R.Optimizers(opt_config(groups), groups)


class SeamPipeline(EngineSeam):
    def __init__(self):
        self.datamanager = SimpleNamespace(next_train=batch_of)
        self.model = self._model = model
        self.world_size = 1


pipeline = SeamPipeline()
trainer = SimpleNamespace(pipeline=pipeline, optimizers=opts, device="cuda:0", mixed_precision=False,
                          gradient_accumulation_steps=collections.defaultdict(lambda: 1),
                          grad_scaler=torch.amp.GradScaler("cuda", enabled=False), config=SimpleNamespace(log_gradients=False))
pipeline.attach_optimizers(opts, trainer)


def step_seam(step):
    model.set_step(step)  # BEFORE_TRAIN_ITERATION callbacks
    R.train_iteration(trainer, step)
    model.after_step(step)  # AFTER_TRAIN_ITERATION callbacks


for s in range(args.warmup):
    step_seam(s)
eng = pipeline._engine
assert eng.reason is None, eng.reason
if args.profile:
    import cProfile
    import pstats

    prof = cProfile.Profile()
    prof.enable()
results["seam"] = windows(step_seam, eng.flush, args.warmup)
if args.profile:
    prof.disable()
    pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
assert eng.trainer.graphs is not None, "the seam did not reach the captured schedule"
med = {k: float(np.median(v)) for k, v in results.items()}
if "direct_pool" not in ARMS or "direct_set_batch" not in ARMS:
    print(json.dumps({"arms": ARMS, "ms_per_step": med, "windows_ms": results}))
    sys.exit(0)
# the arms run the SAME iterations window by window (same init, same batches): the paired ratio per window is the comparison;
# the windows themselves differ by +-15 % with the phase of the proposal-update schedule and the gradient sparsity
paired = [a / b for a, b in zip(results["seam"], results["direct_pool"])]
issue = {k: [round(x, 4) for x in v] for k, v in zip(results.keys(), HOST_ISSUE)}
print(json.dumps({
    "metric": "ms per training iteration, 4096 rays (same box, same batches)", "steps": args.steps, "windows": args.windows,
    "direct_pool_ms": round(med["direct_pool"], 4), "direct_set_batch_ms": round(med["direct_set_batch"], 4),
    "seam_ms": round(med["seam"], 4), "seam_over_direct_pool": round(float(np.median(paired)), 4),
    "seam_over_direct_pool_per_window": [round(x, 4) for x in paired],
    "seam_over_direct_set_batch": round(float(np.median([a / b for a, b in zip(results["seam"], results["direct_set_batch"])])), 4),
    "host_issue_ms_per_step": issue,
    "rays_per_s": {k: round(n / (v * 1e-3), 1) for k, v in med.items()},
    "windows_ms": {k: [round(x, 4) for x in v] for k, v in results.items()},
    "note": "seam = tests/trainer_restatement.train_iteration (the reference's Trainer.train_iteration restated) over "
            "pipeline.EngineSeam.get_train_loss_dict; windows are consecutive iterations (steps "
            f"{args.warmup} .. {args.warmup + args.steps * args.windows - 1}), the direct arms run the same steps"}))
