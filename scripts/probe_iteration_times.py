#!/usr/bin/env python3
"""GPU box: wall time per training iteration INSIDE the real loop of bench.py (alternating update / non-update iterations
from step 10 on), split by kind: HIP events on the launch stream around every Trainer.train_iteration(), median per kind, and
the host-clock average over the whole loop. PROBE_EAGER=1: eager launches instead of hipGraph replay.
Environment: NSAMD_DEFER_MAIN_ADAM, NSAMD_SPLIT_REDUCE, NSAMD_SIDE_STREAM."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402

dev = torch.device("cuda", 0)
F.DIRECT_GRAD = True
eager = os.environ.get("PROBE_EAGER") == "1"
model = bench.build_model(dev, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
tr = bench.Trainer(model, arena, rb, batch, world=1, use_graph=not eager, use_runner=True, pool=pool)
for _ in range(5):
    tr.train_iteration()
tr.finish()
if not eager:
    assert tr.try_capture()
for _ in range(7):
    tr.train_iteration()
tr.finish()
torch.cuda.synchronize()
n = int(os.environ.get("PROBE_STEPS", "100"))
ev, kinds = [], []
t0 = time.perf_counter()
for _ in range(n):
    kinds.append(model.proposal_sampler.updated_this_step())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    tr.train_iteration()
    b.record()
    ev.append((a, b))
tr.finish()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e6
tag = (f"{'eager' if eager else 'graph'} defer={int(tr.defer)} split={int(tr.runner.split_reduce)} "
       f"side={int(tr.runner.side_stream is not None)}")
out = [f"{tag}: loop {wall:7.1f} us/iteration, loss {float(tr.last_loss()):.6f}"]
for kind in (False, True):
    t = sorted(a.elapsed_time(b) * 1e3 for (a, b), k in zip(ev, kinds) if k == kind)
    if t:
        out.append(f"{'updated' if kind else 'not updated'} x{len(t)}: median {t[len(t) // 2]:7.1f} min {t[0]:7.1f}")
print(" | ".join(out))
if os.environ.get("PROBE_DUMP") == "1":  # the sequence itself (is a slow mode periodic?)
    print("sequence (us, * = update):", " ".join(f"{a.elapsed_time(b) * 1e3:.0f}{'*' if k else ''}" for (a, b), k in zip(ev, kinds)))
