#!/usr/bin/env python3
"""GPU box: wall time of each captured hipGraph variant of the N = 1 training iteration (bench.Trainer), replayed back to
back: proposal networks updated / not updated x main-field Adam of the previous iteration pending / not pending. The
replays train (parameters move), which does not matter for timing. One line per variant: median and minimum of
PROBE_REPLAYS (default 60) replays, HIP events on the launch stream.
Environment: NSAMD_DEFER_MAIN_ADAM, NSAMD_SPLIT_REDUCE, NSAMD_SIDE_STREAM (read by bench.Trainer / NerfactoTrainStep)."""
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
tr = bench.Trainer(model, arena, rb, batch, world=1, use_graph=True, use_runner=True, pool=pool)
for _ in range(12):
    tr.train_iteration()
tr.finish()
assert tr.try_capture()
for _ in range(10):
    tr.train_iteration()
tr.finish()
torch.cuda.synchronize()
n = int(os.environ.get("PROBE_REPLAYS", "60"))
tag = f"defer={int(tr.defer)} split={int(tr.runner.split_reduce)} side={int(tr.runner.side_stream is not None)}"
for key, g in sorted(tr.graphs.items(), key=str):
    tr._push_hyper()
    for _ in range(3):
        g.replay()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        g.replay()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print(f"{tag} variant {str(key):24s} median {t[len(t) // 2]:8.1f} us   min {t[0]:8.1f} us")
