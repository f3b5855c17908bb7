#!/usr/bin/env python3
"""Where does the GATED scatter of the proposal levels spend its time while few rays carry gradient (the sparse phase of the
bench run, profiles/r03_proposal_sparsity.txt)? Instrumented build (make -C nerfstudio_amd/csrc probe) through NSAMD_LIB; the
bench's trainer runs eagerly into the sparse phase, then the backward chain of ONE proposal level is repeated alone with the
clock stamps on (route_runs: slots 20-24, route_fine: 0-5, apply: 10-16). GPU box only:
    python scripts/probe_gated_scatter_clocks.py [level]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["NSAMD_LIB"] = os.path.join(ROOT, "nerfstudio_amd", "libnsamd_probe.so")
os.environ.setdefault("NSAMD_SIDE_STREAM", "0")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nerfstudio_amd import _native as N  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
device = torch.device("cuda", 0)
N.load()
F.DIRECT_GRAD = True
model = bench.build_model(device, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(device, seed=1000, workload="bounded")
trainer = bench.Trainer(model, arena, rb, batch, world=1, use_graph=False, use_runner=True, pool=pool)
for _ in range(int(os.environ.get("PROBE_STEPS", "14"))):
    trainer.train_iteration()
while not model.proposal_sampler.updated_this_step():
    trainer.train_iteration()
trainer.train_iteration()  # an update step: its interlevel gradients, masks and features are in the runner's buffers now
trainer.finish()
torch.cuda.synchronize()
r = trainer.runner
n = r.n
mask = r.prop_ray_masks[level]
print(f"step {trainer.step}: level {level} ({r.counts[level]} samples per ray): {int((mask != 0).sum())} of {n} rays marked")
lib = N.load().cdll
lib.nsamd_probe_set_clocks_scatter.argtypes = [C.c_void_p]
buf = torch.zeros(1 << 17, 64, dtype=torch.int64, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
r.backward_proposals(levels=[level])  # warm
torch.cuda.synchronize()
ev[0].record()
r.backward_proposals(levels=[level])
ev[1].record()
torch.cuda.synchronize()
print(f"chain of level {level} alone (weights_bwd_gate + density_mlp_bwd_gated + gated scatter): {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us")
assert lib.nsamd_probe_set_clocks_scatter(buf.data_ptr()) == 0
r.backward_proposals(levels=[level])
torch.cuda.synchronize()
assert lib.nsamd_probe_set_clocks_scatter(None) == 0
t = buf.cpu()


def report(name, labels, first):
    rows = t[t[:, first] > 0]
    if rows.shape[0] == 0:
        print(f"-- {name}: not launched")
        return
    last = labels[-1][0]
    done = rows[rows[:, last] > 0]
    print(f"-- {name}: {rows.shape[0]} waves entered, {done.shape[0]} ran to the end; kernel span "
          f"{(rows[:, first:].max() - rows[:, first].min()).item()} clocks")
    if done.shape[0] == 0:
        return
    print(f"   full waves: lifetime mean {(done[:, last] - done[:, first]).double().mean().item():.0f} max "
          f"{(done[:, last] - done[:, first]).max().item()}; first start -> last end {(done[:, last].max() - rows[:, first].min()).item()}")
    prev = first
    for slot, label in labels:
        ok = (done[:, slot] > 0) & (done[:, prev] > 0)
        if ok.sum():
            d = (done[ok, slot] - done[ok, prev]).double()
            print(f"   {prev:2d} -> {slot:2d}  {label:44s} mean {d.mean().item():8.0f}  min {d.min().item():7.0f}  max {d.max().item():8.0f}  (n={int(ok.sum())})")
        prev = slot
    st = (rows[:, first] - rows[:, first].min()).double()
    q = torch.quantile(st, torch.tensor([0.5, 0.9, 1.0], dtype=torch.float64)).tolist()
    print(f"   wave start times after the first: median {q[0]:.0f} p90 {q[1]:.0f} last {q[2]:.0f}")


report("scatter_route_runs", [(21, "zero counters, loads, positions, vote"), (22, "sweep 0 (count)"), (23, "barrier, reservation, barrier"),
                              (24, "sweep 1 (emit)")], 20)
report("scatter_route_fine", [(1, "zero counters, loads, positions + barrier"), (2, "sweep 0"), (3, "barrier (overflow vote)"),
                              (4, "segment counts, dynamic reservation"), (5, "sweep 1 (overflowed records)")], 0)
report("scatter_apply", [(11, "zero the LDS tile + barrier"), (12, "static segments"), (13, "dynamic area"), (14, "spill fold"),
                         (15, "barrier"), (16, "convert + read-modify-write")], 10)
