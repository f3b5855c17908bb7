#!/usr/bin/env python3
"""Where does the main-table scatter spend its time? Loads the instrumented build (make -C nerfstudio_amd/csrc probe ->
libnsamd_probe.so; lane 0 of every wave stamps the shader clock at phase boundaries) through NSAMD_LIB, runs the benchmark's
training step eagerly and prints mean clocks between the stamps of scatter_route_fine (slots 0-5) and scatter_apply (10-16)
of ONE non-update step. GPU box only:  python scripts/probe_scatter_clocks.py"""
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

device = torch.device("cuda", 0)
N.load()
F.DIRECT_GRAD = True
model = bench.build_model(device, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(device, seed=1000, workload="bounded")
trainer = bench.Trainer(model, arena, rb, batch, world=1, use_graph=False, use_runner=True, pool=pool)
for _ in range(12):
    trainer.train_iteration()
torch.cuda.synchronize()
lib = N.load().cdll
lib.nsamd_probe_set_clocks_scatter.argtypes = [C.c_void_p]
waves = 128 * 16 * 16
buf = torch.zeros(waves, 64, dtype=torch.int64, device="cuda")
while model.proposal_sampler.updated_this_step():  # a step without the proposal backward: only the main table is scattered
    trainer.train_iteration()
assert lib.nsamd_probe_set_clocks_scatter(buf.data_ptr()) == 0
trainer.train_iteration()
torch.cuda.synchronize()
assert lib.nsamd_probe_set_clocks_scatter(None) == 0
t = buf.cpu()


def report(name, rows, labels, first):
    rows = rows[rows[:, first] > 0]
    print(f"-- {name}: {rows.shape[0]} waves, kernel span {(rows[:, labels[-1][0]].max() - rows[:, first].min()).item()} clocks, "
          f"wave lifetime mean {(rows[:, labels[-1][0]] - rows[:, first]).double().mean().item():.0f}")
    prev = first
    for slot, label in labels:
        ok = (rows[:, slot] > 0) & (rows[:, prev] > 0)
        if ok.sum() == 0:
            continue
        d = (rows[ok, slot] - rows[ok, prev]).double()
        print(f"   {prev:2d} -> {slot:2d}  {label:42s} mean {d.mean().item():8.0f}  min {d.min().item():7.0f}  max {d.max().item():8.0f}  (n={int(ok.sum())})")
        prev = slot


report("scatter_route_fine<1024,1,4>", t, [(1, "zero counters, loads, positions + barrier"), (2, "sweep 0: cells, LDS rank, record stores"),
                                          (3, "barrier (overflow vote)"), (4, "segment counts, dynamic reservation"),
                                          (5, "sweep 1 (overflowed records)")], 0)
report("scatter_apply", t, [(11, "zero the LDS tile + barrier"), (12, "static segments"), (13, "dynamic area"),
                            (14, "spill fold"), (15, "barrier"), (16, "convert + store the tile")], 10)
a = t[t[:, 10] > 0]
start = a[:, 10].min()
fin = (a[:, 16] - start).double()
print(f"   apply workgroup finish times (clocks after the first start): min {fin.min().item():.0f} mean {fin.mean().item():.0f} max {fin.max().item():.0f}")
st = (a[:, 10] - start).double()
print(f"   apply workgroup start times: quartiles {[int(v) for v in torch.quantile(st, torch.tensor([0.25, 0.5, 0.75, 1.0], dtype=torch.float64)).tolist()]}")
# per level (blockIdx.y of the apply grid): where the static-segment time goes
bins = 64
per_wg = t.view(-1, 16, 64)  # [workgroup = level * bins + bin][wave][slot]
for level in range(16):
    rows = per_wg[level * bins:(level + 1) * bins].reshape(-1, 64)
    rows = rows[rows[:, 10] > 0]
    if rows.shape[0] == 0:
        continue
    stat = (rows[:, 12] - rows[:, 11]).double()
    life = (rows[:, 16] - rows[:, 10]).double()
    print(f"   apply level {level:2d}: static segments mean {stat.mean().item():8.0f} max {stat.max().item():8.0f}   workgroup lifetime mean {life.mean().item():8.0f}")

