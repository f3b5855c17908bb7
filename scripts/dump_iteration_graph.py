#!/usr/bin/env python3
"""GPU box: the captured training iteration as Graphviz DOT files (hipGraphDebugDotPrint through torch's
CUDAGraph.debug_dump), one per schedule variant — nodes, edges and which nodes the runtime sees as roots / joins. Written to
find out why one more fork / join pair in the captured graph costs ~50 us per replay (profiles/r04_negative_results.txt items
10, 11). Output: gpurun_out/<tag>/iteration_<updated>_<pending>.dot + a one-line summary (nodes, edges, kernel nodes, nodes with
more than one predecessor / successor) per variant.
    python scripts/dump_iteration_graph.py [tag]"""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfstudio_amd import functional as F  # noqa: E402
from nerfstudio_amd.arena import ParamArena  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "graph_dot"
out = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda", 0)
F.DIRECT_GRAD = True
model = bench.build_model(dev, seed=0)
arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
tr = bench.Trainer(model, arena, rb, batch, world=1, use_graph=True, use_runner=True, pool=pool)
for _ in range(5):
    tr.train_iteration()
tr.finish()
tr.warm_variants()
assert tr.defer, "this script dumps the deferred schedule's variants"
for upd in (True, False):
    for pend in (True, False):
        g = torch.cuda.CUDAGraph()
        g.enable_debug_mode()
        with torch.cuda.graph(g):
            tr._deferred_iteration_body(upd, pend)
        path = os.path.join(out, f"iteration_updated{int(upd)}_pending{int(pend)}.dot")
        g.debug_dump(path)
        try:
            dot = open(path).read()
        except OSError:
            print(f"updated={upd} pending={pend}: no DOT file written (debug_dump unsupported on this runtime?)")
            continue
        edges = re.findall(r"\"?(\w+)\"?\s*->\s*\"?(\w+)\"?", dot)
        nodes = set(a for e in edges for a in e)
        preds, succs = {}, {}
        for a, b in edges:
            succs.setdefault(a, set()).add(b)
            preds.setdefault(b, set()).add(a)
        joins = sum(1 for n in nodes if len(preds.get(n, ())) > 1)
        forks = sum(1 for n in nodes if len(succs.get(n, ())) > 1)
        print(f"updated={int(upd)} pending={int(pend)}: {len(nodes)} nodes, {len(edges)} edges, {dot.count('KERNEL')} kernel labels, "
              f"{forks} forks, {joins} joins -> {os.path.relpath(path, ROOT)}")
for name in arena.step_counts:  # (captures executed nothing)
    arena.step_counts[name] = tr._true_steps[name]
