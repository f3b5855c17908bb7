#!/usr/bin/env python3
"""CPU study (no GPU): if the main-field backward's persistent workgroups emitted the scatter's pass-1 records themselves
(DESIGN 7.1 item 3), how full would their static segments get? One workgroup = the 16-point tiles (it * G + w) * 8 + wave,
G = 256 workgroups, 8 waves; per (workgroup, level, tile) the number of x-pair records (4 per point and level) — against a
static capacity C — and the same with runs of consecutive samples of a 16-point tile that share a cell merged into one set of
records. Positions: the oracle's own final samples of one benchmark batch (default init, and tables ~ N(0, 0.3))."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import nerfacto_oracle as orc

torch.set_num_threads(16)
for init in (None, 0.3):
    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0, table_std=init)
    o, d, cam, tgt = (torch.from_numpy(a) for a in bench.synthetic_rays(1000))
    rs = np.random.RandomState(1)
    jit = [torch.from_numpy(rs.uniform(0, 1, (4096, 1)).astype(np.float32)) for _ in range(3)]
    with torch.no_grad():
        out = orc.nerfacto_forward(params, cfg, o, d, cam[:, 0], jit, training=True)
    t_bins = out["t_bins_list"][-1] if "t_bins_list" in out else None
    assert t_bins is not None, list(out.keys())
    pos = orc.sample_positions(o, d, t_bins).reshape(-1, 3)
    x, sel = orc.normalise_positions(pos, True)
    x = x.numpy().astype(np.float32)
    M = x.shape[0]
    scal = cfg.main_grid.scalings() if hasattr(cfg.main_grid, "scalings") else None
    scal = np.asarray(scal, dtype=np.float32)
    T, sl = 1 << 19, 13
    G, W = 256, 8
    tile_of_point = np.arange(M) // 16
    wg = (tile_of_point // W) % G
    P1, P2 = np.uint32(2654435761), np.uint32(805459861)
    print(f"init={init}: M={M}")
    for l, s in enumerate(scal):
        sx = x * s
        lo = np.floor(sx).astype(np.int64); hi = np.ceil(sx).astype(np.int64)
        cell = (lo[:, 0] * 4099 + lo[:, 1]) * 4099 + lo[:, 2]
        # runs inside a 16-point tile: a new run starts at the tile start or when the cell changes
        new = np.ones(M, bool); new[1:] = cell[1:] != cell[:-1]; new[::16] = True
        recs_plain, recs_merged = [], []
        for q in range(4):
            yy = np.where(q & 1, hi[:, 1], lo[:, 1]).astype(np.uint32); zz = np.where(q & 2, hi[:, 2], lo[:, 2]).astype(np.uint32)
            ia = (lo[:, 0].astype(np.uint32) ^ (yy * P1) ^ (zz * P2)) & np.uint32(T - 1)
            bin_ = (ia >> sl).astype(np.int64)
            recs_plain.append(wg * 64 + bin_)
            recs_merged.append((wg * 64 + bin_)[new])
        cp = np.bincount(np.concatenate(recs_plain), minlength=G * 64)
        run_len = M / new.sum()
        # merged: runs of length 1 leave as 4 pair records, longer ones as 8 single records (2x): count both as upper bound 2x for len>1
        cm = np.bincount(np.concatenate(recs_merged), minlength=G * 64)
        line = f"  L{l:2d} res {int(s):5d}: mean {cp.mean():6.1f} max {cp.max():5d} p99.9 {np.percentile(cp, 99.9):6.0f} | run {run_len:5.2f} merged mean {cm.mean():6.1f} max {cm.max():5d} |"
        for C in (96, 128, 192, 256):
            line += f" over(C={C}) {int(np.maximum(cp - C, 0).sum()):7d}/{int(np.maximum(cm - C, 0).sum()):6d}"
        print(line)
