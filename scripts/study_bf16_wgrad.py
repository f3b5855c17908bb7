#!/usr/bin/env python3
"""CPU study (no GPU): how accurate would split-bf16 operands be for the WEIGHT-GRADIENT GEMMs of the main field's MLPs
(profiles/NOTEBOOK.md §7.1 item 2)? The CPU oracle runs training iterations of the benchmark configuration (4096 rays x 48 samples,
full tables); for every linear layer of the main field the layer input X [M, in] and the gradient of its pre-activation
output dY [M, out] are captured, and dW = dY^T X is formed
  * in float64 (the yardstick),
  * in fp32 (what the f32 MFMA chain computes, up to summation order),
  * with both operands split into 2 or 3 bf16 pieces (x = h + m [+ l], each the RNE bf16 of what is left) and the piece
    products accumulated in fp32: 2 pieces -> hh + hm + mh; 3 pieces -> the six products the bf16x3 forward uses.
Reported per layer: max |error| / max |dW| (the form of the tests' gradient tolerance: 2e-5 runner-vs-autograd, 1e-4
against the reference fixtures) and relative L2. STUDY_STEPS (default 3) iterations with Adam in between, so that the
later ones see trained-away-from-init activations; STUDY_TABLE_STD > 0 starts from non-trivial densities as the parity
tests do."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import nerfacto_oracle as orc  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 16))
captured = []  # (layer name, X, holder for dY)
_orig_mlp_forward = orc.mlp_forward


def recording_mlp_forward(x, params, prefix, out_activation=None):
    if not prefix.startswith("field."):
        return _orig_mlp_forward(x, params, prefix, out_activation)
    n_layers = 0
    while f"{prefix}layers.{n_layers}.weight" in params:
        n_layers += 1
    for i in range(n_layers):
        y = x @ params[f"{prefix}layers.{i}.weight"].t() + params[f"{prefix}layers.{i}.bias"]
        slot = {"name": f"{prefix}layers.{i}", "X": x.detach()}
        if y.requires_grad:
            y.register_hook(lambda g, s=slot: s.__setitem__("dY", g.detach()))
            captured.append(slot)
        x = torch.relu(y) if i < n_layers - 1 else y
    if out_activation == "sigmoid":
        x = torch.sigmoid(x)
    return x


orc.mlp_forward = recording_mlp_forward


def bf16_pieces(t, n):
    out, rest = [], t
    for _ in range(n):
        p = rest.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        rest = rest - p  # exact in fp32
    return out


def split_matmul(dY, X, pieces):
    a, b = bf16_pieces(dY, pieces), bf16_pieces(X, pieces)
    # piece products are exact in fp32 (8 x 8 significant bits); accumulate in fp32, smallest products first
    pairs = [(0, 0), (0, 1), (1, 0)] if pieces == 2 else [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    acc = torch.zeros(dY.shape[1], X.shape[1])
    for i, j in reversed(pairs):
        acc = acc + a[i].t() @ b[j]
    return acc


def main():
    steps = int(os.environ.get("STUDY_STEPS", "3"))
    table_std = float(os.environ.get("STUDY_TABLE_STD", "0"))
    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0, table_std=table_std if table_std > 0 else None)
    plist = list(params.values())
    for p in plist:
        p.requires_grad_(True)
    opt = torch.optim.Adam(plist, lr=1e-2, eps=1e-15)
    n = bench.RAYS_PER_GPU
    rs = np.random.RandomState(1)
    print(f"# {n} rays x 48 samples = {n * 48} points per layer GEMM; table_std = {table_std or 'init U(-1e-4,1e-4)'}")
    print("# step layer                              [out x in]   max|err|/max|dW|: fp32   2 pieces   3 pieces   | rel L2: fp32   2 pieces   3 pieces")
    for it in range(steps):
        o, d, cam, tgt = (torch.from_numpy(a) for a in bench.synthetic_rays(1000 + it))
        jit = [torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)) for _ in range(3)]
        captured.clear()
        opt.zero_grad(set_to_none=True)
        out = orc.nerfacto_forward(params, cfg, o, d, cam[:, 0], jit, training=True)
        sum(orc.nerfacto_losses(out, tgt, cfg).values()).backward()
        for s in captured:
            X, dY = s["X"], s["dY"]
            truth = dY.double().t() @ X.double()
            scale = float(truth.abs().max())
            row = []
            for est in (dY.t() @ X, split_matmul(dY, X, 2), split_matmul(dY, X, 3)):
                err = est.double() - truth
                row.append((float(err.abs().max()) / scale, float(err.norm() / truth.norm())))
            print(f"  {it:3d}  {s['name']:34s} [{dY.shape[1]:2d} x {X.shape[1]:2d}]   "
                  f"{row[0][0]:20.2e} {row[1][0]:10.2e} {row[2][0]:10.2e}   | {row[0][1]:14.2e} {row[1][1]:10.2e} {row[2][1]:10.2e}")
        opt.step()


if __name__ == "__main__":
    main()
