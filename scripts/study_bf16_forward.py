#!/usr/bin/env python3
"""CPU study (no GPU), companion of study_bf16_wgrad.py: how far would a TWO-piece split-bf16 FORWARD of the main field's
MLPs (both operands x = h + m, products hh + hm + mh, fp32 accumulation — half the matrix work and two thirds of the
operand bytes of the three-piece forward that exists as NSAMD_FIELD_FWD_BF16X3) move density and rgb? Per-point GEMMs
(K = 32 / 64), so nothing averages the 2^-16 per-product error down. Inputs: the oracle's real hash features, directions
and appearance rows at the benchmark configuration. Yardstick float64; tolerances of the GPU tests: rgb 1e-5 per kernel,
1e-4 end to end (north_star); density 1e-4 relative."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import nerfacto_oracle as orc  # noqa: E402


def bf16_pieces(t, n):
    out, rest = [], t
    for _ in range(n):
        p = rest.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        rest = rest - p  # exact in fp32
    return out


torch.set_num_threads(min(16, os.cpu_count() or 16))


def split_linear(x, W, b, pieces):
    """x @ W^T + b with both operands split into `pieces` bf16 pieces (0 = plain fp32, -1 = float64)."""
    if pieces == -1:
        return x.double() @ W.double().t() + b.double()
    if pieces == 0:
        return x @ W.t() + b
    xs, ws = bf16_pieces(x, pieces), bf16_pieces(W, pieces)
    pairs = [(0, 0), (0, 1), (1, 0)] if pieces == 2 else [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    acc = torch.zeros(x.shape[0], W.shape[0])
    for i, j in reversed(pairs):
        acc = acc + xs[i] @ ws[j].t()
    return acc + b


def field_forward(enc, sh, app, params, cfg, pieces):
    dt = torch.float64 if pieces == -1 else torch.float32
    h = enc.to(dt)
    for i in range(2):
        h = split_linear(h, params[f"field.mlp_base.model.1.layers.{i}.weight"], params[f"field.mlp_base.model.1.layers.{i}.bias"], pieces)
        if i == 0:
            h = torch.relu(h)
    pre, geo = h[:, 0], h[:, 1:]
    density = cfg.average_init_density * torch.exp(pre.clamp(max=15.0))
    x = torch.cat([sh.to(dt), geo, app.to(dt)], dim=-1)
    for i in range(3):
        x = split_linear(x, params[f"field.mlp_head.layers.{i}.weight"], params[f"field.mlp_head.layers.{i}.bias"], pieces)
        if i < 2:
            x = torch.relu(x)
    return density, torch.sigmoid(x)


def main():
    table_std = float(os.environ.get("STUDY_TABLE_STD", "0.4"))
    cfg = orc.NerfactoCfg()
    params = {k: v.detach() for k, v in orc.init_params(cfg, seed=0, table_std=table_std).items()}
    # trained-like MLP weights: a few hundred Adam steps are out of reach on the CPU; scale the initial weights up instead
    # (STUDY_WEIGHT_GAIN) so that pre-activations spread as they do after training
    gain = float(os.environ.get("STUDY_WEIGHT_GAIN", "1"))
    if gain != 1.0:
        for k in params:
            if k.startswith("field.mlp") and k.endswith("weight"):
                params[k] = params[k] * gain
    n = bench.RAYS_PER_GPU
    o, d, cam, _ = (torch.from_numpy(a) for a in bench.synthetic_rays(1000))
    rs = np.random.RandomState(1)
    jit = [torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)) for _ in range(3)]
    with torch.no_grad():
        out = orc.nerfacto_forward(params, cfg, o, d, cam[:, 0], jit, training=True)
        t_bins = out["t_bins_list"][-1]
        S = t_bins.shape[1] - 1
        pos = orc.sample_positions(o, d, t_bins).reshape(-1, 3)
        dirs = d[:, None, :].expand(n, S, 3).reshape(-1, 3)
        cams = cam[:, 0].reshape(n, 1).expand(n, S).reshape(-1)
        g = cfg.main_grid
        p01, sel = orc.normalise_positions(pos, cfg.use_scene_contraction, None)
        enc = orc.hashgrid_encode(p01, params["field.mlp_base.model.0.hash_table"], g.scalings(), g.table_size)
        sh = orc.sh_levels4((dirs + 1.0) / 2.0)
        app = params["field.embedding_appearance.embedding.weight"][cams]
        truth_d, truth_rgb = field_forward(enc, sh, app, params, cfg, -1)
        print(f"# {enc.shape[0]} points; table_std {table_std}, MLP weight gain {gain}; density range "
              f"{float(truth_d.min()):.2e} .. {float(truth_d.max()):.2e}")
        print("# variant        max |d rgb|    max rel |d density|   rms rel |d density|")
        for name, pieces in (("fp32", 0), ("2 pieces", 2), ("3 pieces", 3)):
            dn, rgb = field_forward(enc, sh, app, params, cfg, pieces)
            rel = ((dn.double() - truth_d) / truth_d).abs()
            print(f"  {name:10s} {float((rgb.double() - truth_rgb).abs().max()):14.2e} {float(rel.max()):18.2e} {float(rel.pow(2).mean().sqrt()):20.2e}")


if __name__ == "__main__":
    main()
