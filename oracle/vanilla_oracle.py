"""CPU restatement of the reference's vanilla-NeRF path (BASELINE.json configs[0]: the reference's own CPU-runnable case).

TEST INFRASTRUCTURE ONLY, like oracle/nerfacto_oracle.py: nothing in the product imports it. Pinned against fixtures
generated from the reference itself (tests/golden/make_golden.py gen_vanilla -> tests/golden/vanilla.npz,
tests/test_oracle_vs_golden.py). Each function cites the reference lines it follows (paths under
/root/reference/nerfstudio/)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import nerfacto_oracle as orc

F32 = torch.float32


def nerf_encoding(x: Tensor, num_frequencies: int, min_freq_exp: float, max_freq_exp: float,
                  include_input: bool = False) -> Tensor:
    """NeRFEncoding.pytorch_fwd without covariances (field_components/encodings.py:148-189): sin of 2 pi x 2^k and of the
    same plus pi/2; layout [x0 f0..f_{K-1}, x1 f0.., ...] then the shifted copy, then (optionally) the raw input LAST."""
    scaled = 2 * torch.pi * x
    freqs = 2 ** torch.linspace(min_freq_exp, max_freq_exp, num_frequencies)
    s = (scaled[..., None] * freqs).reshape(*scaled.shape[:-1], -1)
    enc = torch.sin(torch.cat([s, s + torch.pi / 2.0], dim=-1))
    if include_input:
        enc = torch.cat([enc, x], dim=-1)
    return enc


def mlp_skip_forward(x: Tensor, params: Dict[str, Tensor], prefix: str, skip_connections: Sequence[int] = (),
                     out_activation: Optional[str] = None) -> Tensor:
    """MLP.pytorch_fwd with skip connections (field_components/mlp.py:143-179): layer i in `skip_connections` sees
    cat([input, x]); ReLU between layers, `out_activation` after the last."""
    n = 0
    while f"{prefix}layers.{n}.weight" in params:
        n += 1
    assert n > 0, prefix
    inp = x
    for i in range(n):
        if i in skip_connections:
            x = torch.cat([inp, x], dim=-1)
        x = x @ params[f"{prefix}layers.{i}.weight"].t() + params[f"{prefix}layers.{i}.bias"]
        if i < n - 1:
            x = torch.relu(x)
    if out_activation == "relu":
        x = torch.relu(x)
    elif out_activation is not None:
        raise ValueError(out_activation)
    return x


@dataclass(frozen=True)
class VanillaCfg:
    """models/vanilla_nerf.py:42-80 + populate_modules :83-131 (NeRFEncoding 10 / 4 frequencies, 8x256 skip (4,), 2x128)."""

    num_coarse_samples: int = 64
    num_importance_samples: int = 128
    pos_frequencies: int = 10
    pos_max_exp: float = 8.0
    dir_frequencies: int = 4
    dir_max_exp: float = 4.0
    base_layers: int = 8
    base_width: int = 256
    head_layers: int = 2
    head_width: int = 128
    skip_connections: Tuple[int, ...] = (4,)
    near_plane: float = 2.0   # base_model.py:48 collider_params
    far_plane: float = 6.0
    background_color: str = "white"
    rgb_loss_coarse_mult: float = 1.0
    rgb_loss_fine_mult: float = 1.0


def init_field_params(cfg: VanillaCfg, seed: int, prefix: str) -> Dict[str, Tensor]:
    """One NeRFField's tensors under the reference's state-dict names (fields/vanilla_nerf_field.py:45-82), seeded numpy
    init with nn.Linear's default bounds."""
    rs = np.random.RandomState(seed)
    pos_dim = 3 * cfg.pos_frequencies * 2 + 3
    dir_dim = 3 * cfg.dir_frequencies * 2 + 3
    p: Dict[str, Tensor] = {}

    def linear(name, fan_in, fan_out):
        b = 1.0 / math.sqrt(fan_in)
        p[name + ".weight"] = torch.from_numpy(rs.uniform(-b, b, (fan_out, fan_in)).astype(np.float32))
        p[name + ".bias"] = torch.from_numpy(rs.uniform(-b, b, (fan_out,)).astype(np.float32))

    for i in range(cfg.base_layers):
        fan_in = pos_dim if i == 0 else (cfg.base_width + pos_dim if i in cfg.skip_connections else cfg.base_width)
        linear(f"{prefix}mlp_base.layers.{i}", fan_in, cfg.base_width)
    linear(f"{prefix}field_output_density.net", cfg.base_width, 1)
    for i in range(cfg.head_layers):
        linear(f"{prefix}mlp_head.layers.{i}", cfg.base_width + dir_dim if i == 0 else cfg.head_width, cfg.head_width)
    linear(f"{prefix}field_heads.0.net", cfg.head_width, 3)
    return p


def nerf_field(params: Dict[str, Tensor], prefix: str, cfg: VanillaCfg, positions: Tensor, directions: Tensor):
    """NeRFField.forward (fields/vanilla_nerf_field.py:84-112 via base_field.py:118-133): density = softplus(Linear(base)),
    rgb = sigmoid(Linear(head(cat([dir_enc, base])))). positions / directions `[..., 3]` (directions per sample)."""
    enc = nerf_encoding(positions, cfg.pos_frequencies, 0.0, cfg.pos_max_exp, include_input=True)
    base = mlp_skip_forward(enc, params, prefix + "mlp_base.", cfg.skip_connections, out_activation="relu")
    density = torch.nn.functional.softplus(
        base @ params[prefix + "field_output_density.net.weight"].t() + params[prefix + "field_output_density.net.bias"])
    denc = nerf_encoding(directions, cfg.dir_frequencies, 0.0, cfg.dir_max_exp, include_input=True)
    head = mlp_skip_forward(torch.cat([denc, base], dim=-1), params, prefix + "mlp_head.", (), out_activation="relu")
    rgb = torch.sigmoid(head @ params[prefix + "field_heads.0.net.weight"].t() + params[prefix + "field_heads.0.net.bias"])
    return density, rgb


def uniform_bins(nears: Tensor, fars: Tensor, num_samples: int, jitter: Optional[Tensor]):
    """UniformSampler = SpacedSampler with the identity spacing (ray_samplers.py:78-155): s and t edges `[N, S+1]`."""
    return orc.piecewise_bins(nears, fars, num_samples, jitter, uniform=True)


def pdf_resample_with_original(s_bins_prev: Tensor, weights: Tensor, nears: Tensor, fars: Tensor, num_samples: int,
                               jitter: Optional[Tensor], histogram_padding: float = 0.01):
    """PDFSampler(include_original=True) (ray_samplers.py:276-372): the new S+1 edges are merged with the existing ones
    and sorted (:356-357) -> S_prev + S + 2 edges, identity spacing -> euclidean."""
    s_new, _, _ = orc.pdf_resample(s_bins_prev, weights, num_samples, jitter, nears, fars,
                                   histogram_padding=histogram_padding, uniform=True)
    merged, _ = torch.sort(torch.cat([s_bins_prev, s_new], dim=-1), dim=-1)
    return merged, orc.spacing_to_euclidean(merged, nears, fars, True)


def vanilla_forward(params: Dict[str, Tensor], cfg: VanillaCfg, origins: Tensor, directions: Tensor,
                    jitters: Optional[Sequence[Tensor]] = None, training: bool = True) -> Dict[str, Tensor]:
    """NeRFModel.get_outputs (models/vanilla_nerf.py:139-196): uniform 64 -> coarse field -> PDF 128 with the original
    edges (193 samples) -> fine field; white background; median depth (DepthRenderer default)."""
    n = origins.shape[0]
    nears, fars = torch.full((n, 1), cfg.near_plane), torch.full((n, 1), cfg.far_plane)
    j0 = jitters[0] if (training and jitters is not None) else None
    j1 = jitters[1] if (training and jitters is not None) else None
    s0, t0 = uniform_bins(nears, fars, cfg.num_coarse_samples, j0)
    out: Dict[str, Tensor] = {}

    def render(prefix, s_bins, t_bins, tag):
        pos = orc.sample_positions(origins, directions, t_bins)
        dirs = directions[:, None, :].expand(pos.shape)
        density, rgb = nerf_field(params, prefix, cfg, pos, dirs)
        w = orc.weights_from_density(t_bins, density[..., 0])
        out["rgb_" + tag] = orc.composite_rgb(rgb, w, cfg.background_color, training)
        out["accumulation_" + tag] = orc.accumulation(w)
        out["depth_" + tag] = orc.depth_median(w, t_bins)[0]
        out["weights_" + tag] = w
        return w

    w0 = render("field_coarse.", s0, t0, "coarse")
    s1, t1 = pdf_resample_with_original(s0, w0.detach(), nears, fars, cfg.num_importance_samples, j1)
    render("field_fine.", s1, t1, "fine")
    out["s_bins_fine"], out["t_bins_fine"], out["s_bins_coarse"], out["t_bins_coarse"] = s1, t1, s0, t0
    return out


def vanilla_losses(out: Dict[str, Tensor], target_rgb: Tensor, cfg: VanillaCfg) -> Dict[str, Tensor]:
    """NeRFModel.get_loss_dict (models/vanilla_nerf.py:198-217): MSE of both renders (RGB targets: the background blend
    for the loss is the identity, renderers.py:160-199)."""
    return {"rgb_loss_coarse": cfg.rgb_loss_coarse_mult * torch.mean((target_rgb - out["rgb_coarse"]) ** 2),
            "rgb_loss_fine": cfg.rgb_loss_fine_mult * torch.mean((target_rgb - out["rgb_fine"]) ** 2)}
