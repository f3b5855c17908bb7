"""CPU oracle for the nerfacto hot path.  TEST INFRASTRUCTURE ONLY — the product (`nerfstudio_amd/`) never imports it.

A functional fp32 restatement of the reference's `implementation="torch"` path.  torch CPU tensors
are used as the float container (so `autograd` gives the gradient oracle); the hash arithmetic is numpy `uint32`.
All citations are relative to `/root/reference/nerfstudio/`.

Pinned by `tests/test_oracle_vs_golden.py` against fixtures produced by running the reference itself
(`tests/golden/make_golden.py`) and against the seed-free KATs of SURVEY.md §8(c).

Conventions
-----------
* rays are rows: `[N, ...]`; samples along a ray are `[N, S]` (no trailing singleton axis);
* bins: `s_bins [N, S+1]` (normalised "spacing" domain) and `t_bins [N, S+1]` (euclidean distance along the ray);
* parameters live in a plain `dict` keyed with the reference's torch-path `state_dict` names
  (SURVEY.md §5 "Checkpoint / resume"): `hash_table`, `layers.{i}.weight`, ...
* scans that decide integer indices use `torch.cumsum` on CPU, whose fp32 kernel accumulates LEFT-TO-RIGHT IN DOUBLE
  and rounds every output to fp32 (ATen acc_type<float> = double) — the HIP kernels and oracle/hash_oracle.c do the
  same, so they agree bit-for-bit with the reference's own CPU cumsum.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
F32 = torch.float32

# ---------------------------------------------------------------------------------------------------------------
# a8  multiresolution hash encoding, torch-path semantics   (field_components/encodings.py:307-458)
# ---------------------------------------------------------------------------------------------------------------

PRIME_Y = np.uint32(2654435761)  # encodings.py:410
PRIME_Z = np.uint32(805459861)  # encodings.py:410


def hash_level_scalings(num_levels: int, min_res: int, max_res: int) -> Tensor:
    """Per-level grid scale `floor(min_res * growth**l)` evaluated in fp32 (encodings.py:342-344).

    The reference evaluates `np.float64 ** torch.arange(L)`; torch's `__rpow__` wins (array priority) and yields an
    fp32 tensor, which is why the nerfacto main grid tops out at 2047, not 2048.
    """
    if num_levels > 1:
        growth = math.exp((math.log(max_res) - math.log(min_res)) / (num_levels - 1))
    else:
        growth = 1.0
    lv = torch.arange(num_levels)
    return torch.floor(min_res * torch.pow(torch.tensor(growth, dtype=torch.float64).item(), lv)).to(F32)


def hash_corner_index(ix: np.ndarray, iy: np.ndarray, iz: np.ndarray, level: int, table_size: int) -> np.ndarray:
    """Spatial hash of integer grid corners (encodings.py:398-415), in wrap-around uint32 arithmetic.

    The reference hashes in int64 and takes `% table_size`; table_size is always a power of two so only the low bits
    matter and two's-complement uint32 wrap arithmetic is identical (also for negative coordinates).
    """
    assert table_size & (table_size - 1) == 0, "hash table size is 2**log2_hashmap_size in the reference"
    with np.errstate(over="ignore"):
        h = ix.astype(np.int64).astype(np.uint32)
        h = h ^ (iy.astype(np.int64).astype(np.uint32) * PRIME_Y)
        h = h ^ (iz.astype(np.int64).astype(np.uint32) * PRIME_Z)
    return (h & np.uint32(table_size - 1)).astype(np.int64) + level * table_size


def hashgrid_encode(x: Tensor, table: Tensor, scalings: Tensor, table_size: int) -> Tensor:
    """`HashEncoding.pytorch_fwd` (encodings.py:417-458): `[M,3]` in [0,1] -> `[M, L*F]`, level-major.

    Per level: scaled = x*scale; corners are ceil / floor of `scaled` (so an integral coordinate makes both corners
    coincide); blend weight toward the ceil corner is `scaled - floor(scaled)`; blend order x, then y, then z.
    """
    assert x.shape[-1] == 3
    M = x.shape[0]
    L = scalings.numel()
    outs = []
    for lvl in range(L):
        scaled = x * scalings[lvl]
        lo = torch.floor(scaled)
        hi = torch.ceil(scaled)
        w = scaled - lo  # weight of the ceil corner, per axis          (encodings.py:426)
        lo_i = lo.detach().numpy().astype(np.int32)
        hi_i = hi.detach().numpy().astype(np.int32)

        def corner(cx: bool, cy: bool, cz: bool) -> Tensor:
            ix = (hi_i if cx else lo_i)[:, 0]
            iy = (hi_i if cy else lo_i)[:, 1]
            iz = (hi_i if cz else lo_i)[:, 2]
            idx = hash_corner_index(ix, iy, iz, lvl, table_size)
            return table[torch.from_numpy(idx)]

        wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
        # x-blends (encodings.py:446-449): pairs that differ only in the x corner
        cc_c = corner(True, True, True) * wx + corner(False, True, True) * (1 - wx)  # y=c z=c
        fc_c = corner(True, False, True) * wx + corner(False, False, True) * (1 - wx)  # y=f z=c
        ff_f = corner(True, False, False) * wx + corner(False, False, False) * (1 - wx)  # y=f z=f
        cf_f = corner(True, True, False) * wx + corner(False, True, False) * (1 - wx)  # y=c z=f
        # y-blends (encodings.py:451-452)
        z_c = cc_c * wy + fc_c * (1 - wy)
        z_f = cf_f * wy + ff_f * (1 - wy)
        # z-blend (encodings.py:454-456)
        outs.append(z_c * wz + z_f * (1 - wz))
    return torch.cat(outs, dim=-1).reshape(M, -1)


# ---------------------------------------------------------------------------------------------------------------
# a7  L-inf scene contraction   (field_components/spatial_distortions.py:66-69)
# ---------------------------------------------------------------------------------------------------------------


def contract_linf(x: Tensor) -> Tensor:
    mag = x.abs().amax(dim=-1, keepdim=True)
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


# ---------------------------------------------------------------------------------------------------------------
# a12  spherical harmonics, 4 levels (16 comps)   (utils/spherical_harmonics.py:24-93; encodings.py:791-794)
# ---------------------------------------------------------------------------------------------------------------


@torch.no_grad()
def sh_levels4(d: Tensor) -> Tensor:
    """Real SH basis up to degree 3, evaluated on the input AS GIVEN (the torch path feeds (dir+1)/2)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x**2, y**2, z**2
    c = [
        torch.full_like(x, 0.28209479177387814),
        0.4886025119029199 * y,
        0.4886025119029199 * z,
        0.4886025119029199 * x,
        1.0925484305920792 * x * y,
        1.0925484305920792 * y * z,
        0.9461746957575601 * zz - 0.31539156525251999,
        1.0925484305920792 * x * z,
        0.5462742152960396 * (xx - yy),
        0.5900435899266435 * y * (3 * xx - yy),
        2.890611442640554 * x * y * z,
        0.4570457994644658 * y * (5 * zz - 1),
        0.3731763325901154 * z * (5 * zz - 3),
        0.4570457994644658 * x * (5 * zz - 1),
        1.445305721320277 * z * (xx - yy),
        0.5900435899266435 * x * (xx - 3 * yy),
    ]
    return torch.stack(c, dim=-1)


# ---------------------------------------------------------------------------------------------------------------
# a9 / a11  tiny MLP and trunc_exp   (field_components/mlp.py:143-179; field_components/activations.py:28-54)
# ---------------------------------------------------------------------------------------------------------------


def mlp_forward(x: Tensor, params: Dict[str, Tensor], prefix: str, out_activation: Optional[str] = None) -> Tensor:
    """Linear chain with ReLU between layers (mlp.py:160-179). Weights `[out,in]` as in `nn.Linear`."""
    n_layers = 0
    while f"{prefix}layers.{n_layers}.weight" in params:
        n_layers += 1
    assert n_layers > 0, prefix
    for i in range(n_layers):
        x = x @ params[f"{prefix}layers.{i}.weight"].t() + params[f"{prefix}layers.{i}.bias"]
        if i < n_layers - 1:
            x = torch.relu(x)
    if out_activation == "sigmoid":
        x = torch.sigmoid(x)
    elif out_activation is not None:
        raise ValueError(out_activation)
    return x


class _TruncExpFn(torch.autograd.Function):
    """exp forward; backward multiplies by exp(clamp(x,-15,15)) (activations.py:28-42)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExpFn.apply


class _GradientScalerFn(torch.autograd.Function):
    """Identity forward; the gradient is multiplied by `scaling` (model_components/losses.py:534-547)."""

    @staticmethod
    def forward(ctx, value, scaling):
        ctx.save_for_backward(scaling)
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        (scaling,) = ctx.saved_tensors
        return g * scaling, None


def scale_gradients_by_distance_squared(density: Tensor, rgb: Tensor, t_bins: Tensor) -> Tuple[Tensor, Tensor]:
    """model_components/losses.py:550-569 on density `[N,S]` / rgb `[N,S,3]`: scaling = clamp(((start + end) / 2)^2, 0, 1)."""
    scaling = torch.square((t_bins[:, :-1] + t_bins[:, 1:]) / 2).clamp(0, 1)
    return _GradientScalerFn.apply(density, scaling), _GradientScalerFn.apply(rgb, scaling[..., None])

# ---------------------------------------------------------------------------------------------------------------
# configuration records (hyper-parameters only; defaults = nerfacto method config, method_configs.py:87-121)
# ---------------------------------------------------------------------------------------------------------------


@dataclass
class HashGridCfg:
    num_levels: int
    min_res: int
    max_res: int
    log2_hashmap_size: int
    features_per_level: int = 2

    @property
    def table_size(self) -> int:
        return 2**self.log2_hashmap_size

    @property
    def out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def scalings(self) -> Tensor:
        return hash_level_scalings(self.num_levels, self.min_res, self.max_res)


@dataclass
class NerfactoCfg:
    """nerfacto defaults (models/nerfacto.py:60-133 overridden by method_configs.py:99-103)."""

    main_grid: HashGridCfg = field(default_factory=lambda: HashGridCfg(16, 16, 2048, 19))
    prop_grids: Tuple[HashGridCfg, ...] = (HashGridCfg(5, 16, 128, 17), HashGridCfg(5, 16, 256, 17))
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    prop_hidden_dim: int = 16
    geo_feat_dim: int = 15
    appearance_embed_dim: int = 32
    num_images: int = 100
    num_proposal_samples: Tuple[int, ...] = (256, 96)
    num_nerf_samples: int = 48
    near_plane: float = 0.05
    far_plane: float = 1000.0
    average_init_density: float = 0.01
    use_scene_contraction: bool = True
    background_color: str = "last_sample"
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    histogram_padding: float = 0.01
    use_average_appearance_embedding: bool = True
    predict_normals: bool = False  # models/nerfacto.py:119-120
    orientation_loss_mult: float = 0.0001  # :103-104
    pred_normal_loss_mult: float = 0.001  # :105-106


def init_params(cfg: NerfactoCfg, seed: int = 0, table_std: Optional[float] = None) -> Dict[str, Tensor]:
    """Random-init parameter dict with the reference's shapes and state-dict names.

    Initialisation distributions follow the reference (hash tables U(-1e-3,1e-3), encodings.py:372-376; nn.Linear
    kaiming-uniform; nn.Embedding N(0,1)) but use a numpy `RandomState` stream so fixtures can be regenerated
    anywhere. `table_std` replaces the table init by N(0, table_std) to make densities non-trivial.
    """
    rs = np.random.RandomState(seed)
    p: Dict[str, Tensor] = {}

    def table(g: HashGridCfg) -> Tensor:
        n = g.table_size * g.num_levels
        if table_std is None:
            t = (rs.uniform(-1.0, 1.0, size=(n, g.features_per_level)) * 1e-3).astype(np.float32)
        else:
            t = (rs.standard_normal(size=(n, g.features_per_level)) * table_std).astype(np.float32)
        return torch.from_numpy(t)

    def linear(prefix: str, idx: int, fan_in: int, fan_out: int) -> None:
        bound = 1.0 / math.sqrt(fan_in)
        p[f"{prefix}layers.{idx}.weight"] = torch.from_numpy(
            rs.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)
        )
        p[f"{prefix}layers.{idx}.bias"] = torch.from_numpy(rs.uniform(-bound, bound, size=(fan_out,)).astype(np.float32))

    p["field.mlp_base.model.0.hash_table"] = table(cfg.main_grid)
    linear("field.mlp_base.model.1.", 0, cfg.main_grid.out_dim, cfg.hidden_dim)
    linear("field.mlp_base.model.1.", 1, cfg.hidden_dim, 1 + cfg.geo_feat_dim)
    head_in = 16 + cfg.geo_feat_dim + cfg.appearance_embed_dim
    linear("field.mlp_head.", 0, head_in, cfg.hidden_dim_color)
    linear("field.mlp_head.", 1, cfg.hidden_dim_color, cfg.hidden_dim_color)
    linear("field.mlp_head.", 2, cfg.hidden_dim_color, 3)
    if cfg.appearance_embed_dim > 0:
        p["field.embedding_appearance.embedding.weight"] = torch.from_numpy(
            rs.standard_normal(size=(cfg.num_images, cfg.appearance_embed_dim)).astype(np.float32)
        )
    for i, g in enumerate(cfg.prop_grids):
        p[f"proposal_networks.{i}.encoding.hash_table"] = table(g)
        # `mlp_base = Sequential(encoding, MLP)` (density_fields.py:80-90): the hash table is registered twice in the
        # reference's state_dict (`encoding.hash_table` and `mlp_base.0.hash_table` alias the same Parameter).
        linear(f"proposal_networks.{i}.mlp_base.1.", 0, g.out_dim, cfg.prop_hidden_dim)
        linear(f"proposal_networks.{i}.mlp_base.1.", 1, cfg.prop_hidden_dim, 1)
    if cfg.predict_normals:  # nerfacto_field.py:181-191 (drawn LAST: the stream of every other tensor is unchanged)
        linear("field.mlp_pred_normals.", 0, cfg.geo_feat_dim + 12, 64)
        linear("field.mlp_pred_normals.", 1, 64, 64)
        linear("field.mlp_pred_normals.", 2, 64, 64)
        bound = 1.0 / math.sqrt(64)
        p["field.field_head_pred_normals.net.weight"] = torch.from_numpy(rs.uniform(-bound, bound, size=(3, 64)).astype(np.float32))
        p["field.field_head_pred_normals.net.bias"] = torch.from_numpy(rs.uniform(-bound, bound, size=(3,)).astype(np.float32))
    return p


# ---------------------------------------------------------------------------------------------------------------
# a6 / a10 / a12  fields
# ---------------------------------------------------------------------------------------------------------------


def normalise_positions(positions: Tensor, contraction: bool, aabb: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """contraction -> (x+2)/4 -> in-range selector -> masked positions (density_fields.py:95-103;
    nerfacto_field.py:205-214). Without contraction: aabb normalisation (data/scene_box.py:62-71)."""
    if contraction:
        pos = (contract_linf(positions) + 2.0) / 4.0
    else:
        assert aabb is not None
        pos = (positions - aabb[0]) / (aabb[1] - aabb[0])
    selector = ((pos > 0.0) & (pos < 1.0)).all(dim=-1)
    pos = pos * selector[..., None]
    return pos, selector


def proposal_density(
    positions: Tensor, params: Dict[str, Tensor], level: int, cfg: NerfactoCfg, aabb: Optional[Tensor] = None
) -> Tensor:
    """`HashMLPDensityField.get_density` (fields/density_fields.py:94-117) on `[..., 3]` positions -> `[...]`."""
    g = cfg.prop_grids[level]
    shape = positions.shape[:-1]
    pos, sel = normalise_positions(positions.reshape(-1, 3), cfg.use_scene_contraction, aabb)
    enc = hashgrid_encode(pos, params[f"proposal_networks.{level}.encoding.hash_table"], g.scalings(), g.table_size)
    pre = mlp_forward(enc, params, f"proposal_networks.{level}.mlp_base.1.")[:, 0]
    dens = cfg.average_init_density * trunc_exp(pre)
    return (dens * sel).reshape(shape)


def nerf_encode(x: Tensor, num_frequencies: int, min_freq_exp: float, max_freq_exp: float) -> Tensor:
    """`NeRFEncoding.pytorch_fwd` without covariances (encodings.py:148-166): sin of [2 pi x 2^k, ... + pi/2]."""
    freqs = 2 ** torch.linspace(min_freq_exp, max_freq_exp, num_frequencies)
    scaled = (2 * torch.pi * x)[..., None] * freqs
    scaled = scaled.reshape(*scaled.shape[:-2], -1)
    return torch.sin(torch.cat([scaled, scaled + torch.pi / 2.0], dim=-1))


def nerfacto_field(
    positions: Tensor,
    directions: Tensor,
    camera_indices: Tensor,
    params: Dict[str, Tensor],
    cfg: NerfactoCfg,
    training: bool = True,
    aabb: Optional[Tensor] = None,
    normals_out: Optional[Dict[str, Tensor]] = None,
) -> Tuple[Tensor, Tensor, Tensor]:
    """`NerfactoField.forward` = get_density + get_outputs (fields/nerfacto_field.py:203-310).

    positions/directions `[M,3]`, camera_indices `[M]` int64 -> density `[M]`, rgb `[M,3]`, geo features `[M,15]`.
    `normals_out` (a dict, cfg.predict_normals): receives "normals" — minus the normalised gradient of the density
    pre-activation with respect to the NORMALISED, selector-masked positions, first order only (base_field.py:79-99,
    nerfacto_field.py:215-223) — and "pred_normals" (nerfacto_field.py:287-295, field_heads.py:190-206), `[M,3]` each.
    """
    g = cfg.main_grid
    pos, sel = normalise_positions(positions, cfg.use_scene_contraction, aabb)
    want_normals = normals_out is not None
    if want_normals:
        grad_mode = torch.enable_grad()
        grad_mode.__enter__()
        if not pos.requires_grad:
            pos.requires_grad_(True)
    enc = hashgrid_encode(pos, params["field.mlp_base.model.0.hash_table"], g.scalings(), g.table_size)
    h = mlp_forward(enc, params, "field.mlp_base.model.1.")
    pre, geo = h[:, 0], h[:, 1:]
    if want_normals:
        grad_mode.__exit__(None, None, None)
    density = cfg.average_init_density * trunc_exp(pre) * sel

    sh = sh_levels4((directions + 1.0) / 2.0)  # base_field.py:136-142; SH is no-grad (encodings.py:791)
    feats = [sh, geo]
    if cfg.appearance_embed_dim > 0:
        emb_w = params["field.embedding_appearance.embedding.weight"]
        if training:
            app = emb_w[camera_indices]  # nerfacto_field.py:252
        elif cfg.use_average_appearance_embedding:
            app = torch.ones((positions.shape[0], cfg.appearance_embed_dim)) * emb_w.mean(dim=0)  # :255-257
        else:
            app = torch.zeros((positions.shape[0], cfg.appearance_embed_dim))  # :259-261
        feats.append(app)
    rgb = mlp_forward(torch.cat(feats, dim=-1), params, "field.mlp_head.", out_activation="sigmoid")
    if want_normals:
        x = torch.cat([nerf_encode(positions, 2, 0.0, 1.0), geo], dim=-1)
        x = mlp_forward(x, params, "field.mlp_pred_normals.")
        x = torch.tanh(x @ params["field.field_head_pred_normals.net.weight"].t() + params["field.field_head_pred_normals.net.bias"])
        normals_out["pred_normals"] = torch.nn.functional.normalize(x, dim=-1)
        with torch.enable_grad():
            grad = torch.autograd.grad(pre, pos, grad_outputs=torch.ones_like(pre), retain_graph=True)[0]
        normals_out["normals"] = -torch.nn.functional.normalize(grad, dim=-1)
    return density, rgb, geo


# ---------------------------------------------------------------------------------------------------------------
# a4  piecewise initial sampler   (model_components/ray_samplers.py:78-128, 225-248)
# ---------------------------------------------------------------------------------------------------------------


def spacing_fn(x: Tensor) -> Tensor:  # ray_samplers.py:244
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def spacing_fn_inv(x: Tensor) -> Tensor:  # ray_samplers.py:245
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


def spacing_to_euclidean(s: Tensor, nears: Tensor, fars: Tensor, uniform: bool = False) -> Tensor:
    """closure built at ray_samplers.py:115-116; nears/fars `[N,1]`. `uniform`: UniformSampler's identity spacing
    function (ray_samplers.py:131-155) instead of the piecewise one."""
    if uniform:
        return s * fars + (1 - s) * nears
    s_near, s_far = spacing_fn(nears), spacing_fn(fars)
    return spacing_fn_inv(s * s_far + (1 - s) * s_near)


def piecewise_bins(nears: Tensor, fars: Tensor, num_samples: int, jitter: Optional[Tensor], uniform: bool = False
                   ) -> Tuple[Tensor, Tensor]:
    """Returns (s_bins, t_bins), both `[N, S+1]`. `jitter` `[N,1]` = the `torch.rand` draw of single-jitter
    stratified training (ray_samplers.py:103-111); None = eval (plain linspace)."""
    N = nears.shape[0]
    edges = torch.linspace(0.0, 1.0, num_samples + 1)[None, :]
    if jitter is not None:
        mid = (edges[:, 1:] + edges[:, :-1]) / 2.0
        upper = torch.cat([mid, edges[:, -1:]], -1)
        lower = torch.cat([edges[:, :1], mid], -1)
        edges = lower + (upper - lower) * jitter
    s_bins = edges.expand(N, num_samples + 1).contiguous()
    return s_bins, spacing_to_euclidean(s_bins, nears, fars, uniform)


# ---------------------------------------------------------------------------------------------------------------
# a13  weights from density   (cameras/rays.py:129-152)
# ---------------------------------------------------------------------------------------------------------------


def weights_from_density(t_bins: Tensor, density: Tensor) -> Tensor:
    """`[N,S+1]`, `[N,S]` -> `[N,S]`: alpha_i * exp(-sum_{j<i} delta_j sigma_j), nan -> 0."""
    deltas = t_bins[:, 1:] - t_bins[:, :-1]  # rays.py:265
    ds = deltas * density
    alphas = 1 - torch.exp(-ds)
    acc = torch.cumsum(ds[:, :-1], dim=-1)
    acc = torch.cat([torch.zeros((ds.shape[0], 1), dtype=ds.dtype), acc], dim=-1)  # rays.py:141-144 (also right for S = 1)
    return torch.nan_to_num(alphas * torch.exp(-acc))


# ---------------------------------------------------------------------------------------------------------------
# a14  PDF resampling   (model_components/ray_samplers.py:276-372, include_original=False)
# ---------------------------------------------------------------------------------------------------------------


def pdf_resample(
    s_bins_prev: Tensor,
    weights: Tensor,
    num_samples: int,
    jitter: Optional[Tensor],
    nears: Tensor,
    fars: Tensor,
    histogram_padding: float = 0.01,
    eps: float = 1e-5,
    debug: Optional[dict] = None,
    uniform: bool = False,
) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns (s_bins `[N,S+1]`, t_bins `[N,S+1]`, inds `[N,S+1]` int64 = the searchsorted result).
    `debug`, if given, receives the intermediate `cdf` and `u` (used by the tie analysis in the tests).

    `jitter` `[N,1]` is the raw `torch.rand` draw (divided by num_bins here, ray_samplers.py:320-322); None = eval.
    The weight sum is `cumsum(w)[-1]` (double-accumulated, rounded once; see module docstring) where the reference
    calls torch.sum; everything downstream follows the reference's operation order exactly.
    """
    S_prev = weights.shape[-1]
    nb = num_samples + 1
    w = weights + histogram_padding
    w_sum = torch.cumsum(w, dim=-1)[:, -1:]  # double-accumulated sum         (ray_samplers.py:306)
    pad = torch.relu(eps - w_sum)
    w = w + pad / S_prev
    w_sum = w_sum + pad
    pdf = w / w_sum
    cdf = torch.minimum(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=-1)  # [N, S_prev+1]
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
    if jitter is not None:
        u = u[None, :] + jitter / nb
    else:
        u = (u + 1.0 / (2 * nb))[None, :].expand(cdf.shape[0], nb)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.detach(), u, side="right")
    if debug is not None:
        debug["cdf"], debug["u"] = cdf.detach(), u
    below = torch.clamp(inds - 1, 0, S_prev)
    above = torch.clamp(inds, 0, S_prev)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(s_bins_prev, -1, below), torch.gather(s_bins_prev, -1, above)
    t = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
    s_bins = (b0 + t * (b1 - b0)).detach()  # gradients stop here            (ray_samplers.py:360)
    return s_bins, spacing_to_euclidean(s_bins, nears, fars, uniform), inds


# ---------------------------------------------------------------------------------------------------------------
# a16-a18  compositing   (model_components/renderers.py)
# ---------------------------------------------------------------------------------------------------------------


def composite_rgb(rgb: Tensor, weights: Tensor, background: str = "last_sample", training: bool = True) -> Tensor:
    """`RGBRenderer.forward/combine_rgb` (renderers.py:72-119, 201-232). rgb `[N,S,3]`, weights `[N,S]`."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights[..., None] * rgb, dim=-2)
    acc = torch.sum(weights, dim=-1, keepdim=True)
    if background == "last_sample":
        comp = comp + rgb[:, -1, :] * (1.0 - acc)
    elif background == "white":
        comp = comp + 1.0 * (1.0 - acc)
    elif background == "black":
        comp = comp + 0.0 * (1.0 - acc)
    elif background != "random":
        raise ValueError(background)
    if not training:
        comp = comp.clamp(0.0, 1.0)
    return comp


def accumulation(weights: Tensor) -> Tensor:  # renderers.py:293-317
    return torch.sum(weights, dim=-1, keepdim=True)


def depth_median(weights: Tensor, t_bins: Tensor) -> Tuple[Tensor, Tensor]:
    """renderers.py:354-364: first sample whose running weight sum reaches 0.5. Returns (depth `[N,1]`, index)."""
    steps = (t_bins[:, :-1] + t_bins[:, 1:]) / 2
    cum = torch.cumsum(weights, dim=-1)
    idx = torch.searchsorted(cum.detach().contiguous(), torch.full((weights.shape[0], 1), 0.5), side="left")
    idx = torch.clamp(idx, 0, steps.shape[-1] - 1)
    return torch.gather(steps, -1, idx), idx


def depth_expected(weights: Tensor, t_bins: Tensor) -> Tensor:
    """renderers.py:365-383; the clip bounds are the GLOBAL min/max of the sample midpoints over the batch."""
    steps = (t_bins[:, :-1] + t_bins[:, 1:]) / 2
    d = torch.sum(weights * steps, dim=-1, keepdim=True) / (torch.sum(weights, dim=-1, keepdim=True) + 1e-10)
    return torch.clip(d, steps.min(), steps.max())


# ---------------------------------------------------------------------------------------------------------------
# a19  proposal losses   (model_components/losses.py:53-154)
# ---------------------------------------------------------------------------------------------------------------

LOSS_EPS = 1.0e-7  # losses.py:35


def _outer_bound(c: Tensor, cp: Tensor, wp: Tensor) -> Tensor:
    """`outer` (losses.py:53-82): for each fine interval, total proposal weight of every proposal interval it
    touches (an upper bound on what the proposal histogram assigns to it)."""
    S1 = wp.shape[-1]
    cum = torch.cat([torch.zeros_like(wp[:, :1]), torch.cumsum(wp, dim=-1)], dim=-1)
    lo = torch.searchsorted(cp[:, :-1].contiguous(), c[:, :-1].contiguous(), side="right") - 1
    lo = torch.clamp(lo, 0, S1 - 1)
    hi = torch.searchsorted(cp[:, 1:].contiguous(), c[:, 1:].contiguous(), side="right")
    hi = torch.clamp(hi, 0, S1 - 1)
    return torch.gather(cum[:, 1:], -1, hi) - torch.gather(cum[:, :-1], -1, lo)


def interlevel_loss(weights_list: Sequence[Tensor], s_bins_list: Sequence[Tensor]) -> Tensor:
    """losses.py:113-131; final-level weights/bins are detached."""
    c = s_bins_list[-1].detach()
    w = weights_list[-1].detach()
    total = 0.0
    for cp, wp in zip(s_bins_list[:-1], weights_list[:-1]):
        bound = _outer_bound(c, cp, wp)
        total = total + torch.mean(torch.clip(w - bound, min=0) ** 2 / (w + LOSS_EPS))
    return total


def distortion_loss(weights: Tensor, s_bins: Tensor) -> Tensor:
    """losses.py:135-154 on the final level."""
    mid = (s_bins[:, 1:] + s_bins[:, :-1]) / 2
    pair = torch.abs(mid[:, :, None] - mid[:, None, :])
    inter = torch.sum(weights * torch.sum(weights[:, None, :] * pair, dim=-1), dim=-1)
    intra = torch.sum(weights**2 * (s_bins[:, 1:] - s_bins[:, :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


# ---------------------------------------------------------------------------------------------------------------
# a1  pinhole ray generation   (cameras/cameras.py:598-634, 655-656, 781-787, 887-909)
# ---------------------------------------------------------------------------------------------------------------


def raygen_pinhole(
    ray_indices: Tensor, c2w: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor
) -> Dict[str, Tensor]:
    """ray_indices `[N,3]` (camera,row,col) int64; c2w `[C,3,4]`; intrinsics `[C]`. Pixel centres at +0.5
    (cameras.py:312-313). Returns origins, directions (unit), pixel_area `[N,1]`, directions_norm `[N,1]`."""
    cam = ray_indices[:, 0]
    y = ray_indices[:, 1].to(F32) + 0.5
    x = ray_indices[:, 2].to(F32) + 0.5
    fxr, fyr, cxr, cyr = fx[cam], fy[cam], cx[cam], cy[cam]
    rot = c2w[cam][:, :3, :3]

    def world_dir(px: Tensor, py: Tensor) -> Tuple[Tensor, Tensor]:
        local = torch.stack([px / fxr, -(py / fyr), -torch.ones_like(px)], dim=-1)  # OpenCV -> OpenGL y flip
        d = torch.sum(local[:, None, :] * rot, dim=-1)
        n = torch.linalg.norm(d, dim=-1, keepdim=True)
        return d / n, n

    d0, n0 = world_dir(x - cxr, y - cyr)
    dxv, _ = world_dir(x - cxr + 1, y - cyr)
    dyv, _ = world_dir(x - cxr, y - cyr + 1)
    dx = torch.sqrt(torch.sum((d0 - dxv) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((d0 - dyv) ** 2, dim=-1))
    return {
        "origins": c2w[cam][:, :3, 3],
        "directions": d0,
        "pixel_area": (dx * dy)[:, None],
        "directions_norm": n0,
    }


# ---------------------------------------------------------------------------------------------------------------
# a15 + wiring  the nerfacto forward   (ray_samplers.py:576-617; models/nerfacto.py:298-348, 363-375)
# ---------------------------------------------------------------------------------------------------------------


def sample_positions(origins: Tensor, directions: Tensor, t_bins: Tensor) -> Tensor:
    """`Frustums.get_positions` (cameras/rays.py:50-59): o + d * (start+end)/2 -> `[N,S,3]`."""
    mid = (t_bins[:, :-1] + t_bins[:, 1:])[..., None]
    return origins[:, None, :] + directions[:, None, :] * mid / 2


def nerfacto_forward(
    params: Dict[str, Tensor],
    cfg: NerfactoCfg,
    origins: Tensor,
    directions: Tensor,
    camera_indices: Tensor,
    jitters: Optional[Sequence[Tensor]] = None,
    training: bool = True,
    anneal: float = 1.0,
    proposal_requires_grad: bool = True,
    aabb: Optional[Tensor] = None,
    use_gradient_scaling: bool = False,
) -> Dict[str, object]:
    """Full hot path for one ray batch. `jitters` = one `[N,1]` uniform draw per sampling level (len = 1 + number of
    proposal levels; `[N, S+1]` = one per bin edge, use_single_jitter=False) in training; ignored (eval sampling) when
    `training` is False. camera_indices `[N]`. use_gradient_scaling: models/nerfacto.py:321-322."""
    N = origins.shape[0]
    near = cfg.near_plane if training else 0.0  # scene_colliders.py:186-191 (reset_near_plane in eval)
    nears = torch.full((N, 1), near)
    fars = torch.full((N, 1), cfg.far_plane)
    n_prop = len(cfg.prop_grids)
    counts = list(cfg.num_proposal_samples) + [cfg.num_nerf_samples]
    weights_list: List[Tensor] = []
    s_bins_list: List[Tensor] = []
    t_bins_list: List[Tensor] = []
    inds_list: List[Tensor] = []
    s_bins = t_bins = weights = None
    for lvl in range(n_prop + 1):
        j = jitters[lvl] if (training and jitters is not None) else None
        if lvl == 0:
            s_bins, t_bins = piecewise_bins(nears, fars, counts[0], j)
        else:
            annealed = torch.pow(weights, anneal)  # ray_samplers.py:601
            s_bins, t_bins, inds = pdf_resample(
                s_bins, annealed, counts[lvl], j, nears, fars, histogram_padding=cfg.histogram_padding
            )
            inds_list.append(inds)
        if lvl < n_prop:
            pos = sample_positions(origins, directions, t_bins)
            if proposal_requires_grad:
                dens = proposal_density(pos, params, lvl, cfg, aabb)
            else:
                with torch.no_grad():
                    dens = proposal_density(pos, params, lvl, cfg, aabb)
            weights = weights_from_density(t_bins, dens)
            weights_list.append(weights)
            s_bins_list.append(s_bins)
            t_bins_list.append(t_bins)
    S = counts[-1]
    pos = sample_positions(origins, directions, t_bins).reshape(-1, 3)
    dirs = directions[:, None, :].expand(N, S, 3).reshape(-1, 3)
    cams = camera_indices.reshape(N, 1).expand(N, S).reshape(-1)
    nrm: Optional[Dict[str, Tensor]] = {} if cfg.predict_normals else None
    density, rgb, _ = nerfacto_field(pos, dirs, cams, params, cfg, training=training, aabb=aabb, normals_out=nrm)
    density, rgb = density.reshape(N, S), rgb.reshape(N, S, 3)
    if use_gradient_scaling:
        density, rgb = scale_gradients_by_distance_squared(density, rgb, t_bins)
    weights = weights_from_density(t_bins, density)
    weights_list.append(weights)
    s_bins_list.append(s_bins)
    t_bins_list.append(t_bins)
    out: Dict[str, object] = {
        "rgb": composite_rgb(rgb, weights, cfg.background_color, training),
        "accumulation": accumulation(weights),
        "expected_depth": depth_expected(weights, t_bins),
        "depth": depth_median(weights.detach(), t_bins)[0],
        "weights_list": weights_list,
        "s_bins_list": s_bins_list,
        "t_bins_list": t_bins_list,
        "inds_list": inds_list,
        "density": density,
        "rgb_samples": rgb,
    }
    for i in range(n_prop):
        out[f"prop_depth_{i}"] = depth_median(weights_list[i], t_bins_list[i])[0]
    if nrm is not None:  # models/nerfacto.py:325-344
        n_s, p_s = nrm["normals"].reshape(N, S, 3), nrm["pred_normals"].reshape(N, S, 3)
        out["normals_samples"], out["pred_normals_samples"] = n_s, p_s

        def render(v: Tensor) -> Tensor:  # NormalsRenderer + safe_normalize (renderers.py:429-449, utils/math.py:214-227)
            n = torch.sum(weights[..., None] * v, dim=-2)
            n = n / (torch.norm(n, dim=-1, keepdim=True) + 1e-10)
            return (n + 1) / 2  # NormalsShader (shaders.py:75)

        out["normals"], out["pred_normals"] = render(n_s), render(p_s)
        if training:
            wd = weights.detach()
            n_dot_v = (n_s * (directions * -1)[:, None, :]).sum(dim=-1)  # losses.py:201-214
            out["rendered_orientation_loss"] = (wd * torch.fmin(torch.zeros_like(n_dot_v), n_dot_v) ** 2).sum(dim=-1)
            out["rendered_pred_normal_loss"] = (wd * (1.0 - torch.sum(n_s.detach() * p_s, dim=-1))).sum(dim=-1)  # :217-222
    return out


def nerfacto_losses(out: Dict[str, object], target_rgb: Tensor, cfg: NerfactoCfg) -> Dict[str, Tensor]:
    """models/nerfacto.py:363-375 with `MSELoss` (losses.py:31); background "last_sample" needs no GT blending
    for RGB targets."""
    wl, sl = out["weights_list"], out["s_bins_list"]
    losses = {
        "rgb_loss": torch.mean((target_rgb - out["rgb"]) ** 2),
        "interlevel_loss": cfg.interlevel_loss_mult * interlevel_loss(wl, sl),
        "distortion_loss": cfg.distortion_loss_mult * distortion_loss(wl[-1], sl[-1]),
    }
    if cfg.predict_normals:  # models/nerfacto.py:379-388
        losses["orientation_loss"] = cfg.orientation_loss_mult * torch.mean(out["rendered_orientation_loss"])
        losses["pred_normal_loss"] = cfg.pred_normal_loss_mult * torch.mean(out["rendered_pred_normal_loss"])
    return losses


def synthetic_rays(num_rays: int, num_images: int, seed: int = 0, origin_scale: float = 0.5):
    """BASELINE.md §2 synthetic batch: origins ~ N(0, 0.5^2), unit directions, camera ids U{0..C-1}, targets U(0,1).
    Uses a numpy RandomState stream so the HIP bench and the oracle see identical rays on any machine."""
    rs = np.random.RandomState(seed)
    o = (rs.standard_normal((num_rays, 3)) * origin_scale).astype(np.float32)
    d = rs.standard_normal((num_rays, 3)).astype(np.float32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    cam = rs.randint(0, num_images, size=(num_rays,)).astype(np.int64)
    tgt = rs.uniform(0, 1, size=(num_rays, 3)).astype(np.float32)
    return torch.from_numpy(o), torch.from_numpy(d.astype(np.float32)), torch.from_numpy(cam), torch.from_numpy(tgt)
