"""CPU stand-ins for the kernel wrappers of nerfstudio_amd.functional, built from the oracle's torch restatements
(oracle/nerfacto_oracle.py). TEST INFRASTRUCTURE: they let the host side of the package — the nn.Module mirror of the
reference interface, the plugin model that lives inside the reference's own classes — run end to end on the CPU, so that
interface mismatches (argument order, shapes, dictionary keys, autograd wiring, parameter names) show up where no GPU is
present. Nothing in the product imports this; on a GPU box the same calls go to the kernels.

    with cpu_kernels.installed(monkeypatch): ...

Same signatures and return conventions as the functions they replace (functional.py); differentiable wherever the kernel
wrapper is.
"""
from typing import Optional, Sequence

import torch
from torch import Tensor

from oracle import nerfacto_oracle as orc


def _positions(spec) -> Tensor:
    if spec.positions is not None:
        return spec.positions
    mid = ((spec.t_bins[:, :-1] + spec.t_bins[:, 1:]) / 2)[..., None]
    return (spec.origins[:, None, :] + spec.directions[:, None, :] * mid).reshape(-1, 3)


def _normalise(pos: Tensor, transform: int, aabb):
    from nerfstudio_amd import _native as N

    if transform == N.XFORM_NONE:
        return pos, torch.ones(pos.shape[0], dtype=torch.bool)
    if transform == N.XFORM_CONTRACT:
        return orc.normalise_positions(pos, True)
    box = aabb
    if not torch.is_tensor(box):  # the fields hand their host copy of the box (N.Aabb)
        box = torch.tensor([list(box.lo), list(box.hi)], dtype=torch.float32)
    return orc.normalise_positions(pos, False, box.reshape(2, 3))


def _mlp(x: Tensor, params: Sequence[Tensor], out_activation: Optional[str] = None) -> Tensor:
    layers = list(zip(params[0::2], params[1::2]))
    for i, (W, b) in enumerate(layers):
        x = x @ W.t() + b
        if i < len(layers) - 1:
            x = torch.relu(x)
    return torch.sigmoid(x) if out_activation == "sigmoid" else x


def density_field(spec, table, W0, b0, W1, b1, grid, transform, aabb, average_init_density):
    pos, sel = _normalise(_positions(spec), transform, aabb)
    enc = orc.hashgrid_encode(pos, table, grid.scalings(), grid.table_size)
    pre = _mlp(enc, [W0, b0, W1, b1])[:, 0]
    return average_init_density * orc.trunc_exp(pre) * sel


def nerfacto_field(spec, table, base_params, head_params, appearance, view_dirs, camera_indices, appearance_const, dir_group,
                   grid, transform, aabb, average_init_density):
    pos, sel = _normalise(_positions(spec), transform, aabb)
    M = pos.shape[0]
    enc = orc.hashgrid_encode(pos, table, grid.scalings(), grid.table_size)
    h = _mlp(enc, list(base_params))
    density = average_init_density * orc.trunc_exp(h[:, 0]) * sel
    rows = torch.arange(M) // int(dir_group)
    feats = [orc.sh_levels4((view_dirs.detach()[rows] + 1.0) / 2.0), h[:, 1:]]
    if appearance is not None:
        if camera_indices is not None:
            feats.append(appearance[camera_indices.reshape(-1)[rows]])
        else:
            feats.append(appearance_const.detach()[None].expand(M, -1))
    rgb = _mlp(torch.cat(feats, dim=-1), list(head_params), "sigmoid")
    return density, rgb


def piecewise_bins(nears, fars, num_samples, jitter, spacing=0):
    n = nears.reshape(-1, 1)
    with torch.no_grad():
        return orc.piecewise_bins(n, fars.reshape(-1, 1), num_samples, jitter, uniform=bool(spacing))


def pdf_resample(s_bins_prev, weights, num_samples, jitter, nears, fars, anneal=1.0, histogram_padding=0.01, eps=1e-5,
                 return_indices=False, anneal_dev=None, spacing=0, include_original=False):
    assert spacing in (0, 1), "stand-in: piecewise (nerfacto) or uniform spacing"
    a = float(anneal_dev) if anneal_dev is not None else float(anneal)
    n, f = nears.reshape(-1, 1), fars.reshape(-1, 1)
    with torch.no_grad():
        s, t, inds = orc.pdf_resample(s_bins_prev, torch.pow(weights.detach(), a), num_samples, jitter, n, f,
                                      histogram_padding=histogram_padding, eps=eps, uniform=bool(spacing))
        if include_original:  # the new edges merged into the existing ones (ray_samplers.py:356-357)
            s = torch.sort(torch.cat([s_bins_prev, s], dim=-1), dim=-1)[0]
            t = orc.spacing_to_euclidean(s, n, f, uniform=bool(spacing))
    return (s, t, inds.to(torch.int32)) if return_indices else (s, t)


def weights_from_density(t_bins, density):
    return orc.weights_from_density(t_bins, density)


def composite(rgb, weights, t_bins=None, background="last_sample", expected_depth=True):
    assert isinstance(background, str), "stand-in: named backgrounds"
    out = orc.composite_rgb(rgb, weights, background, training=True)
    depth = orc.depth_expected(weights, t_bins)[:, 0] if (expected_depth and t_bins is not None) else None
    return out, orc.accumulation(weights)[:, 0], depth


def composite_eval(rgb, weights, t_bins, background="last_sample"):
    with torch.no_grad():
        return (orc.composite_rgb(rgb, weights, background, training=False), orc.accumulation(weights)[:, 0],
                orc.depth_expected(weights, t_bins)[:, 0], orc.depth_median(weights, t_bins)[0][:, 0])


def depth_median(weights, t_bins, return_index=False):
    with torch.no_grad():
        d, idx = orc.depth_median(weights.detach(), t_bins)
    return (d[:, 0], idx.reshape(-1).to(torch.int32)) if return_index else d[:, 0]


def accumulation(weights):
    with torch.no_grad():
        return orc.accumulation(weights)[:, 0]


def interlevel_loss(weights_list, s_bins_list):
    return orc.interlevel_loss(list(weights_list), list(s_bins_list))


def distortion_loss(weights, s_bins):
    return orc.distortion_loss(weights, s_bins)


def scale_gradients_by_distance_squared(density, rgb, t_bins):
    n, s1 = t_bins.shape
    d, r = orc.scale_gradients_by_distance_squared(density.reshape(n, s1 - 1), rgb.reshape(n, s1 - 1, 3), t_bins)
    return d.view_as(density), r.view_as(rgb)


def hashgrid_encode(x, table, grid):
    shape = x.shape[:-1]
    return orc.hashgrid_encode(x.reshape(-1, 3), table, grid.scalings(), grid.table_size).view(*shape, grid.out_dim)


def linear(x, W, b, activation=None):
    y = x @ W.t() + (b if b is not None else 0.0)
    return {None: lambda v: v, "relu": torch.relu, "sigmoid": torch.sigmoid, "softplus": torch.nn.functional.softplus}[activation](y)


def sh4_encode(d):
    return orc.sh_levels4(d.detach())


def nerf_encode(spec, num_frequencies, min_freq_exp, max_freq_exp, include_input=False):
    x = _positions(spec)
    out = orc.nerf_encode(x, num_frequencies, min_freq_exp, max_freq_exp)
    return torch.cat([out, x], dim=-1) if include_input else out


# ---- the packed (instant-ngp) path: oracle/packed_oracle.py -------------------------------------------------------------
def _ray_indices_of(packed_info: Tensor) -> Tensor:
    return torch.repeat_interleave(torch.arange(packed_info.shape[0]), packed_info[:, 1])


def packed_info_from_counts(counts):
    c = counts.to(torch.int64)
    starts = torch.cumsum(c, 0) - c
    return torch.stack([starts, c], dim=-1), int(c.sum())


def occgrid_march(origins, directions, binaries, roi_aabb, step_size, near_plane=0.0, far_plane=1e10, t_min=None, t_max=None,
                  cone_angle=0.0, jitter=None, coarse=None):
    from oracle import packed_oracle as po

    n = origins.shape[0]
    ri, ts, te = po.occgrid_march(origins.numpy(), directions.numpy(), binaries.numpy().astype(bool), list(roi_aabb), step_size,
                                  near_plane=near_plane, far_plane=min(float(far_plane), 3.0e38),
                                  t_min=None if t_min is None else t_min.numpy(), t_max=None if t_max is None else t_max.numpy(),
                                  cone_angle=cone_angle, jitter=None if jitter is None else jitter.reshape(-1).numpy())
    ri = torch.from_numpy(ri).to(torch.int64)
    info = packed_info_from_counts(torch.bincount(ri, minlength=n))[0]
    return ri, torch.from_numpy(ts), torch.from_numpy(te), info


def packed_positions(origins, directions, ray_indices, t_starts, t_ends):
    return origins[ray_indices] + directions[ray_indices] * ((t_starts + t_ends) / 2)[:, None]


def packed_visibility_compact(ray_indices, t_starts, t_ends, sigmas, packed_info, early_stop_eps=1e-4, alpha_thre=0.0):
    from oracle import packed_oracle as po

    n = packed_info.shape[0]
    mask = po.render_visibility_from_density(t_starts, t_ends, sigmas.detach(), ray_indices, n, early_stop_eps, alpha_thre)
    ri = ray_indices[mask]
    info2 = packed_info_from_counts(torch.bincount(ri, minlength=n))[0]
    return ri, t_starts[mask], t_ends[mask], info2, mask.to(torch.uint8)


def packed_weights(sigmas, t_starts, t_ends, packed_info):
    from oracle import packed_oracle as po

    return po.render_weight_from_density(t_starts, t_ends, sigmas, _ray_indices_of(packed_info), packed_info.shape[0])[0]


def packed_composite(rgb, weights, ray_indices, packed_info, t_starts=None, t_ends=None, background="random", eval_mode=False):
    from oracle import packed_oracle as po

    n = packed_info.shape[0]
    bg = background if isinstance(background, str) else "random"
    comp, acc, _ = po.composite_packed(rgb, weights, t_starts if t_starts is not None else torch.zeros_like(weights),
                                       t_ends if t_ends is not None else torch.ones_like(weights), ray_indices, n, bg,
                                       training=not eval_mode)
    depth = None
    if t_starts is not None:  # (unclipped: the DepthRenderer clips, renderers.py:381-383)
        depth = (po.accumulate_along_rays(weights, ((t_starts + t_ends) / 2)[:, None], ray_indices, n) / (acc + 1e-10))[:, 0]
    return comp, acc[:, 0], depth


_NAMES = ("packed_info_from_counts", "occgrid_march", "packed_positions", "packed_visibility_compact", "packed_weights",
          "packed_composite", "density_field", "nerfacto_field", "piecewise_bins", "pdf_resample", "weights_from_density", "composite",
          "composite_eval", "depth_median", "accumulation", "interlevel_loss", "distortion_loss",
          "scale_gradients_by_distance_squared", "hashgrid_encode", "linear", "sh4_encode", "nerf_encode")


class installed:
    """Context manager: nerfstudio_amd.functional's kernel wrappers replaced by the stand-ins above (via monkeypatch)."""

    def __init__(self, monkeypatch):
        self.mp = monkeypatch

    def __enter__(self):
        from nerfstudio_amd import functional as F

        for name in _NAMES:
            self.mp.setattr(F, name, globals()[name])
        return self

    def __exit__(self, *a):
        return False
