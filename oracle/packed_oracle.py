"""CPU restatement of the PACKED-sample arithmetic of the instant-ngp path (BASELINE.json
configs[3]). TEST INFRASTRUCTURE ONLY.

The arithmetic lives in a third-party dependency that is absent from /root/reference: **nerfacc == 0.5.2**
(pyproject.toml:34). This module restates the published semantics of the three nerfacc functions the reference calls
after sampling — `pack_info`, `render_weight_from_density`, `accumulate_along_rays` (call sites:
models/instant_ngp.py:192-198, model_components/renderers.py:93-102, 310-314, 369-377) — and
`render_visibility_from_density` (inside `OccGridEstimator.sampling`, called at ray_samplers.py:481-493). Anchors: for
rays with equally many samples the packed results must equal the reference's dense path (`RaySamples.get_weights`,
the dense branches of the renderers), which IS pinned (tests/golden/render.npz, samplers.npz); that is what
tests/test_oracle_vs_golden.py::test_packed_* check. The occupancy-grid marcher itself (`traverse_grids`) is NOT
restated: its sample placement cannot be pinned in this container (nerfacc is not installable, the reference's only
test of it is the opt-in tests/utils/test_aabb_intersection.py) — **parity of config 3 stays unpinned** until it can."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor


def pack_info(ray_indices: Tensor, num_rays: int) -> Tensor:
    """nerfacc.pack_info: `[num_rays, 2]` (start, count) of each ray's contiguous run in the packed arrays; samples of a
    ray are contiguous and rays appear in increasing order (what the sampler produces)."""
    counts = torch.bincount(ray_indices, minlength=num_rays)
    starts = torch.cumsum(counts, 0) - counts
    return torch.stack([starts, counts], dim=-1)


def _exclusive_segment_cumsum(x: Tensor, ray_indices: Tensor, num_rays: int) -> Tensor:
    """Exclusive running sum restarting at every ray (packed analogue of rays.py:141-144). Accumulated in double and
    rounded per element, like torch's CPU cumsum does for each row of the dense path (the global scan minus the ray's
    base would cancel catastrophically in fp32)."""
    xd = x.double()
    excl = torch.cumsum(xd, 0) - xd
    info = pack_info(ray_indices, num_rays)
    first = info[:, 0].clamp(max=max(x.shape[0] - 1, 0))
    base = torch.where(info[:, 1] > 0, excl[first], torch.zeros(num_rays, dtype=torch.float64))
    return (excl - base[ray_indices]).to(x.dtype)


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, ray_indices: Tensor,
                               num_rays: int) -> Tuple[Tensor, Tensor, Tensor]:
    """nerfacc.render_weight_from_density: alpha = 1 - exp(-sigma dt), T = exp(-exclusive sum of sigma dt along the
    ray), w = T alpha. Returns (weights, transmittance, alphas), all `[n_samples]`. Same formulas as the dense
    `RaySamples.get_weights` (cameras/rays.py:129-152) without its nan_to_num."""
    sd = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sd)
    trans = torch.exp(-_exclusive_segment_cumsum(sd, ray_indices, num_rays))
    return trans * alphas, trans, alphas


def render_visibility_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, ray_indices: Tensor, num_rays: int,
                                   early_stop_eps: float = 1e-4, alpha_thre: float = 0.0) -> Tensor:
    """nerfacc.render_visibility_from_density: a sample is kept while the transmittance in front of it is at least
    `early_stop_eps` and its own alpha reaches `alpha_thre`."""
    _, trans, alphas = render_weight_from_density(t_starts, t_ends, sigmas, ray_indices, num_rays)
    return (trans >= early_stop_eps) & (alphas >= alpha_thre)


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor], ray_indices: Tensor, num_rays: int) -> Tensor:
    """nerfacc.accumulate_along_rays: per-ray sum of w (values None -> `[num_rays, 1]`) or of w * values (`[num_rays, D]`)."""
    src = weights[:, None] if values is None else weights[:, None] * values
    out = torch.zeros((num_rays, src.shape[-1]), dtype=src.dtype)
    return out.index_add(0, ray_indices, src)


def composite_packed(rgb: Tensor, weights: Tensor, t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor, num_rays: int,
                     background: str = "black", training: bool = True):
    """The packed branches of RGBRenderer / AccumulationRenderer / DepthRenderer("expected") as NGPModel.get_outputs
    uses them (models/instant_ngp.py:199-215; renderers.py:93-119, 310-317, 365-383). "last_sample" is not defined for
    packed samples (renderers.py:95-96)."""
    if background == "last_sample":
        raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = accumulate_along_rays(weights, rgb, ray_indices, num_rays)
    acc = accumulate_along_rays(weights, None, ray_indices, num_rays)
    if background == "white":
        comp = comp + 1.0 * (1.0 - acc)
    elif background == "black":
        comp = comp + 0.0 * (1.0 - acc)
    elif background != "random":
        raise ValueError(background)
    if not training:
        comp = comp.clamp(0.0, 1.0)
    steps = (t_starts + t_ends) / 2
    depth = accumulate_along_rays(weights, steps[:, None], ray_indices, num_rays) / (acc + 1e-10)
    depth = torch.clip(depth, steps.min(), steps.max())
    return comp, acc, depth
