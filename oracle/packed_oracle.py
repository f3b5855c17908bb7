"""CPU restatement of the PACKED-sample arithmetic of the instant-ngp path (BASELINE.json
configs[3]). TEST INFRASTRUCTURE ONLY.

The arithmetic lives in a third-party dependency that is absent from /root/reference: **nerfacc == 0.5.2**
(pyproject.toml:34). This module restates the published semantics of the three nerfacc functions the reference calls
after sampling — `pack_info`, `render_weight_from_density`, `accumulate_along_rays` (call sites:
models/instant_ngp.py:192-198, model_components/renderers.py:93-102, 310-314, 369-377) — and
`render_visibility_from_density` (inside `OccGridEstimator.sampling`, called at ray_samplers.py:481-493). Anchors: for
rays with equally many samples the packed results must equal the reference's dense path (`RaySamples.get_weights`,
the dense branches of the renderers), which IS pinned (tests/golden/render.npz, samplers.npz); that is what
tests/test_oracle_vs_golden.py::test_packed_* check. The occupancy-grid marcher (`occgrid_march`, further down) restates
the marching of the instant-ngp paper on nerfacc's grid layout, not nerfacc's `traverse_grids` step for step: its sample
placement cannot be pinned in this container (nerfacc is not installable, the reference's only test of it is the opt-in
tests/utils/test_aabb_intersection.py) — **parity of config 3's sample positions stays unpinned** until it can."""
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


# ---------------------------------------------------------------------------------------------------------------------
# Occupancy-grid marching and bookkeeping (nerfacc.OccGridEstimator) — PARITY UNPINNED (module docstring): nerfacc's
# traverse_grids walks the cells with a DDA and re-anchors its steps at cell boundaries; what is restated here is the
# marching of the instant-ngp paper (arXiv:2201.05989, appendix E.1) on nerfacc's multi-level grid layout, as the product
# kernel (csrc/packed.hip) implements it: a fixed lattice t0 + k dt per ray, a step kept when the cell (finest level
# containing its midpoint) is occupied. The GPU tests pin the kernel to this restatement bit-exactly; nothing pins either to
# nerfacc's exact sample positions.
# ---------------------------------------------------------------------------------------------------------------------
def _ray_box(o, d, lo, hi):
    """Slab test in fp32 (numpy), rays [N,3] -> (hit [N] bool, t0 [N], t1 [N])."""
    import numpy as np

    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (np.float32(1.0) / d).astype(np.float32)
        ta = ((lo - o) * inv).astype(np.float32)
        tb = ((hi - o) * inv).astype(np.float32)
    swap = ta > tb
    ta, tb = np.where(swap, tb, ta), np.where(swap, ta, tb)
    nan = np.isnan(ta) | np.isnan(tb)
    outside = nan & ((o < lo) | (o > hi))
    ta = np.where(nan, -np.float32(3.4028234663852886e38), ta)
    tb = np.where(nan, np.float32(3.4028234663852886e38), tb)
    t0 = np.maximum.reduce([np.full(len(o), -3.4028234663852886e38, np.float32), ta[:, 0], ta[:, 1], ta[:, 2]])
    t1 = np.minimum.reduce([np.full(len(o), 3.4028234663852886e38, np.float32), tb[:, 0], tb[:, 1], tb[:, 2]])
    return (~outside.any(axis=1)) & (t0 <= t1), t0.astype(np.float32), t1.astype(np.float32)


def occgrid_march(origins, directions, binaries, roi_aabb, step_size, near_plane=0.0, far_plane=1e10, t_min=None, t_max=None,
                  cone_angle=0.0, jitter=None, max_steps=1 << 20):
    """-> (ray_indices int64 [n], t_starts fp32 [n], t_ends fp32 [n]); binaries [levels, R, R, R] bool, roi_aabb [6].
    All arithmetic in fp32 with the kernel's operation order (one ray at a time would be the same; vectorised over rays,
    sequential over steps)."""
    import numpy as np

    f = np.float32
    o = np.asarray(origins, f).reshape(-1, 3)
    d = np.asarray(directions, f).reshape(-1, 3)
    B = np.asarray(binaries).astype(bool)
    L, R = B.shape[0], B.shape[1]
    aabb = np.asarray(roi_aabb, f).reshape(6)
    centre = ((aabb[:3] + aabb[3:]) * f(0.5)).astype(f)
    half = ((aabb[3:] - aabb[:3]) * f(0.5)).astype(f)
    outer = f(1 << (L - 1))
    hit, ta, tb = _ray_box(o, d, (centre - half * outer).astype(f), (centre + half * outer).astype(f))
    n = len(o)
    t_lo = np.full(n, near_plane, f) if t_min is None else np.maximum(f(near_plane), np.asarray(t_min, f).reshape(-1))
    t_hi = np.full(n, min(far_plane, 3.4028234663852886e38), f) if t_max is None else np.minimum(f(min(far_plane, 3.4028234663852886e38)), np.asarray(t_max, f).reshape(-1))
    t = np.maximum(t_lo, ta).astype(f)
    t_end = np.minimum(t_hi, tb).astype(f)
    if jitter is not None:
        t = (t + np.asarray(jitter, f).reshape(-1) * f(step_size)).astype(f)
    alive = hit.copy()
    rows = [[] for _ in range(n)]
    step, cone = f(step_size), f(cone_angle)
    for _ in range(max_steps):
        alive &= t < t_end
        if not alive.any():
            break
        dt = np.minimum(np.maximum((t * cone).astype(f), step), f(1e10)).astype(f)
        mid = (t + (dt * f(0.5)).astype(f)).astype(f)
        p = (o + (d * mid[:, None]).astype(f)).astype(f)
        m = np.max((np.abs(p - centre) / half).astype(f), axis=1)
        level = np.zeros(n, np.int64)
        scale = np.ones(n, f)
        for _l in range(L - 1):
            grow = (level < L - 1) & (m > scale)
            scale = np.where(grow, scale * f(2.0), scale).astype(f)
            level = np.where(grow, level + 1, level)
        inside = m <= scale
        lo_l = (centre[None] - (half[None] * scale[:, None]).astype(f)).astype(f)
        u = (((p - lo_l).astype(f) / ((f(2.0) * half[None]).astype(f) * scale[:, None]).astype(f)).astype(f) * f(R)).astype(f)
        c = np.clip(np.floor(u).astype(np.int64), 0, R - 1)
        occ = B[level, c[:, 0], c[:, 1], c[:, 2]]
        keep = alive & inside & occ
        for r in np.nonzero(keep)[0]:
            rows[r].append((t[r], f(t[r] + dt[r])))
        t = np.where(alive, (t + dt).astype(f), t)
    idx = np.concatenate([np.full(len(r), i, np.int64) for i, r in enumerate(rows)]) if n else np.zeros(0, np.int64)
    flat = [x for r in rows for x in r]
    ts = np.array([x[0] for x in flat], f)
    te = np.array([x[1] for x in flat], f)
    return idx, ts, te


def occgrid_thresholds(occs, occ_thre: float = 0.01):
    """nerfacc OccGridEstimator._update tail: binaries = occs > min(mean of the cells evaluated so far, occ_thre)."""
    import numpy as np

    occs = np.asarray(occs, np.float32)
    seen = occs >= 0
    # the mean in double (the product's kernel and its torch path both sum in double), the threshold rounded to fp32 once
    mean = float(occs[seen].astype(np.float64).mean()) if seen.any() else 0.0
    thre = np.float32(min(mean, float(np.float32(occ_thre))))
    return occs > thre
