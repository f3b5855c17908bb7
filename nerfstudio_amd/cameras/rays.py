"""Ray datastructures: the layout contract of the boundary (reference: nerfstudio/cameras/rays.py:34-295).

`Frustums`, `RaySamples` and `RayBundle` keep the reference's field names, tensor shapes (`[..., S, 1]` trailing
singleton axes) and `TensorDataclass` behaviour (fields broadcast to a common batch shape, batch-wise indexing / reshape /
flatten / broadcast_to / to — utils/tensor_dataclass.py), so code written against nerfstudio reads the same; the mirror
modules equally accept the reference's own containers (they only use the reference's field names). In addition a
`RaySamples` produced by one of this package's samplers carries a `RayPack`: the dense per-ray arrays
(`origins [N,3]`, `directions [N,3]`, `t_bins / s_bins [N,S+1]`) the HIP kernels consume directly — sample positions
are never materialised in HBM on that path. Everything the reference fields expose is a zero-copy view of the pack.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from .. import functional as F
from ..utils.tensor_dataclass import TensorDataclass


@dataclass
class RayPack:
    """Dense per-ray arrays behind a `RaySamples` (what the kernels read)."""

    origins: Tensor  # [N,3]
    directions: Tensor  # [N,3]
    t_bins: Tensor  # [N,S+1] euclidean bin edges
    s_bins: Optional[Tensor] = None  # [N,S+1] normalised spacing bin edges
    nears: Optional[Tensor] = None  # [N]
    fars: Optional[Tensor] = None  # [N]
    spacing: int = 0  # s -> t map of the initial sampler: 0 = uniform / linear-in-disparity piecewise, 1 = uniform


@dataclass
class Gaussians:
    """Mean `[*bs,3]` and covariance `[*bs,3,3]` of a 3-D Gaussian (utils/math.py:29-39)."""

    mean: Tensor
    cov: Tensor


@dataclass(init=False)
class Frustums(TensorDataclass):
    """Region of space as a frustum (cameras/rays.py:34-104)."""

    origins: Tensor  # [*bs,3]
    directions: Tensor  # [*bs,3]
    starts: Tensor  # [*bs,1]
    ends: Tensor  # [*bs,1]
    pixel_area: Tensor  # [*bs,1]
    offsets: Optional[Tensor] = None

    def __init__(self, origins, directions, starts, ends, pixel_area, offsets=None) -> None:
        self.origins, self.directions, self.starts, self.ends = origins, directions, starts, ends
        self.pixel_area, self.offsets = pixel_area, offsets
        self.__post_init__()

    def get_positions(self) -> Tensor:
        """"Center" of each frustum: o + d * (start + end) / 2 (cameras/rays.py:50-59). Generic-path helper; the
        fused fields compute this inside the hash-encode kernel instead."""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos

    def get_start_positions(self) -> Tensor:
        return self.origins + self.directions * self.starts

    def get_gaussian_blob(self) -> Gaussians:
        """mip-NeRF's Gaussian approximation of the conical frustum (cameras/rays.py:66-81 -> utils/math.py:95-121: eq. 7 of
        arXiv:2103.13415 in its stable form, c = interval centre, h = half width): for integrated encodings, which are not
        on the nerfacto path — a few elementwise torch ops, part of the Frustums interface for completeness. The cone's radius
        at distance 1 is the one whose disc has the pixel's area."""
        if self.offsets is not None:
            raise NotImplementedError()
        radius = torch.sqrt(self.pixel_area) / 1.7724538509055159  # sqrt(area / pi)
        c, h = (self.starts + self.ends) / 2.0, (self.ends - self.starts) / 2.0
        q = 3.0 * c**2 + h**2
        mean = self.origins + self.directions * (c + (2.0 * c * h**2) / q)
        var_along = h**2 / 3 - (4 / 15) * (h**4 * (12 * c**2 - h**2)) / q**2
        var_across = radius**2 * (c**2 / 4 + (5 / 12) * h**2 - (4 / 15) * h**4 / q)
        d = self.directions
        along = d[..., :, None] * d[..., None, :]
        across = torch.eye(3, device=d.device) - d[..., :, None] * (d / torch.clamp((d**2).sum(-1, keepdim=True), min=1e-10))[..., None, :]
        return Gaussians(mean=mean, cov=var_along[..., None] * along + var_across[..., None] * across)

    def set_offsets(self, offsets: Tensor) -> None:
        self.offsets = offsets

    @classmethod
    def get_mock_frustum(cls, device="cpu") -> "Frustums":
        one3, one1 = torch.ones((1, 3), device=device), torch.ones((1, 1), device=device)
        return Frustums(origins=one3, directions=one3.clone(), starts=one1, ends=one1.clone(), pixel_area=one1.clone())


def t_bins_of(ray_samples) -> Tensor:
    """Dense `[*bs, S+1]` euclidean bin edges of any RaySamples (ours or the reference's): the pack when a
    nerfstudio_amd sampler made it, else the contiguous frustum bins (ends[i] == starts[i+1] for every sampler in the
    reference, ray_samplers.py)."""
    pk = getattr(ray_samples, "pack", None)
    if pk is not None:
        return pk.t_bins
    fr = ray_samples.frustums
    return torch.cat([fr.starts[..., 0], fr.ends[..., -1:, 0]], dim=-1)


@dataclass(init=False)
class RaySamples(TensorDataclass):
    """Samples along rays (cameras/rays.py:108-188)."""

    frustums: Frustums
    camera_indices: Optional[Tensor] = None  # [*bs,1]
    deltas: Optional[Tensor] = None  # [*bs,1]
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None
    pack: Optional[RayPack] = None  # set by nerfstudio_amd samplers

    def __init__(self, frustums, camera_indices=None, deltas=None, spacing_starts=None, spacing_ends=None,
                 spacing_to_euclidean_fn=None, metadata=None, times=None, pack=None) -> None:
        self.frustums, self.camera_indices, self.deltas = frustums, camera_indices, deltas
        self.spacing_starts, self.spacing_ends, self.spacing_to_euclidean_fn = spacing_starts, spacing_ends, spacing_to_euclidean_fn
        self.metadata, self.times, self.pack = metadata, times, pack
        self.__post_init__()

    def _map_tensors(self, fn, **overrides):
        # the pack is the dense per-ray layout of THIS batch shape: any batch operation invalidates it
        overrides.setdefault("pack", None)
        return super()._map_tensors(fn, **overrides)

    def _t_bins(self) -> Tensor:
        return t_bins_of(self)

    def get_weights(self, densities: Tensor) -> Tensor:
        """alpha_i * prod_{j<i}(1 - alpha_j) from densities `[N,S,1]` -> `[N,S,1]` (cameras/rays.py:129-152),
        as one HIP kernel (left-to-right transmittance scan per ray) with its own backward."""
        assert self.deltas is not None, "Deltas must be set to compute weights"
        t_bins = self._t_bins()
        n, s1 = t_bins.shape[-2], t_bins.shape[-1]
        w = F.weights_from_density(t_bins.reshape(-1, s1), densities.reshape(-1, s1 - 1))
        return w.view(*densities.shape)


@dataclass(init=False)
class RayBundle(TensorDataclass):
    """A bundle of ray parameters (cameras/rays.py:192-295)."""

    origins: Tensor  # [*bs,3]
    directions: Tensor  # [*bs,3]
    pixel_area: Tensor  # [*bs,1]
    camera_indices: Optional[Tensor] = None  # [*bs,1]
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    def __init__(self, origins, directions, pixel_area, camera_indices=None, nears=None, fars=None, metadata=None,
                 times=None) -> None:
        self.origins, self.directions, self.pixel_area, self.camera_indices = origins, directions, pixel_area, camera_indices
        self.nears, self.fars, self.times = nears, fars, times
        self.metadata = {} if metadata is None else metadata
        self.__post_init__()

    def __len__(self) -> int:
        """Number of rays (cameras/rays.py:224-226) — the reference overrides TensorDataclass.__len__ here."""
        return torch.numel(self.origins) // self.origins.shape[-1]

    def _map(self, fn) -> "RayBundle":
        """`fn(tensor)` on every tensor field (batch shape may change; trailing feature axis kept by the caller)."""
        return self._map_tensors(lambda t, k: fn(t))

    def set_camera_indices(self, camera_index: int) -> None:
        self.camera_indices = torch.ones_like(self.origins[..., 0:1]).long() * camera_index

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        return self.flatten()[start_idx:end_idx]

    def get_ray_samples(
        self,
        bin_starts: Tensor,
        bin_ends: Tensor,
        spacing_starts: Optional[Tensor] = None,
        spacing_ends: Optional[Tensor] = None,
        spacing_to_euclidean_fn: Optional[Callable] = None,
        pack: Optional[RayPack] = None,
    ) -> RaySamples:
        """Samples for each ray from bin edges `[..., S, 1]` (cameras/rays.py:251-295). All fields are views."""
        return ray_samples_of(self, bin_starts, bin_ends, spacing_starts, spacing_ends, spacing_to_euclidean_fn, pack)


def pack_of(ray_samples) -> Optional[RayPack]:
    """The dense per-ray arrays behind `ray_samples` when one of this package's samplers produced it; None for the
    reference's own RaySamples (which has no such field) and after any batch operation."""
    return getattr(ray_samples, "pack", None)


def ray_samples_of(ray_bundle, bin_starts: Tensor, bin_ends: Tensor, spacing_starts: Optional[Tensor] = None,
                   spacing_ends: Optional[Tensor] = None, spacing_to_euclidean_fn: Optional[Callable] = None,
                   pack: Optional[RayPack] = None) -> RaySamples:
    """RayBundle.get_ray_samples (cameras/rays.py:251-295) for ANY bundle with the reference's field names — this
    package's RayBundle or the reference's own (a sampler here may be handed either)."""
    deltas = bin_ends - bin_starts
    cam = ray_bundle.camera_indices[..., None] if ray_bundle.camera_indices is not None else None
    frustums = Frustums(
        origins=ray_bundle.origins[..., None, :],
        directions=ray_bundle.directions[..., None, :],
        starts=bin_starts,
        ends=bin_ends,
        pixel_area=ray_bundle.pixel_area[..., None, :],
    )
    metadata = getattr(ray_bundle, "metadata", None)
    md = {k: v[..., None, :] if isinstance(v, Tensor) else v for k, v in metadata.items()} if metadata else None
    times = getattr(ray_bundle, "times", None)
    return RaySamples(
        frustums=frustums,
        camera_indices=cam,
        deltas=deltas,
        spacing_starts=spacing_starts,
        spacing_ends=spacing_ends,
        spacing_to_euclidean_fn=spacing_to_euclidean_fn,
        metadata=md,
        times=None if times is None else times[..., None],
        pack=pack,
    )


def samples_from_bins(ray_bundle, s_bins: Tensor, t_bins: Tensor, spacing_to_euclidean_fn: Optional[Callable],
                      spacing: int = 0) -> RaySamples:
    """RaySamples over dense `[N,S+1]` bin-edge arrays (what the HIP samplers emit), with the pack attached."""
    pack = RayPack(
        origins=ray_bundle.origins,
        directions=ray_bundle.directions,
        t_bins=t_bins,
        s_bins=s_bins,
        nears=None if ray_bundle.nears is None else ray_bundle.nears.reshape(-1),
        fars=None if ray_bundle.fars is None else ray_bundle.fars.reshape(-1),
        spacing=spacing,
    )
    return ray_samples_of(
        ray_bundle,
        bin_starts=t_bins[..., :-1, None],
        bin_ends=t_bins[..., 1:, None],
        spacing_starts=s_bins[..., :-1, None],
        spacing_ends=s_bins[..., 1:, None],
        spacing_to_euclidean_fn=spacing_to_euclidean_fn,
        pack=pack,
    )
