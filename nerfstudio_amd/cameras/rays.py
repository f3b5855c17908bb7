"""Ray datastructures: the layout contract of the boundary (reference: nerfstudio/cameras/rays.py:34-295).

`Frustums`, `RaySamples` and `RayBundle` keep the reference's field names and tensor shapes (`[..., S, 1]` trailing
singleton axes, broadcast origins/directions) so code written against nerfstudio reads the same. In addition a
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
class Frustums:
    """Region of space as a frustum (cameras/rays.py:34-104)."""

    origins: Tensor  # [*bs,3]
    directions: Tensor  # [*bs,3]
    starts: Tensor  # [*bs,1]
    ends: Tensor  # [*bs,1]
    pixel_area: Tensor  # [*bs,1]
    offsets: Optional[Tensor] = None

    @property
    def shape(self):
        return torch.broadcast_shapes(self.origins.shape[:-1], self.starts.shape[:-1])

    def get_positions(self) -> Tensor:
        """"Center" of each frustum: o + d * (start + end) / 2 (cameras/rays.py:50-59). Generic-path helper; the
        fused fields compute this inside the hash-encode kernel instead."""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos

    def get_start_positions(self) -> Tensor:
        return self.origins + self.directions * self.starts

    def set_offsets(self, offsets: Tensor) -> None:
        self.offsets = offsets

    @classmethod
    def get_mock_frustum(cls, device="cpu") -> "Frustums":
        one3, one1 = torch.ones((1, 3), device=device), torch.ones((1, 1), device=device)
        return Frustums(origins=one3, directions=one3.clone(), starts=one1, ends=one1.clone(), pixel_area=one1.clone())


@dataclass
class RaySamples:
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

    @property
    def shape(self):
        return self.frustums.shape

    def _t_bins(self) -> Tensor:
        if self.pack is not None:
            return self.pack.t_bins
        # contiguous bins: ends[i] == starts[i+1] (true for every sampler in the reference)
        return torch.cat([self.frustums.starts[..., 0], self.frustums.ends[..., -1:, 0]], dim=-1)

    def get_weights(self, densities: Tensor) -> Tensor:
        """alpha_i * prod_{j<i}(1 - alpha_j) from densities `[N,S,1]` -> `[N,S,1]` (cameras/rays.py:129-152),
        as one HIP kernel (left-to-right transmittance scan per ray) with its own backward."""
        assert self.deltas is not None, "Deltas must be set to compute weights"
        t_bins = self._t_bins()
        n, s1 = t_bins.shape[-2], t_bins.shape[-1]
        w = F.weights_from_density(t_bins.reshape(-1, s1), densities.reshape(-1, s1 - 1))
        return w.view(*densities.shape)


@dataclass
class RayBundle:
    """A bundle of ray parameters (cameras/rays.py:192-295)."""

    origins: Tensor  # [*bs,3]
    directions: Tensor  # [*bs,3]
    pixel_area: Tensor  # [*bs,1]
    camera_indices: Optional[Tensor] = None  # [*bs,1]
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    def __len__(self) -> int:
        return torch.numel(self.origins) // self.origins.shape[-1]

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def _map(self, fn) -> "RayBundle":
        kw = {}
        for f_ in fields(self):
            v = getattr(self, f_.name)
            if isinstance(v, Tensor):
                v = fn(v)
            elif isinstance(v, dict):
                v = {k: fn(x) if isinstance(x, Tensor) else x for k, x in v.items()}
            kw[f_.name] = v
        return RayBundle(**kw)

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def __getitem__(self, idx) -> "RayBundle":
        return self._map(lambda t: t[idx])

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
        deltas = bin_ends - bin_starts
        cam = self.camera_indices[..., None] if self.camera_indices is not None else None
        frustums = Frustums(
            origins=self.origins[..., None, :],
            directions=self.directions[..., None, :],
            starts=bin_starts,
            ends=bin_ends,
            pixel_area=self.pixel_area[..., None, :],
        )
        md = {k: v[..., None, :] if isinstance(v, Tensor) else v for k, v in self.metadata.items()} if self.metadata else None
        return RaySamples(
            frustums=frustums,
            camera_indices=cam,
            deltas=deltas,
            spacing_starts=spacing_starts,
            spacing_ends=spacing_ends,
            spacing_to_euclidean_fn=spacing_to_euclidean_fn,
            metadata=md,
            times=None if self.times is None else self.times[..., None],
            pack=pack,
        )


def samples_from_bins(ray_bundle: RayBundle, s_bins: Tensor, t_bins: Tensor, spacing_to_euclidean_fn: Optional[Callable],
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
    return ray_bundle.get_ray_samples(
        bin_starts=t_bins[..., :-1, None],
        bin_ends=t_bins[..., 1:, None],
        spacing_starts=s_bins[..., :-1, None],
        spacing_ends=s_bins[..., 1:, None],
        spacing_to_euclidean_fn=spacing_to_euclidean_fn,
        pack=pack,
    )
