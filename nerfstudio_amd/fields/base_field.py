"""Field base class (reference: nerfstudio/fields/base_field.py:40-142)."""
from abc import abstractmethod
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import functional as F
from ..cameras.rays import Frustums, RaySamples, pack_of
from ..field_components.field_heads import FieldHeadNames


def point_spec(ray_samples: RaySamples) -> Tuple[F.PointSpec, Tuple[int, ...]]:
    """How the kernels should obtain the sample points of `ray_samples`: straight from the per-ray pack when a
    nerfstudio_amd sampler produced it (positions never touch HBM), otherwise from materialised frustum centres."""
    shape = tuple(ray_samples.frustums.shape)
    pk = pack_of(ray_samples)  # None for the reference's own RaySamples: positions are materialised then
    if pk is not None and ray_samples.frustums.offsets is None and pk.origins.dim() == 2:
        return F.PointSpec(origins=pk.origins, directions=pk.directions, t_bins=pk.t_bins), shape
    return F.PointSpec(positions=ray_samples.frustums.get_positions().reshape(-1, 3)), shape


class Field(nn.Module):
    """Base class for fields."""

    def __init__(self) -> None:
        super().__init__()
        self._sample_locations = None
        self._density_before_activation = None
        self._compute_normals = False

    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None) -> Tensor:
        """Density only, on explicit positions `[*bs,3]` -> `[*bs,1]` (base_field.py:48-68)."""
        del times
        ray_samples = RaySamples(
            frustums=Frustums(
                origins=positions,
                directions=torch.ones_like(positions),
                starts=torch.zeros_like(positions[..., :1]),
                ends=torch.zeros_like(positions[..., :1]),
                pixel_area=torch.ones_like(positions[..., :1]),
            )
        )
        density, _ = self.get_density(ray_samples)
        return density

    @abstractmethod
    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Optional[Tensor]]:
        """Densities `[*bs,1]` and an optional feature tensor."""

    def get_normals(self) -> Tensor:
        """Analytic normals (base_field.py:79-99): minus the normalised gradient of the density pre-activation with respect
        to the sample locations the field recorded in `get_density` — for the hash-grid fields the NORMALISED positions
        (after contraction / box scaling and the selector, nerfacto_field.py:215-217). First order only (no create_graph):
        the normals are constants for the losses built on them, exactly as in the reference."""
        assert self._sample_locations is not None, "Sample locations must be set before calling get_normals."
        assert self._density_before_activation is not None, "Density must be set before calling get_normals."
        assert self._sample_locations.shape[:-1] == self._density_before_activation.shape[:-1], (
            "Sample locations and density must have the same shape besides the last dimension.")
        normals = torch.autograd.grad(self._density_before_activation, self._sample_locations,
                                      grad_outputs=torch.ones_like(self._density_before_activation), retain_graph=True)[0]
        return -torch.nn.functional.normalize(normals, dim=-1)

    @abstractmethod
    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None
                    ) -> Dict[FieldHeadNames, Tensor]:
        """Field outputs conditioned on the density embedding."""

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[FieldHeadNames, Tensor]:
        """base_field.py:113-133. `compute_normals` tells `get_density` (through `self._compute_normals`) to keep the graph
        from the sample locations to the density pre-activation — also under `torch.no_grad()` (eval renders)."""
        self._compute_normals = bool(compute_normals)
        try:
            if compute_normals:
                with torch.enable_grad():
                    density, density_embedding = self.get_density(ray_samples)
            else:
                density, density_embedding = self.get_density(ray_samples)
            field_outputs = self.get_outputs(ray_samples, density_embedding=density_embedding)
            field_outputs[FieldHeadNames.DENSITY] = density
            if compute_normals:
                with torch.enable_grad():
                    field_outputs[FieldHeadNames.NORMALS] = self.get_normals()
        finally:
            self._compute_normals = False
        return field_outputs


def get_normalized_directions(directions: Tensor) -> Tensor:
    """SH encoding input in [0,1] (base_field.py:136-142). The fused kernel applies this itself."""
    return (directions + 1.0) / 2.0
