"""Proposal network field (reference: nerfstudio/fields/density_fields.py:33-120)."""
from typing import Literal, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _native as N
from .. import functional as F
from ..cameras.rays import RaySamples
from ..field_components.activations import trunc_exp
from ..field_components.base_field_component import check_implementation
from ..field_components.encodings import HashEncoding
from ..field_components.mlp import MLP
from ..field_components.spatial_distortions import SceneContraction, SpatialDistortion
from .base_field import Field, point_spec


def transform_of(spatial_distortion: Optional[SpatialDistortion]) -> int:
    if spatial_distortion is None:
        return N.XFORM_AABB
    if isinstance(spatial_distortion, SceneContraction):
        return N.XFORM_CONTRACT
    raise ValueError("the hip fields support spatial_distortion None (aabb normalisation) or SceneContraction(inf)")


class HashMLPDensityField(Field):
    """A lightweight density field: scene contraction -> hash grid -> MLP -> trunc_exp, one fused gfx950 pipeline
    (csrc/hashgrid.hip + csrc/density_mlp.hip). Arguments as the reference (density_fields.py:46-61)."""

    aabb: Tensor

    def __init__(self, aabb: Tensor, num_layers: int = 2, hidden_dim: int = 64,
                 spatial_distortion: Optional[SpatialDistortion] = None, use_linear: bool = False, num_levels: int = 8,
                 max_res: int = 1024, base_res: int = 16, log2_hashmap_size: int = 18, features_per_level: int = 2,
                 average_init_density: float = 1.0, implementation: Literal["hip"] = "hip") -> None:
        super().__init__()
        check_implementation(implementation, "HashMLPDensityField")
        if not use_linear and num_layers != 2:
            raise ValueError("the hip density head is MLP(num_layers=2): in -> hidden -> 1")
        # state-dict contract of the reference (density_fields.py:63-71): the scene box and three grid hyper-parameters are buffers
        for name, value in (("aabb", aabb), ("max_res", torch.tensor(max_res)), ("num_levels", torch.tensor(num_levels)),
                            ("log2_hashmap_size", torch.tensor(log2_hashmap_size))):
            self.register_buffer(name, value)
        self.spatial_distortion, self.use_linear, self.average_init_density = spatial_distortion, use_linear, average_init_density
        grid = HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res, log2_hashmap_size=log2_hashmap_size,
                            features_per_level=features_per_level, implementation=implementation)
        self.encoding = grid
        if use_linear:  # density_fields.py:81-84: one dense layer on the hash features
            self.linear = torch.nn.Linear(grid.get_out_dim(), 1)
        else:  # `mlp_base.0` is the grid, `mlp_base.1` the two-layer head: the reference's parameter names
            head = MLP(in_dim=grid.get_out_dim(), num_layers=num_layers, layer_width=hidden_dim, out_dim=1, activation=nn.ReLU(),
                       out_activation=None, implementation=implementation)
            self.mlp_base = torch.nn.Sequential(grid, head)
        self._transform = transform_of(spatial_distortion)
        self._box = N.make_aabb(aabb)  # host copy of the scene box: no device sync on the hot path

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, None]:
        spec, shape = point_spec(ray_samples)
        if self.use_linear:
            # gather kernel (normalisation + selector fused) -> dense layer kernel -> trunc_exp; not a fused launch: this is
            # the reference's ablation setting (density_fields.py:107-109), the proposal networks of nerfacto use the MLP
            enc, sel = F.spec_encode(spec, self.encoding.hash_table, self.encoding.spec, self._transform, self._box)
            pre = F.linear(enc, self.linear.weight, self.linear.bias)
            density = self.average_init_density * trunc_exp(pre) * sel[:, None]
            return density.view(*shape, 1), None
        mlp: MLP = self.mlp_base[1]
        density = F.density_field(spec, self.encoding.hash_table, *mlp.param_tensors(), self.encoding.spec,
                                  self._transform, self._box, self.average_init_density)
        return density.view(*shape, 1), None

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> dict:
        return {}
