"""Classic NeRF field (reference: nerfstudio/fields/vanilla_nerf_field.py:30-125): frequency encoding -> 8 x 256 MLP with a
skip connection -> softplus density head; [encoded direction, features] -> 2 x 128 MLP -> sigmoid RGB head. Every dense layer
is a csrc/linear.hip launch (fp32 MFMA, 128 x 128 blocks of W, activation fused), the encodings one kernel each; the sample
midpoints are formed inside the encoding kernel when the samples come from this package's samplers."""
from typing import Dict, Optional, Tuple, Type

import torch
from torch import Tensor, nn

from ..cameras.rays import RaySamples
from ..field_components.encodings import Encoding, Identity, NeRFEncoding
from ..field_components.field_heads import DensityFieldHead, FieldHead, FieldHeadNames, RGBFieldHead
from ..field_components.mlp import MLP
from ..field_components.spatial_distortions import SpatialDistortion
from .base_field import Field, point_spec


class NeRFField(Field):
    """Arguments as the reference (vanilla_nerf_field.py:45-58). Integrated encodings and temporal / spatial distortions
    of the sample positions are not built (the vanilla-nerf method config uses neither)."""

    def __init__(
        self,
        position_encoding: Encoding = Identity(in_dim=3),
        direction_encoding: Encoding = Identity(in_dim=3),
        base_mlp_num_layers: int = 8,
        base_mlp_layer_width: int = 256,
        head_mlp_num_layers: int = 2,
        head_mlp_layer_width: int = 128,
        skip_connections: Tuple[int] = (4,),
        field_heads: Optional[Tuple[Type[FieldHead]]] = (RGBFieldHead,),
        use_integrated_encoding: bool = False,
        spatial_distortion: Optional[SpatialDistortion] = None,
    ) -> None:
        super().__init__()
        if use_integrated_encoding:
            raise NotImplementedError("integrated (mip-NeRF) encodings are not built for the hip backend")
        if spatial_distortion is not None:
            raise NotImplementedError("NeRFField(spatial_distortion=...) is not built for the hip backend")
        self.position_encoding = position_encoding
        self.direction_encoding = direction_encoding
        self.use_integrated_encoding = use_integrated_encoding
        self.spatial_distortion = spatial_distortion
        self.mlp_base = MLP(
            in_dim=self.position_encoding.get_out_dim(),
            num_layers=base_mlp_num_layers,
            layer_width=base_mlp_layer_width,
            skip_connections=skip_connections,
            out_activation=nn.ReLU(),
        )
        self.field_output_density = DensityFieldHead(in_dim=self.mlp_base.get_out_dim())
        if field_heads:
            self.mlp_head = MLP(
                in_dim=self.mlp_base.get_out_dim() + self.direction_encoding.get_out_dim(),
                num_layers=head_mlp_num_layers,
                layer_width=head_mlp_layer_width,
                out_activation=nn.ReLU(),
            )
        self.field_heads = nn.ModuleList([field_head() for field_head in field_heads] if field_heads else [])
        for field_head in self.field_heads:
            field_head.set_in_dim(self.mlp_head.get_out_dim())

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        spec, shape = point_spec(ray_samples)
        if isinstance(self.position_encoding, NeRFEncoding):
            encoded_xyz = self.position_encoding.spec_forward(spec)  # midpoints o + d (s + e) / 2 formed in the kernel
        else:
            encoded_xyz = self.position_encoding(ray_samples.frustums.get_positions().reshape(-1, 3))
        base_mlp_out = self.mlp_base(encoded_xyz)
        density = self.field_output_density(base_mlp_out)
        return density.view(*shape, 1), base_mlp_out.view(*shape, -1)

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> Dict[FieldHeadNames, Tensor]:
        outputs = {}
        for field_head in self.field_heads:
            encoded_dir = self.direction_encoding(ray_samples.frustums.directions)
            mlp_out = self.mlp_head(torch.cat([encoded_dir, density_embedding], dim=-1))
            outputs[field_head.field_head_name] = field_head(mlp_out)
        return outputs
