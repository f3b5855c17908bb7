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
    """The reference's constructor contract (vanilla_nerf_field.py:45-58: argument names, defaults, sub-module names — what
    the method config and a checkpoint's state dict address). Integrated encodings and temporal / spatial distortions of the
    sample positions are not built (the vanilla-nerf method config uses neither)."""

    def __init__(self, position_encoding: Encoding = Identity(in_dim=3), direction_encoding: Encoding = Identity(in_dim=3),
                 base_mlp_num_layers: int = 8, base_mlp_layer_width: int = 256, head_mlp_num_layers: int = 2,
                 head_mlp_layer_width: int = 128, skip_connections: Tuple[int] = (4,),
                 field_heads: Optional[Tuple[Type[FieldHead]]] = (RGBFieldHead,), use_integrated_encoding: bool = False,
                 spatial_distortion: Optional[SpatialDistortion] = None) -> None:
        super().__init__()
        for refused, what in ((use_integrated_encoding, "integrated (mip-NeRF) encodings are"),
                              (spatial_distortion is not None, "NeRFField(spatial_distortion=...) is")):
            if refused:
                raise NotImplementedError(f"{what} not built for the hip backend")
        self.position_encoding, self.direction_encoding = position_encoding, direction_encoding
        self.use_integrated_encoding, self.spatial_distortion = False, None
        width_in, width_dir = position_encoding.get_out_dim(), direction_encoding.get_out_dim()
        # trunk: frequency-encoded position -> features (ReLU on the last layer too), density read off the features
        self.mlp_base = MLP(in_dim=width_in, num_layers=base_mlp_num_layers, layer_width=base_mlp_layer_width,
                            skip_connections=skip_connections, out_activation=nn.ReLU())
        features = self.mlp_base.get_out_dim()
        self.field_output_density = DensityFieldHead(in_dim=features)
        # colour branch, only when a head asks for it: [encoded direction, features] -> head MLP -> the heads
        heads = [make() for make in (field_heads or ())]
        if heads:
            self.mlp_head = MLP(in_dim=features + width_dir, num_layers=head_mlp_num_layers, layer_width=head_mlp_layer_width,
                                out_activation=nn.ReLU())
        self.field_heads = nn.ModuleList(heads)
        for head in heads:
            head.set_in_dim(self.mlp_head.get_out_dim())

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        spec, shape = point_spec(ray_samples)
        enc = self.position_encoding
        if isinstance(enc, NeRFEncoding):
            x = enc.spec_forward(spec)  # midpoints o + d (s + e) / 2 formed in the kernel
        else:
            x = enc(ray_samples.frustums.get_positions().reshape(-1, 3))
        features = self.mlp_base(x)
        return self.field_output_density(features).view(*shape, 1), features.view(*shape, -1)

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> Dict[FieldHeadNames, Tensor]:
        if len(self.field_heads) == 0:
            return {}
        # (one head MLP evaluation serves every head: the reference recomputes the identical tensor per head, :119-124)
        view = self.direction_encoding(ray_samples.frustums.directions)
        hidden = self.mlp_head(torch.cat([view, density_embedding], dim=-1))
        return {head.field_head_name: head(hidden) for head in self.field_heads}
