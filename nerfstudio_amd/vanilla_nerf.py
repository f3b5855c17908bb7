"""Vanilla NeRF on the MI355X kernels: the wiring of the reference's NeRFModel (nerfstudio/models/vanilla_nerf.py:40-217,
BASELINE configs[0] — the reference's own CPU-runnable case) over this package's modules: UniformSampler(64) ->
coarse NeRFField -> weights -> PDFSampler(128, include_original=True) -> fine NeRFField (193 samples), white background,
median depth, MSE of both renders. Temporal distortion (D-NeRF) and gradient scaling are not built."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Literal

import torch
from torch import Tensor, nn
from torch.nn import Parameter

from .cameras.rays import RayBundle
from .field_components.encodings import NeRFEncoding
from .field_components.field_heads import FieldHeadNames
from .fields.vanilla_nerf_field import NeRFField
from .model_components.losses import MSELoss
from .model_components.ray_samplers import PDFSampler, UniformSampler
from .model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer
from .model_components.scene_colliders import NearFarCollider


@dataclass
class VanillaModelConfig:
    """models/vanilla_nerf.py:40-57 (+ the base ModelConfig's collider and loss coefficients, base_model.py:38-53)."""

    num_coarse_samples: int = 64
    num_importance_samples: int = 128
    enable_temporal_distortion: bool = False
    use_gradient_scaling: bool = False
    background_color: Literal["random", "last_sample", "black", "white"] = "white"
    near_plane: float = 2.0
    far_plane: float = 6.0
    rgb_loss_coarse_mult: float = 1.0
    rgb_loss_fine_mult: float = 1.0


class NeRFModel(nn.Module):
    """models/vanilla_nerf.py:60-217. `forward(ray_bundle)` = collider + get_outputs (base_model.py:139-150)."""

    def __init__(self, config: VanillaModelConfig) -> None:
        super().__init__()
        if config.enable_temporal_distortion or config.use_gradient_scaling:
            raise NotImplementedError("temporal distortion / gradient scaling are not built for the hip backend")
        self.config = config
        self.populate_modules()

    def populate_modules(self) -> None:
        position_encoding = NeRFEncoding(in_dim=3, num_frequencies=10, min_freq_exp=0.0, max_freq_exp=8.0, include_input=True)
        direction_encoding = NeRFEncoding(in_dim=3, num_frequencies=4, min_freq_exp=0.0, max_freq_exp=4.0, include_input=True)
        self.field_coarse = NeRFField(position_encoding=position_encoding, direction_encoding=direction_encoding)
        self.field_fine = NeRFField(position_encoding=position_encoding, direction_encoding=direction_encoding)
        self.sampler_uniform = UniformSampler(num_samples=self.config.num_coarse_samples)
        self.sampler_pdf = PDFSampler(num_samples=self.config.num_importance_samples)
        self.renderer_rgb = RGBRenderer(background_color=self.config.background_color)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer()
        self.rgb_loss = MSELoss()
        self.collider = NearFarCollider(near_plane=self.config.near_plane, far_plane=self.config.far_plane)

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        return {"fields": list(self.field_coarse.parameters()) + list(self.field_fine.parameters())}

    def get_outputs(self, ray_bundle: RayBundle, jitters=None) -> Dict[str, Tensor]:
        """`jitters` (optional): the two samplers' draws, `[N, 65]` and `[N, 129]` — injected by the parity tests."""
        j0, j1 = jitters if jitters is not None else (None, None)
        ray_samples_uniform = self.sampler_uniform(ray_bundle, jitter=j0)
        field_outputs_coarse = self.field_coarse.forward(ray_samples_uniform)
        weights_coarse = ray_samples_uniform.get_weights(field_outputs_coarse[FieldHeadNames.DENSITY])
        rgb_coarse = self.renderer_rgb(rgb=field_outputs_coarse[FieldHeadNames.RGB], weights=weights_coarse)
        accumulation_coarse = self.renderer_accumulation(weights_coarse)
        depth_coarse = self.renderer_depth(weights_coarse, ray_samples_uniform)
        ray_samples_pdf = self.sampler_pdf(ray_bundle, ray_samples_uniform, weights_coarse, jitter=j1)
        field_outputs_fine = self.field_fine.forward(ray_samples_pdf)
        weights_fine = ray_samples_pdf.get_weights(field_outputs_fine[FieldHeadNames.DENSITY])
        rgb_fine = self.renderer_rgb(rgb=field_outputs_fine[FieldHeadNames.RGB], weights=weights_fine)
        accumulation_fine = self.renderer_accumulation(weights_fine)
        depth_fine = self.renderer_depth(weights_fine, ray_samples_pdf)
        return {
            "rgb_coarse": rgb_coarse, "rgb_fine": rgb_fine,
            "accumulation_coarse": accumulation_coarse, "accumulation_fine": accumulation_fine,
            "depth_coarse": depth_coarse, "depth_fine": depth_fine,
            "weights_coarse": weights_coarse, "weights_fine": weights_fine,  # (not in the reference's dict: for the tests)
        }

    def forward(self, ray_bundle: RayBundle, jitters=None) -> Dict[str, Tensor]:
        return self.get_outputs(self.collider(ray_bundle), jitters=jitters)

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, Tensor]:
        image = batch["image"].to(outputs["rgb_coarse"].device)
        coarse_pred, coarse_image = self.renderer_rgb.blend_background_for_loss_computation(
            pred_image=outputs["rgb_coarse"], pred_accumulation=outputs["accumulation_coarse"], gt_image=image)
        fine_pred, fine_image = self.renderer_rgb.blend_background_for_loss_computation(
            pred_image=outputs["rgb_fine"], pred_accumulation=outputs["accumulation_fine"], gt_image=image)
        return {"rgb_loss_coarse": self.config.rgb_loss_coarse_mult * self.rgb_loss(coarse_image, coarse_pred),
                "rgb_loss_fine": self.config.rgb_loss_fine_mult * self.rgb_loss(fine_image, fine_pred)}
