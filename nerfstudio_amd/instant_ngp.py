"""instant-ngp on the MI355X kernels: the wiring of the reference's NGPModel (nerfstudio/models/instant_ngp.py:40-262)
over this package's modules — NerfactoField (hash grid + fused MLPs), the occupancy-grid VolumetricSampler, packed
weights (transmittance scan) and the packed renderer branches (BASELINE configs[3]).

What nerfacc provides to the reference (OccGridEstimator, pack_info, render_weight_from_density, accumulate_along_rays)
is csrc/packed.hip here; see model_components/occupancy.py for what is and is not pinned.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Literal, Optional

import torch
from torch import Tensor, nn
from torch.nn import Parameter

from . import functional as F
from .cameras.rays import RayBundle
from .field_components.field_heads import FieldHeadNames
from .field_components.spatial_distortions import SceneContraction
from .fields.nerfacto_field import NerfactoField
from .model_components.losses import MSELoss
from .model_components.occupancy import OccGridEstimator
from .model_components.ray_samplers import VolumetricSampler
from .model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer


@dataclass
class InstantNGPModelConfig:
    """models/instant_ngp.py:40-80."""

    grid_resolution: int = 128
    grid_levels: int = 4
    max_res: int = 2048
    log2_hashmap_size: int = 19
    alpha_thre: float = 0.01
    cone_angle: float = 0.004
    render_step_size: Optional[float] = None
    near_plane: float = 0.05
    far_plane: float = 1e3
    use_gradient_scaling: bool = False
    """Scale the field gradients by the squared ray distance (models/instant_ngp.py:72-73, :189-190)."""
    use_appearance_embedding: bool = False
    background_color: Literal["random", "black", "white"] = "random"
    disable_scene_contraction: bool = False
    eval_num_rays_per_chunk: int = 8192
    # Training iterations on the explicit kernel schedule behind this same Model API (ngp_step.NgpFusedStep): static
    # capacity-sized buffers, back-to-back launches, the two sample counts as the only host reads. Not a reference field.
    fused_train_step: bool = False


class _PackedGradientScale(torch.autograd.Function):
    """scale_gradients_by_distance_squared (model_components/losses.py:538-569) for packed samples: identity forward, the
    gradient multiplied by clamp(((start + end) / 2)^2, 0, 1) per sample (elementwise torch ops: `[n]` values)."""

    @staticmethod
    def forward(ctx, value: Tensor, scaling: Tensor):
        ctx.save_for_backward(scaling)
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        (scaling,) = ctx.saved_tensors
        return g * scaling, None


def ngp_outputs(model, ray_bundle: RayBundle, jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """NGPModel.get_outputs (models/instant_ngp.py:172-217) over this package's sampler / field / packed kernels — shared by
    `NGPModel` here and by the plugin's subclass of the reference's NGPModel (plugin.py). `model` needs `config`
    (near_plane, far_plane, render_step_size, alpha_thre, cone_angle[, use_gradient_scaling]), `sampler`, `field` and the
    three renderers."""
    c = model.config
    num_rays = len(ray_bundle)
    with torch.no_grad():
        kw = {} if jitter is None else {"jitter": jitter}
        ray_samples, ray_indices = model.sampler(ray_bundle=ray_bundle, near_plane=c.near_plane, far_plane=c.far_plane,
                                                 render_step_size=c.render_step_size, alpha_thre=c.alpha_thre,
                                                 cone_angle=c.cone_angle, **kw)
    field_outputs = model.field(ray_samples)
    if getattr(c, "use_gradient_scaling", False):
        mid = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
        scaling = torch.square(mid).clamp(0, 1)
        field_outputs = {k: _PackedGradientScale.apply(v, scaling) for k, v in field_outputs.items()}
    # accumulation (models/instant_ngp.py:191-199): nerfacc.pack_info + render_weight_from_density
    packed_info = getattr(ray_indices, "_nsamd_packed_info", None)  # the sampler's own rows (no recount, no host sync)
    if packed_info is None or packed_info.shape[0] != num_rays:
        counts = torch.bincount(ray_indices, minlength=num_rays).to(torch.int32)
        packed_info, _ = F.packed_info_from_counts(counts)
    starts, ends = ray_samples.frustums.starts[..., 0].contiguous(), ray_samples.frustums.ends[..., 0].contiguous()
    weights = F.packed_weights(field_outputs[FieldHeadNames.DENSITY][..., 0], starts, ends, packed_info)[..., None]
    rgb = model.renderer_rgb(rgb=field_outputs[FieldHeadNames.RGB], weights=weights, ray_indices=ray_indices, num_rays=num_rays)
    depth = model.renderer_depth(weights=weights, ray_samples=ray_samples, ray_indices=ray_indices, num_rays=num_rays)
    accumulation = model.renderer_accumulation(weights=weights, ray_indices=ray_indices, num_rays=num_rays)
    return {"rgb": rgb, "accumulation": accumulation, "depth": depth, "num_samples_per_ray": packed_info[:, 1]}


class NGPModel(nn.Module):
    """Instant NGP model (models/instant_ngp.py:83-262). `forward(ray_bundle)` = get_outputs (no collider)."""

    def __init__(self, config: InstantNGPModelConfig, aabb: Tensor, num_train_data: int) -> None:
        super().__init__()
        self.config = config
        self.register_buffer("aabb", aabb.float())
        self.num_train_data = num_train_data
        self.populate_modules()

    def populate_modules(self) -> None:
        c = self.config
        contraction = None if c.disable_scene_contraction else SceneContraction(order=float("inf"))
        self.field = NerfactoField(
            aabb=self.aabb,
            # (sic) models/instant_ngp.py:104: the embedding is 32 wide when use_appearance_embedding is FALSE, as upstream
            appearance_embedding_dim=0 if c.use_appearance_embedding else 32,
            num_images=self.num_train_data, log2_hashmap_size=c.log2_hashmap_size, max_res=c.max_res,
            spatial_distortion=contraction)
        self.scene_aabb = Parameter(self.aabb.flatten(), requires_grad=False)
        if c.render_step_size is None:
            # auto step size: ~1000 samples in the base level grid (models/instant_ngp.py:113-115)
            c.render_step_size = float(((self.scene_aabb[3:] - self.scene_aabb[:3]) ** 2).sum().sqrt().item() / 1000)
        self.occupancy_grid = OccGridEstimator(roi_aabb=self.scene_aabb, resolution=c.grid_resolution, levels=c.grid_levels)
        self.sampler = VolumetricSampler(occupancy_grid=self.occupancy_grid, density_fn=self.field.density_fn)
        self.renderer_rgb = RGBRenderer(background_color=c.background_color)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer(method="expected")
        self.rgb_loss = MSELoss()

    def update_occupancy_grid(self, step: int) -> None:
        """BEFORE_TRAIN_ITERATION callback (models/instant_ngp.py:150-163)."""
        self.occupancy_grid.update_every_n_steps(
            step=step, occ_eval_fn=lambda x: self.field.density_fn(x) * float(self.config.render_step_size))

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        return {"fields": list(self.field.parameters())}

    def forward(self, ray_bundle: RayBundle, jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
        return self.get_outputs(ray_bundle, jitter)

    def _fused_step(self):
        if not (self.training and self.config.fused_train_step and torch.is_grad_enabled()):
            return None
        if getattr(self.config, "use_gradient_scaling", False):
            raise NotImplementedError("fused_train_step: use_gradient_scaling is only on the module path")
        from .fused_step import ddp_reason

        if ddp_reason() is not None:
            raise NotImplementedError(f"fused_train_step: {ddp_reason()} is only on the module path")
        if getattr(self, "_fused", None) is None:
            from .ngp_step import NgpFusedStep

            self._fused = NgpFusedStep(self)
        return self._fused

    def get_outputs(self, ray_bundle: RayBundle, jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
        fused = self._fused_step()
        if fused is not None:
            return fused.get_outputs(ray_bundle, jitter)
        return ngp_outputs(self, ray_bundle, jitter)

    def get_metrics_dict(self, outputs, batch) -> Dict[str, Tensor]:
        image = self.renderer_rgb.blend_background(batch["image"].to(outputs["rgb"].device))
        mse = torch.mean((outputs["rgb"].detach() - image) ** 2)
        return {"psnr": -10.0 * torch.log10(mse), "num_samples_per_batch": outputs["num_samples_per_ray"].sum()}

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, Tensor]:
        if "ngp_step" in outputs:
            return outputs["ngp_step"].get_loss_dict(outputs, batch)
        image = batch["image"].to(outputs["rgb"].device)
        pred_rgb, image = self.renderer_rgb.blend_background_for_loss_computation(
            pred_image=outputs["rgb"], pred_accumulation=outputs["accumulation"], gt_image=image)
        return {"rgb_loss": self.rgb_loss(image, pred_rgb)}

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """Chunked full-image render (models/base_model.py:178-205)."""
        image_shape = camera_ray_bundle.origins.shape[:-1]
        num_rays = len(camera_ray_bundle)
        outs: Dict[str, List[Tensor]] = {}
        for i in range(0, num_rays, self.config.eval_num_rays_per_chunk):
            rb = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + self.config.eval_num_rays_per_chunk)
            for k, v in self.forward(rb).items():
                if torch.is_tensor(v):
                    outs.setdefault(k, []).append(v)
        return {k: torch.cat(v).view(*image_shape, -1) for k, v in outs.items()}


class DynamicBatch:
    """The ray-count feedback of DynamicBatchPipeline (pipelines/dynamic_batch.py:30-95): the occupancy-grid sampler places
    a data-dependent number of samples per ray, so the pipeline rescales the NEXT batch's ray count by
    target_num_samples / (samples the last batch produced) to keep the work per step constant. Host logic; the caller
    (a datamanager's pixel sampler) draws `num_rays_per_batch` rays for the next step."""

    def __init__(self, target_num_samples: int = 1 << 18, max_num_samples_per_ray: int = 1 << 10) -> None:
        self.target_num_samples = int(target_num_samples)
        self.max_num_samples_per_ray = int(max_num_samples_per_ray)
        self.num_rays_per_batch = self.target_num_samples // self.max_num_samples_per_ray  # dynamic_batch.py:62

    def update(self, metrics_dict: Dict[str, Tensor]) -> int:
        """After a training step (dynamic_batch.py:71-95): metrics_dict = NGPModel.get_metrics_dict(...)."""
        if "num_samples_per_batch" not in metrics_dict:
            raise ValueError("'num_samples_per_batch' is not in metrics_dict."
                             "Please return 'num_samples_per_batch' in the models get_metrics_dict function to use this method.")
        num_samples_per_batch = int(metrics_dict["num_samples_per_batch"])
        self.num_rays_per_batch = int(self.num_rays_per_batch * (self.target_num_samples / num_samples_per_batch))
        return self.num_rays_per_batch

