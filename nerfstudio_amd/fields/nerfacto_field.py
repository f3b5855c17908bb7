"""nerfacto main field (reference: nerfstudio/fields/nerfacto_field.py:42-310)."""
from typing import Dict, Literal, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _native as N
from .. import functional as F
from ..cameras.rays import RaySamples
from ..field_components.base_field_component import check_implementation
from ..field_components.embedding import Embedding
from ..field_components.activations import trunc_exp
from ..field_components.encodings import NeRFEncoding, SHEncoding
from ..field_components.field_heads import FieldHeadNames, PredNormalsFieldHead
from ..field_components.mlp import MLP, MLPWithHashEncoding
from ..field_components.spatial_distortions import SpatialDistortion
from .base_field import Field, get_normalized_directions, point_spec
from .density_fields import transform_of


class NerfactoField(Field):
    """Compound field: hash grid + base MLP (density, geo features) and SH + appearance embedding + colour MLP.

    Arguments as the reference (nerfacto_field.py:72-98). The hip backend builds the nerfacto shape
    (hidden_dim = hidden_dim_color = 64, geo_feat_dim = 15, num_levels*features = 32, num_layers 2 / 3,
    appearance_embedding_dim 32 or 0); the transient / semantic heads belong to other methods.

    `get_density` + `get_outputs` are ONE pipeline on the GPU (hash encode -> MFMA MLP chain), so `get_density`
    evaluates both and hands rgb to `get_outputs` through the density embedding slot.

    Normals (`use_pred_normals`, `forward(..., compute_normals=True)`; nerfacto_field.py:181-191, 215-223, 287-295,
    base_field.py:79-99): the analytic normals need the gradient of the density pre-activation with respect to the
    normalised sample positions and the predicted ones the geometry features — neither leaves the fused pipeline, so
    these two options evaluate the field as the reference composes it: position normalisation (torch, differentiable) ->
    hash-encode kernel (its backward returns dL/dx) -> dense-layer kernels (csrc/linear.hip) -> trunc_exp; SH kernel +
    embedding lookup -> colour MLP; frequency encoding + geometry features -> normals MLP -> tanh head. Same parameters,
    same names, same arithmetic up to fp32 rounding; not the benchmarked path.
    """

    aabb: Tensor

    def __init__(
        self,
        aabb: Tensor,
        num_images: int,
        num_layers: int = 2,
        hidden_dim: int = 64,
        geo_feat_dim: int = 15,
        num_levels: int = 16,
        base_res: int = 16,
        max_res: int = 2048,
        log2_hashmap_size: int = 19,
        num_layers_color: int = 3,
        num_layers_transient: int = 2,
        features_per_level: int = 2,
        hidden_dim_color: int = 64,
        hidden_dim_transient: int = 64,
        appearance_embedding_dim: int = 32,
        transient_embedding_dim: int = 16,
        use_transient_embedding: bool = False,
        use_semantics: bool = False,
        num_semantic_classes: int = 100,
        pass_semantic_gradients: bool = False,
        use_pred_normals: bool = False,
        use_average_appearance_embedding: bool = False,
        spatial_distortion: Optional[SpatialDistortion] = None,
        average_init_density: float = 1.0,
        implementation: Literal["hip"] = "hip",
    ) -> None:
        super().__init__()
        check_implementation(implementation, "NerfactoField")
        if use_transient_embedding or use_semantics:
            raise ValueError("transient / semantic heads are not part of the hip nerfacto field")
        if (hidden_dim, hidden_dim_color, geo_feat_dim, num_layers, num_layers_color) != (64, 64, 15, 2, 3) or \
                num_levels * features_per_level != 32 or appearance_embedding_dim not in (0, 32):
            raise ValueError(
                "the hip NerfactoField is built for hidden_dim=64, hidden_dim_color=64, geo_feat_dim=15, num_layers=2, "
                "num_layers_color=3, num_levels*features_per_level=32, appearance_embedding_dim in {0, 32}")
        self.register_buffer("aabb", aabb)
        self.geo_feat_dim = geo_feat_dim
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion
        self.num_images = num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.embedding_appearance = Embedding(num_images, appearance_embedding_dim) if appearance_embedding_dim > 0 else None
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.use_transient_embedding = use_transient_embedding
        self.use_semantics = use_semantics
        self.use_pred_normals = use_pred_normals
        self.base_res = base_res
        self.average_init_density = average_init_density
        self.step = 0
        self.direction_encoding = SHEncoding(levels=4, implementation=implementation)
        self.position_encoding = NeRFEncoding(in_dim=3, num_frequencies=2, min_freq_exp=0, max_freq_exp=2 - 1,
                                              implementation=implementation)
        self.mlp_base = MLPWithHashEncoding(
            num_levels=num_levels,
            min_res=base_res,
            max_res=max_res,
            log2_hashmap_size=log2_hashmap_size,
            features_per_level=features_per_level,
            num_layers=num_layers,
            layer_width=hidden_dim,
            out_dim=1 + self.geo_feat_dim,
            activation=nn.ReLU(),
            out_activation=None,
            implementation=implementation,
        )
        if self.use_pred_normals:  # nerfacto_field.py:181-191
            self.mlp_pred_normals = MLP(
                in_dim=self.geo_feat_dim + self.position_encoding.get_out_dim(),
                num_layers=3,
                layer_width=64,
                out_dim=hidden_dim_transient,
                activation=nn.ReLU(),
                out_activation=None,
                implementation=implementation,
            )
            self.field_head_pred_normals = PredNormalsFieldHead(in_dim=self.mlp_pred_normals.get_out_dim())
        self.mlp_head = MLP(
            in_dim=self.direction_encoding.get_out_dim() + self.geo_feat_dim + self.appearance_embedding_dim,
            num_layers=num_layers_color,
            layer_width=hidden_dim_color,
            out_dim=3,
            activation=nn.ReLU(),
            out_activation=nn.Sigmoid(),
            implementation=implementation,
        )
        self._transform = transform_of(spatial_distortion)
        self._box = N.make_aabb(aabb)  # host copy of the scene box: no device sync on the hot path

    # -----------------------------------------------------------------------------------------------------------
    def _evaluate(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        spec, shape = point_spec(ray_samples)
        fr = ray_samples.frustums
        if spec.ray_mode:
            view_dirs = spec.directions  # ray mode: the pack's per-ray directions
            dir_group = spec.samples_per_ray
            cams = ray_samples.camera_indices.reshape(view_dirs.shape[0], -1)[:, 0]
        else:
            view_dirs = fr.directions.expand(*shape, 3).reshape(-1, 3)
            dir_group = 1
            cams = ray_samples.camera_indices.expand(*shape, 1).reshape(-1)
        emb = self.embedding_appearance.embedding.weight if self.embedding_appearance is not None else None
        app_const = None
        if emb is not None and not self.training:
            # nerfacto_field.py:253-261: eval uses the mean embedding (or zeros) for every sample
            cams = None
            with torch.no_grad():
                app_const = emb.mean(dim=0) if self.use_average_appearance_embedding else torch.zeros_like(emb[0])
            app_const = app_const.contiguous()
        if emb is None:
            cams = None
        enc = self.mlp_base.encoding
        density, rgb = F.nerfacto_field(
            spec, enc.hash_table, self.mlp_base.mlp.param_tensors(), self.mlp_head.param_tensors(), emb, view_dirs, cams,
            app_const, dir_group, enc.spec, self._transform, self._box, self.average_init_density)
        return density.view(*shape, 1), rgb.view(*shape, 3)

    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None) -> Tensor:
        """Density only, on explicit positions `[*bs,3]` -> `[*bs,1]` (base_field.py:48-68) — what the occupancy grid
        and the VolumetricSampler's sigma_fn call (models/instant_ngp.py:133,153; ray_samplers.py:420-429). No camera
        and no view direction are involved: the fused kernel runs with a constant appearance row and a fixed direction,
        its colour output is dropped."""
        del times
        shape = positions.shape[:-1]
        pos = positions.reshape(-1, 3)
        emb = self.embedding_appearance.embedding.weight if self.embedding_appearance is not None else None
        app_const = torch.zeros(emb.shape[1], device=pos.device) if emb is not None else None
        view = torch.zeros((1, 3), device=pos.device)
        enc = self.mlp_base.encoding
        density, _ = F.nerfacto_field(F.PointSpec(positions=pos), enc.hash_table, self.mlp_base.mlp.param_tensors(),
                                      self.mlp_head.param_tensors(), emb, view, None, app_const, max(pos.shape[0], 1),
                                      enc.spec, self._transform, self._box, self.average_init_density)
        return density.view(*shape, 1)

    # ---- the field as the reference composes it (normals) ----------------------------------------------------------
    def _composed(self) -> bool:
        return self.use_pred_normals or self._compute_normals

    def _normalised_positions(self, raw: Tensor) -> Tuple[Tensor, Tensor]:
        """nerfacto_field.py:205-214 in torch ops (differentiable with respect to the raw positions)."""
        if self.spatial_distortion is not None:
            mag = torch.linalg.norm(raw, ord=float("inf"), dim=-1)[..., None]  # spatial_distortions.py:66-69
            pos = (torch.where(mag < 1, raw, (2 - (1 / mag)) * (raw / mag)) + 2.0) / 4.0
        else:
            pos = (raw - self.aabb[0]) / (self.aabb[1] - self.aabb[0])  # data/scene_box.py:62-71
        selector = ((pos > 0.0) & (pos < 1.0)).all(dim=-1)
        return pos * selector[..., None], selector

    def _density_composed(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        positions, selector = self._normalised_positions(ray_samples.frustums.get_positions())
        self._sample_locations = positions
        if not self._sample_locations.requires_grad:
            self._sample_locations.requires_grad = True
        h = self.mlp_base(positions.view(-1, 3)).view(*ray_samples.frustums.shape, -1)
        density_before_activation, base_mlp_out = torch.split(h, [1, self.geo_feat_dim], dim=-1)
        self._density_before_activation = density_before_activation
        density = self.average_init_density * trunc_exp(density_before_activation)
        return density * selector[..., None], base_mlp_out

    def _outputs_composed(self, ray_samples: RaySamples, density_embedding: Tensor) -> Dict[FieldHeadNames, Tensor]:
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        fr = ray_samples.frustums
        shape = tuple(fr.shape)
        outputs = {}
        directions = get_normalized_directions(fr.directions.expand(*shape, 3))
        d = self.direction_encoding(directions.reshape(-1, 3))
        feats = [d, density_embedding.reshape(-1, self.geo_feat_dim)]
        if self.embedding_appearance is not None:
            if self.training:
                cams = ray_samples.camera_indices.expand(*shape, 1).reshape(-1)
                feats.append(self.embedding_appearance(cams))
            else:  # nerfacto_field.py:253-261
                emb = self.embedding_appearance.embedding.weight
                row = emb.mean(dim=0) if self.use_average_appearance_embedding else torch.zeros_like(emb[0])
                feats.append(row.detach().expand(d.shape[0], -1))
        if self.use_pred_normals:  # nerfacto_field.py:287-295: encoded RAW positions + geometry features
            positions_flat = self.position_encoding(fr.get_positions().reshape(-1, 3))
            x = self.mlp_pred_normals(torch.cat([positions_flat, feats[1]], dim=-1)).view(*shape, -1)
            outputs[FieldHeadNames.PRED_NORMALS] = self.field_head_pred_normals(x)
        rgb = self.mlp_head(torch.cat(feats, dim=-1)).view(*shape, -1)
        outputs[FieldHeadNames.RGB] = rgb
        return outputs

    # -----------------------------------------------------------------------------------------------------------------
    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        """Densities `[*bs,1]`; the second value carries the rgb already evaluated by the fused pipeline (or, on the
        composed path of the normals options, the geometry features as in the reference)."""
        self._composed_active = self._composed()
        if self._composed_active:
            return self._density_composed(ray_samples)
        density, rgb = self._evaluate(ray_samples)
        return density, rgb

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None
                    ) -> Dict[FieldHeadNames, Tensor]:
        assert density_embedding is not None
        if getattr(self, "_composed_active", False):
            return self._outputs_composed(ray_samples, density_embedding)
        return {FieldHeadNames.RGB: density_embedding}
