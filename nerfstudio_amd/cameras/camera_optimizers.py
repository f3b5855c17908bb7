"""Pose optimisation of the training cameras (reference: nerfstudio/cameras/camera_optimizers.py:41-245).

nerfacto's default is `CameraOptimizerConfig(mode="SO3xR3")` (configs/method_configs.py:102): every training ray's origin
and direction go through a learned per-camera correction, so dL/d(origins, directions) must come back out of the field.
That gradient is produced on the device (nsamd_hashgrid_encode_bwd_rays: dL/dposition through the hash encoding, the
selector, the affine map and the contraction Jacobian, reduced per ray); the `[num_cameras, 6]` parameter, the
exponential map and the regulariser are host-side torch, as in the reference.
"""
import os
from dataclasses import dataclass
from typing import Literal, Optional, Union

import numpy
import torch
from torch import Tensor, nn

from .lie_groups import exp_map_SE3, exp_map_SO3xR3


@dataclass
class CameraOptimizerConfig:
    """camera_optimizers.py:41-82 (the deprecated optimizer / scheduler fields are not carried)."""

    mode: Literal["off", "SO3xR3", "SE3"] = "off"
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3

    def setup(self, num_cameras: int, device: Union[torch.device, str], **kwargs) -> "CameraOptimizer":
        return CameraOptimizer(self, num_cameras=num_cameras, device=device, **kwargs)


class CameraOptimizer(nn.Module):
    """Layer that modifies camera poses to be optimised together with the field (camera_optimizers.py:85-245)."""

    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device: Union[torch.device, str],
                 non_trainable_camera_indices: Optional[Tensor] = None, **kwargs) -> None:
        super().__init__()
        if config.mode not in ("off", "SO3xR3", "SE3"):
            raise ValueError(f"unknown camera optimizer mode {config.mode!r}")
        self.config = config
        self.num_cameras = num_cameras
        self._init_device = device  # where the parameter is created; `device` below follows `.to()`
        self.non_trainable_camera_indices = non_trainable_camera_indices
        if config.mode != "off":
            self.pose_adjustment = torch.nn.Parameter(torch.zeros((num_cameras, 6), device=device))

    @property
    def device(self):
        """The device the corrections live on: the parameter's (a stored constructor argument goes stale after `.to()`)."""
        p = getattr(self, "pose_adjustment", None)
        return p.device if p is not None else torch.device(self._init_device)

    def forward(self, indices: Tensor) -> Tensor:
        """`[n]` camera indices -> `[n,3,4]` corrections (optimised camera -> given camera coordinates)."""
        if self.config.mode == "off":
            return torch.eye(4, device=indices.device)[None, :3, :4].tile(indices.shape[0], 1, 1)
        exp_map = exp_map_SO3xR3 if self.config.mode == "SO3xR3" else exp_map_SE3
        out = exp_map(self.pose_adjustment[indices, :])
        if self.non_trainable_camera_indices is not None:
            fixed = self.non_trainable_camera_indices.to(self.pose_adjustment.device)
            out[fixed] = torch.eye(4, device=self.pose_adjustment.device)[:3, :4]
        return out

    def corrected_rays(self, origins: Tensor, directions: Tensor, camera_indices: Tensor):
        """origins + t, R @ directions for rays `[n,3]` of cameras `[n]` (the arithmetic of apply_to_raybundle)."""
        if self.config.mode == "off":
            return origins, directions
        pose = self.pose_adjustment
        if (pose.is_cuda and pose.dtype == torch.float32 and self.non_trainable_camera_indices is None and origins.is_cuda
                and not origins.requires_grad and not directions.requires_grad and origins.dim() == 2
                and os.environ.get("NSAMD_CAMERA_KERNELS", "1") == "1"):
            # the two camera kernels (functional.camera_correct_rays) — the same launches the explicit schedule makes, so
            # both routes hand identical rays to the field
            from .. import functional as F

            return F.camera_correct_rays(pose, self.config.mode, origins, directions, camera_indices)
        c = self(camera_indices.reshape(-1))
        return origins + c[:, :3, 3], torch.bmm(c[:, :3, :3], directions[..., None]).squeeze(-1)

    def apply_to_raybundle(self, raybundle) -> None:
        """Apply the pose correction to the ray bundle in place (camera_optimizers.py:148-153)."""
        if self.config.mode != "off":
            raybundle.origins, raybundle.directions = self.corrected_rays(raybundle.origins, raybundle.directions,
                                                                          raybundle.camera_indices)

    def get_loss_dict(self, loss_dict: dict) -> None:
        """L2 regulariser on the translation and rotation parts (camera_optimizers.py:179-185)."""
        if self.config.mode != "off":
            loss_dict["camera_opt_regularizer"] = (
                self.pose_adjustment[:, :3].norm(dim=-1).mean() * self.config.trans_l2_penalty
                + self.pose_adjustment[:, 3:].norm(dim=-1).mean() * self.config.rot_l2_penalty)

    def get_correction_matrices(self) -> Tensor:
        return self(torch.arange(0, self.num_cameras).long())

    def get_metrics_dict(self, metrics_dict: dict) -> None:
        if self.config.mode != "off":
            trans = self.pose_adjustment[:, :3].detach().norm(dim=-1)
            rot = self.pose_adjustment[:, 3:].detach().norm(dim=-1)
            metrics_dict["camera_opt_translation_max"] = trans.max()
            metrics_dict["camera_opt_translation_mean"] = trans.mean()
            metrics_dict["camera_opt_rotation_mean"] = numpy.rad2deg(rot.mean().cpu())
            metrics_dict["camera_opt_rotation_max"] = numpy.rad2deg(rot.max().cpu())

    def get_param_groups(self, param_groups: dict) -> None:
        params = list(self.parameters())
        if self.config.mode != "off":
            assert len(params) > 0
            param_groups["camera_opt"] = params
        else:
            assert len(params) == 0
