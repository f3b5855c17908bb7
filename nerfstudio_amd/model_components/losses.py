"""Losses on the nerfacto path (reference: nerfstudio/model_components/losses.py — MSELoss :31, interlevel_loss
:113-131, distortion_loss :149-154, orientation_loss :201-214, pred_normal_loss :217-222). The proposal losses are per-ray
fused value+gradient kernels (csrc/losses.hip); the two normals terms (predict_normals, off by default) are a handful of
elementwise torch ops on tensors the field already produced."""
from typing import List

import torch
from torch import Tensor, nn

from .. import functional as F
from ..cameras.rays import RaySamples, pack_of

MSELoss = nn.MSELoss
EPS = 1.0e-7


def ray_samples_to_sdist(ray_samples: RaySamples) -> Tensor:
    """Spacing-domain bin edges `[num_rays, S+1]` (losses.py:105-110)."""
    pk = pack_of(ray_samples)
    if pk is not None and pk.s_bins is not None:
        return pk.s_bins
    starts, ends = ray_samples.spacing_starts, ray_samples.spacing_ends
    return torch.cat([starts[..., 0], ends[..., -1:, 0]], dim=-1)


def interlevel_loss(weights_list: List[Tensor], ray_samples_list: List[RaySamples]) -> Tensor:
    """Proposal loss of mip-NeRF 360 (losses.py:113-131). weights `[N,S_i,1]`."""
    bins = [ray_samples_to_sdist(rs) for rs in ray_samples_list]
    ws = [w[..., 0] for w in weights_list]
    return F.interlevel_loss(ws, bins)


def distortion_loss(weights_list: List[Tensor], ray_samples_list: List[RaySamples]) -> Tensor:
    """Distortion loss of mip-NeRF 360 on the final level (losses.py:149-154)."""
    return F.distortion_loss(weights_list[-1][..., 0], ray_samples_to_sdist(ray_samples_list[-1]))


def orientation_loss(weights: Tensor, normals: Tensor, viewdirs: Tensor) -> Tensor:
    """Ref-NeRF orientation loss (losses.py:201-214): a visible normal should not point away from the camera. weights
    `[*bs,S,1]`, normals `[*bs,S,3]`, viewdirs `[*bs,3]` -> `[*bs]`: sum_s w_s min(0, n_s . (-d))^2."""
    towards_camera = -(normals * viewdirs[..., None, :]).sum(dim=-1)  # n . (-d); negation is exact
    back_facing = torch.fmin(towards_camera, torch.zeros_like(towards_camera))  # (fmin: a NaN dot product counts as 0)
    return (weights[..., 0] * back_facing**2).sum(dim=-1)


def pred_normal_loss(weights: Tensor, normals: Tensor, pred_normals: Tensor) -> Tensor:
    """Predicted normals against the ones computed from the density (losses.py:217-222): sum_s w_s (1 - n_s . p_s) -> `[*bs]`."""
    agreement = (normals * pred_normals).sum(dim=-1)
    return (weights[..., 0] * (1.0 - agreement)).sum(dim=-1)
