"""Losses on the nerfacto path (reference: nerfstudio/model_components/losses.py — MSELoss :31, interlevel_loss
:113-131, distortion_loss :149-154). Per-ray fused value+gradient kernels (csrc/losses.hip)."""
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
