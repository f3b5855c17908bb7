"""Exponential maps for learned camera-pose corrections (reference: nerfstudio/cameras/lie_groups.py:25-117).

Host-side torch (the camera optimiser stays in torch; what the kernels owe it is dL/d(origins,
directions) per ray — nsamd_hashgrid_encode_bwd_rays). Written from the closed forms, same branch points and clamps as
the reference so that pose gradients agree to rounding:
  SO3xR3:  R = I + (sin t / t) K + ((1 - cos t) / t^2) K^2,   t = sqrt(max(|w|^2, 1e-4)),  translation = v
  SE3:     R as Rodrigues with Taylor branches for t < 1e-2,  translation = V(w) v,
           V = (sin t / t) I + ((1 - cos t) / t^2) K + ((t - sin t) / t^3) w w^T
"""
import torch
from torch import Tensor


def _hat(w: Tensor) -> Tensor:
    """[b,3] -> skew-symmetric [b,3,3] with hat(w) x = w cross x."""
    zero = torch.zeros_like(w[:, 0])
    return torch.stack([torch.stack([zero, -w[:, 2], w[:, 1]], -1), torch.stack([w[:, 2], zero, -w[:, 0]], -1),
                        torch.stack([-w[:, 1], w[:, 0], zero], -1)], -2)


def exp_map_SO3xR3(tangent_vector: Tensor) -> Tensor:
    """`[b,6]` (translation, so(3) vector) -> `[b,3,4]` [R|t] of the direct product SO(3) x R^3 (lie_groups.py:25-60)."""
    v, w = tangent_vector[:, :3], tangent_vector[:, 3:]
    theta = torch.clamp((w * w).sum(1), 1e-4).sqrt()
    inv = 1.0 / theta
    a = inv * theta.sin()
    b = inv * inv * (1.0 - theta.cos())
    K = _hat(w)
    eye = torch.eye(3, dtype=w.dtype, device=w.device)[None]
    R = a[:, None, None] * K + b[:, None, None] * torch.bmm(K, K) + eye
    return torch.cat([R, v[:, :, None]], dim=-1)


def exp_map_SE3(tangent_vector: Tensor) -> Tensor:
    """`[b,6]` se(3) tangent (linear, angular) -> `[b,3,4]` (lie_groups.py:63-117), Taylor branches below theta = 1e-2."""
    v, w = tangent_vector[:, :3, None], tangent_vector[:, 3:, None]  # [b,3,1]
    theta = torch.linalg.norm(w, dim=1).unsqueeze(1)  # [b,1,1]
    t2, t3 = theta**2, theta**3
    small = theta < 1e-2
    one = torch.ones(1, dtype=tangent_vector.dtype, device=tangent_vector.device)
    theta_s, t2_s, t3_s = torch.where(small, one, theta), torch.where(small, one, t2), torch.where(small, one, t3)
    sin = theta.sin()
    cos = torch.where(small, 8 / (4 + t2) - 1, theta.cos())
    a = torch.where(small, 0.5 * cos + 0.5, sin / theta_s)             # sin t / t
    b = torch.where(small, 0.5 * a, (1 - cos) / t2_s)                  # (1 - cos t) / t^2
    R = b * w @ w.transpose(1, 2) + cos * torch.eye(3, dtype=w.dtype, device=w.device)[None] + _hat((a.view(-1, 1) * w.view(-1, 3)))
    a_t = torch.where(small, 1 - t2 / 6, a)
    b_t = torch.where(small, 0.5 - t2 / 24, b)
    c_t = torch.where(small, 1.0 / 6 - t2 / 120, (theta - sin) / t3_s)  # (t - sin t) / t^3
    t = a_t * v + b_t * torch.cross(w, v, dim=1) + c_t * (w @ (w.transpose(1, 2) @ v))
    return torch.cat([R, t], dim=-1)
