"""Shaders (reference: nerfstudio/model_components/shaders.py — NormalsShader :57-78)."""
from typing import Optional

from torch import Tensor, nn


class NormalsShader(nn.Module):
    """Normals as colours: (n + 1) / 2, optionally scaled by per-pixel weights (shaders.py:60-78)."""

    @classmethod
    def forward(cls, normals: Tensor, weights: Optional[Tensor] = None) -> Tensor:
        normals = (normals + 1) / 2
        if weights is not None:
            normals = normals * weights
        return normals
