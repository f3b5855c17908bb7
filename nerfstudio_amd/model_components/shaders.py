"""Shaders (reference: nerfstudio/model_components/shaders.py — NormalsShader :57-78)."""
from typing import Optional

from torch import Tensor, nn


class NormalsShader(nn.Module):
    """Unit normals as colours: every component moved from [-1, 1] to [0, 1]; optional per-pixel weights (an accumulation
    mask) scale the result (shaders.py:60-78)."""

    @classmethod
    def forward(cls, normals: Tensor, weights: Optional[Tensor] = None) -> Tensor:
        colours = 0.5 * normals + 0.5  # (= (n + 1) / 2 bit for bit: halving is exact)
        return colours if weights is None else colours * weights
