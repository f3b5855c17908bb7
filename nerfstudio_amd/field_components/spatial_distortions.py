"""Space distortions (reference: nerfstudio/field_components/spatial_distortions.py:28-104)."""
from typing import Optional, Union

from torch import Tensor, nn

from .. import functional as F


class SpatialDistortion(nn.Module):
    """Apply spatial distortions"""

    def forward(self, positions: Tensor) -> Tensor:
        raise NotImplementedError


class SceneContraction(SpatialDistortion):
    """MipNeRF-360 contraction, x -> (2 - 1/|x|)(x/|x|) for |x| > 1 (spatial_distortions.py:42-104).

    Only the L-inf norm (the one every hash-grid method uses, models/nerfacto.py:151) is implemented on the GPU.
    Inside the fused fields the contraction and its Jacobian run in the hash-encode kernels; calling the module
    directly runs the forward-only kernel."""

    def __init__(self, order: Optional[Union[float, int]] = None) -> None:
        super().__init__()
        if order != float("inf"):
            raise ValueError("nerfstudio_amd implements SceneContraction(order=float('inf')) only "
                             "(the hash-grid configuration); got order=%r" % (order,))
        self.order = order

    def forward(self, positions: Tensor) -> Tensor:
        return F.contract_linf(positions)
