"""FieldComponent base (reference: nerfstudio/field_components/base_field_component.py:26-70)."""
from abc import abstractmethod
from typing import Optional

from torch import Tensor, nn


class FieldComponent(nn.Module):
    """Field modules that can be combined to store and compute the fields."""

    def __init__(self, in_dim: Optional[int] = None, out_dim: Optional[int] = None) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim

    def build_nn_modules(self) -> None:
        """Function instantiates any torch.nn members within the module."""

    def set_in_dim(self, in_dim: int) -> None:
        if in_dim <= 0:
            raise ValueError("Input dimension should be greater than zero")
        self.in_dim = in_dim

    def get_out_dim(self) -> int:
        if self.out_dim is None:
            raise ValueError("Output dimension has not been set")
        return self.out_dim

    @abstractmethod
    def forward(self, in_tensor: Tensor) -> Tensor:
        raise NotImplementedError


def check_implementation(implementation: str, who: str) -> None:
    """The reference's backend switch is `implementation in {"tcnn","torch"}` (models/nerfacto.py:125); this package
    is the third backend. Asking it for another one is an error, never a silent fallback."""
    if implementation != "hip":
        raise ValueError(
            f"{who}: nerfstudio_amd provides implementation='hip' only (got {implementation!r}); "
            "use nerfstudio's own modules for 'torch' / 'tcnn'."
        )
