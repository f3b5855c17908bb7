"""Field outputs (reference: nerfstudio/field_components/field_heads.py — FieldHeadNames :27-41, FieldHead :44-93,
DensityFieldHead :96-108, RGBFieldHead :111-123, PredNormalsFieldHead :190-206). A head is one dense layer with its
activation; it runs as a single csrc/linear.hip launch (activation fused; tanh follows the launch as a torch op)."""
from enum import Enum
from typing import Optional

import torch
from torch import Tensor, nn

from .base_field_component import FieldComponent


class FieldHeadNames(Enum):
    RGB = "rgb"
    SH = "sh"
    DENSITY = "density"
    NORMALS = "normals"
    PRED_NORMALS = "pred_normals"
    UNCERTAINTY = "uncertainty"
    BACKGROUND_RGB = "background_rgb"
    TRANSIENT_RGB = "transient_rgb"
    TRANSIENT_DENSITY = "transient_density"
    SEMANTICS = "semantics"
    SDF = "sdf"
    ALPHA = "alpha"
    GRADIENT = "gradient"

    # A field of this package can sit inside the REFERENCE's model classes (plugin.HipNerfactoModel inherits the
    # reference's get_outputs, which indexes `field_outputs[FieldHeadNames.DENSITY]` with the reference's own enum class,
    # models/nerfacto.py:304-324): members compare and hash equal to the same-named member of any `FieldHeadNames` enum,
    # so the dictionaries interchange in both directions without importing nerfstudio here.
    def __eq__(self, other):
        if self is other:
            return True
        return isinstance(other, Enum) and type(other).__name__ == "FieldHeadNames" and other.name == self.name \
            and other.value == self.value

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._name_)  # (= Enum's own hash: equal for the reference's member of the same name)


_HEAD_ACTIVATIONS = {type(None): None, nn.ReLU: "relu", nn.Sigmoid: "sigmoid", nn.Softplus: "softplus", nn.Tanh: "tanh"}


class FieldHead(FieldComponent):
    """Base field output: `net = nn.Linear(in_dim, out_dim)` (state-dict name `net.*`) + activation."""

    def __init__(self, out_dim: int, field_head_name: FieldHeadNames, in_dim: Optional[int] = None,
                 activation: Optional[nn.Module] = None) -> None:
        super().__init__()
        if type(activation) not in _HEAD_ACTIVATIONS:
            raise ValueError(f"field head activation {activation!r}: the hip heads take None / ReLU / Sigmoid / Softplus / Tanh")
        if isinstance(activation, nn.Softplus) and (activation.beta != 1 or activation.threshold != 20):
            raise ValueError("the fused Softplus is torch's default (beta = 1, threshold = 20)")
        self.out_dim = out_dim
        self.activation = activation
        self.field_head_name = field_head_name
        self.net = None
        if in_dim is not None:
            self.in_dim = in_dim
            self._construct_net()

    def set_in_dim(self, in_dim: int) -> None:
        self.in_dim = in_dim
        self._construct_net()

    def _construct_net(self) -> None:
        self.net = nn.Linear(self.in_dim, self.out_dim)

    def forward(self, in_tensor: Tensor) -> Tensor:
        from .. import functional as F

        if not self.net:
            raise SystemError("in_dim not set. Must be provided to constructor, or set_in_dim() should be called.")
        act = _HEAD_ACTIVATIONS[type(self.activation)]
        if act == "tanh":
            return torch.tanh(F.linear(in_tensor, self.net.weight, self.net.bias, None))
        return F.linear(in_tensor, self.net.weight, self.net.bias, act)


class DensityFieldHead(FieldHead):
    def __init__(self, in_dim: Optional[int] = None, activation: Optional[nn.Module] = nn.Softplus()) -> None:
        super().__init__(in_dim=in_dim, out_dim=1, field_head_name=FieldHeadNames.DENSITY, activation=activation)


class RGBFieldHead(FieldHead):
    def __init__(self, in_dim: Optional[int] = None, activation: Optional[nn.Module] = nn.Sigmoid()) -> None:
        super().__init__(in_dim=in_dim, out_dim=3, field_head_name=FieldHeadNames.RGB, activation=activation)


class PredNormalsFieldHead(FieldHead):
    """Predicted normals (field_heads.py:190-206): tanh head, then normalised to unit length."""

    def __init__(self, in_dim: Optional[int] = None, activation: Optional[nn.Module] = nn.Tanh()) -> None:
        super().__init__(in_dim=in_dim, out_dim=3, field_head_name=FieldHeadNames.PRED_NORMALS, activation=activation)

    def forward(self, in_tensor: Tensor) -> Tensor:
        return torch.nn.functional.normalize(super().forward(in_tensor), dim=-1)
