"""MLP containers (reference: nerfstudio/field_components/mlp.py — MLP :61-184, MLPWithHashEncoding :187-295).

The parameters are plain `nn.Linear` weights/biases under the reference's names (`layers.{i}.weight`), so
`state_dict`s interchange with the torch path. The arithmetic of the nerfacto shapes runs in the fused field kernels
(csrc/density_mlp.hip for the proposal heads, csrc/field_mlp.hip on MFMA for the main field), which read these
tensors in place; a stand-alone `MLP.forward` of any shape (any width, skip connections) runs layer by layer on
csrc/linear.hip.
"""
from typing import Literal, Optional, Set, Tuple

import torch
from torch import Tensor, nn

from .base_field_component import FieldComponent, check_implementation
from .encodings import HashEncoding


class MLP(FieldComponent):
    def __init__(
        self,
        in_dim: int,
        num_layers: int,
        layer_width: int,
        out_dim: Optional[int] = None,
        skip_connections: Optional[Tuple[int]] = None,
        activation: Optional[nn.Module] = nn.ReLU(),
        out_activation: Optional[nn.Module] = None,
        implementation: Literal["hip"] = "hip",
    ) -> None:
        super().__init__()
        check_implementation(implementation, "MLP")
        self.in_dim = in_dim
        assert self.in_dim > 0
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.num_layers = num_layers
        self.layer_width = layer_width
        self.skip_connections = skip_connections
        self._skip_connections: Set[int] = set(skip_connections) if skip_connections else set()
        if activation is not None and not isinstance(activation, nn.ReLU):
            raise ValueError("nerfstudio_amd MLPs use ReLU hidden activations (the nerfacto / vanilla-nerf configuration)")
        if out_activation is not None and not isinstance(out_activation, (nn.Sigmoid, nn.ReLU)):
            raise ValueError("nerfstudio_amd MLPs support out_activation None, ReLU or Sigmoid")
        self.activation = activation
        self.out_activation = out_activation
        self.build_nn_modules()

    def build_nn_modules(self) -> None:
        layers = []
        if self.num_layers == 1:
            layers.append(nn.Linear(self.in_dim, self.out_dim))
        else:
            for i in range(self.num_layers - 1):
                if i == 0:
                    assert i not in self._skip_connections, "Skip connection at layer 0 doesn't make sense."
                    layers.append(nn.Linear(self.in_dim, self.layer_width))
                elif i in self._skip_connections:  # mlp.py:151-152: the layer sees [input, hidden]
                    layers.append(nn.Linear(self.layer_width + self.in_dim, self.layer_width))
                else:
                    layers.append(nn.Linear(self.layer_width, self.layer_width))
            layers.append(nn.Linear(self.layer_width, self.out_dim))
        self.layers = nn.ModuleList(layers)

    def param_tensors(self):
        """[W0, b0, W1, b1, ...] in layer order — what the fused kernels bind."""
        out = []
        for layer in self.layers:
            out += [layer.weight, layer.bias]
        return out

    def forward(self, in_tensor: Tensor) -> Tensor:
        """`[*bs, in_dim] -> [*bs, out_dim]` (mlp.py:160-179): one MFMA dense-layer kernel per 128 x 128 block of a
        layer (csrc/linear.hip), ReLU between layers, optional ReLU / Sigmoid at the end, `cat([input, hidden])` in front of
        the skip layers (:171-172; the concatenation is a copy, no arithmetic)."""
        from .. import functional as F

        x = in_tensor
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            if i in self._skip_connections:
                x = torch.cat([in_tensor, x], -1)
            if i < last:
                act = "relu" if self.activation is not None else None
            elif self.out_activation is None:
                act = None
            else:
                act = "sigmoid" if isinstance(self.out_activation, nn.Sigmoid) else "relu"
            x = F.linear(x, layer.weight, layer.bias, act)
        return x


class MLPWithHashEncoding(FieldComponent):
    """Hash encoding + MLP (mlp.py:187-295); `model = Sequential(HashEncoding, MLP)` keeps the reference's
    state-dict names `model.0.hash_table`, `model.1.layers.{i}.*`."""

    def __init__(
        self,
        num_levels: int = 16,
        min_res: int = 16,
        max_res: int = 1024,
        log2_hashmap_size: int = 19,
        features_per_level: int = 2,
        hash_init_scale: float = 0.001,
        interpolation: Optional[Literal["Nearest", "Linear", "Smoothstep"]] = None,
        num_layers: int = 2,
        layer_width: int = 64,
        out_dim: Optional[int] = None,
        skip_connections: Optional[Tuple[int]] = None,
        activation: Optional[nn.Module] = nn.ReLU(),
        out_activation: Optional[nn.Module] = None,
        implementation: Literal["hip"] = "hip",
    ) -> None:
        super().__init__()
        check_implementation(implementation, "MLPWithHashEncoding")
        self.in_dim = 3
        self.out_dim = out_dim if out_dim is not None else layer_width
        encoder = HashEncoding(num_levels, min_res, max_res, log2_hashmap_size, features_per_level, hash_init_scale,
                               implementation, interpolation)
        mlp = MLP(encoder.get_out_dim(), num_layers, layer_width, self.out_dim, skip_connections, activation,
                  out_activation, implementation)
        self.model = torch.nn.Sequential(encoder, mlp)

    @property
    def encoding(self) -> HashEncoding:
        return self.model[0]

    @property
    def mlp(self) -> MLP:
        return self.model[1]

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.model[1](self.model[0](in_tensor))
