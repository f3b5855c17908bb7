"""Encodings (reference: nerfstudio/field_components/encodings.py — Encoding :36-55, Identity :58-68, NeRFEncoding
:90-189, HashEncoding :307-463, SHEncoding :752-799)."""
from abc import abstractmethod
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from .. import functional as F
from .base_field_component import FieldComponent, check_implementation


class Encoding(FieldComponent):
    """Encode an input tensor. Intended to be subclassed."""

    def __init__(self, in_dim: int) -> None:
        if in_dim <= 0:
            raise ValueError("Input dimension should be greater than zero")
        super().__init__(in_dim=in_dim)

    @abstractmethod
    def forward(self, in_tensor: Tensor) -> Tensor:
        raise NotImplementedError


class Identity(Encoding):
    """Identity encoding (encodings.py:58-68)."""

    def get_out_dim(self) -> int:
        if self.in_dim is None:
            raise ValueError("Input dimension has not been set")
        return self.in_dim

    def forward(self, in_tensor: Tensor) -> Tensor:
        return in_tensor


class NeRFEncoding(Encoding):
    """Frequency encoding of vanilla NeRF (encodings.py:90-189): `[sin(2 pi x 2^k), sin(... + pi/2), x]`, one kernel
    (csrc/hashgrid.hip nerf_encode_kernel). Integrated (mip-NeRF, `covs`) encodings are not built."""

    def __init__(self, in_dim: int, num_frequencies: int, min_freq_exp: float, max_freq_exp: float,
                 include_input: bool = False, implementation: Literal["hip"] = "hip") -> None:
        super().__init__(in_dim)
        check_implementation(implementation, "NeRFEncoding")
        if in_dim != 3:
            raise ValueError("the hip NeRFEncoding encodes 3-vectors (positions, directions)")
        self.num_frequencies = num_frequencies
        self.min_freq = min_freq_exp
        self.max_freq = max_freq_exp
        self.include_input = include_input

    def get_out_dim(self) -> int:
        out_dim = self.in_dim * self.num_frequencies * 2
        if self.include_input:
            out_dim += self.in_dim
        return out_dim

    def spec_forward(self, spec: "F.PointSpec") -> Tensor:
        """Encoding of the points of a PointSpec (rays + bin edges: the sample midpoints are formed in the kernel)."""
        return F.nerf_encode(spec, self.num_frequencies, self.min_freq, self.max_freq, self.include_input)

    def forward(self, in_tensor: Tensor, covs: Optional[Tensor] = None) -> Tensor:
        if covs is not None:
            raise NotImplementedError("integrated (mip-NeRF) encodings are not built for the hip backend")
        shape = in_tensor.shape[:-1]
        if in_tensor.requires_grad and torch.is_grad_enabled():
            # the kernel has no backward; points that carry gradient (pose corrections of a camera optimiser behind the
            # predicted-normals head) take the same arithmetic through torch (encodings.py:148-166)
            freqs = (2 ** torch.linspace(self.min_freq, self.max_freq, self.num_frequencies)).to(in_tensor.device)
            scaled = ((2 * torch.pi * in_tensor)[..., None] * freqs).reshape(*shape, -1)
            out = torch.sin(torch.cat([scaled, scaled + torch.pi / 2.0], dim=-1))
            return torch.cat([out, in_tensor], dim=-1) if self.include_input else out
        out = self.spec_forward(F.PointSpec(positions=in_tensor.detach().reshape(-1, 3)))
        return out.view(*shape, self.get_out_dim())


class HashEncoding(Encoding):
    """Multiresolution hash encoding with the reference's TORCH-path semantics (encodings.py:417-458), as HIP
    kernels (csrc/hashgrid.hip).

    Args mirror the reference (encodings.py:321-331). `hash_table` is an `nn.Parameter` `[L * 2^log2, F]` with the
    same name, shape and init as the torch path, so checkpoints interchange.
    """

    def __init__(
        self,
        num_levels: int = 16,
        min_res: int = 16,
        max_res: int = 1024,
        log2_hashmap_size: int = 19,
        features_per_level: int = 2,
        hash_init_scale: float = 0.001,
        implementation: Literal["hip"] = "hip",
        interpolation: Optional[Literal["Nearest", "Linear", "Smoothstep"]] = None,
    ) -> None:
        super().__init__(in_dim=3)
        check_implementation(implementation, "HashEncoding")
        assert interpolation is None or interpolation == "Linear", (
            f"interpolation '{interpolation}' is not supported for the hip encoding backend"
        )
        self.num_levels = num_levels
        self.min_res = min_res
        self.max_res = max_res
        self.features_per_level = features_per_level
        self.hash_init_scale = hash_init_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.hash_table_size = 2**log2_hashmap_size
        self.spec = F.HashGridSpec(num_levels, min_res, max_res, log2_hashmap_size, features_per_level)
        self.growth_factor = self.spec.growth_factor
        self.scalings = self.spec.scalings()
        self.hash_offset = torch.arange(num_levels) * self.hash_table_size
        self.build_nn_modules()

    def build_nn_modules(self) -> None:
        table = torch.rand(size=(self.hash_table_size * self.num_levels, self.features_per_level)) * 2 - 1
        table *= self.hash_init_scale
        self.hash_table = nn.Parameter(table)

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def forward(self, in_tensor: Tensor) -> Tensor:
        """`[*bs, 3]` in [0,1] -> `[*bs, num_levels * features_per_level]` (level-major, feature-minor)."""
        assert in_tensor.shape[-1] == 3
        return F.hashgrid_encode(in_tensor, self.hash_table, self.spec)


class SHEncoding(Encoding):
    """Spherical harmonic encoding, `levels` = degree + 1 (encodings.py:752-799). levels=4 (nerfacto) is a kernel."""

    def __init__(self, levels: int = 4, implementation: Literal["hip"] = "hip") -> None:
        super().__init__(in_dim=3)
        check_implementation(implementation, "SHEncoding")
        if levels <= 0 or levels > 5:
            raise ValueError(f"Spherical harmonic encoding only supports 1 to 5 levels, requested {levels}")
        if levels != 4:
            raise ValueError("nerfstudio_amd implements SHEncoding(levels=4) (the nerfacto direction encoding) only")
        self.levels = levels

    def get_out_dim(self) -> int:
        return self.levels**2

    @torch.no_grad()
    def forward(self, in_tensor: Tensor) -> Tensor:
        return F.sh4_encode(in_tensor)
