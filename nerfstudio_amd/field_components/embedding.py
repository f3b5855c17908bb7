"""Embedding table (reference: nerfstudio/field_components/embedding.py:26-54). The fused main-field kernel gathers
rows of `embedding.weight` itself; `forward` here is the stand-alone lookup (an index op, no arithmetic)."""
import torch
from torch import Tensor

from .base_field_component import FieldComponent


class Embedding(FieldComponent):
    """`in_dim` rows of `out_dim` floats under the reference's parameter name `embedding.weight` (what the appearance
    embedding of a nerfacto checkpoint is stored as)."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__(in_dim=in_dim, out_dim=out_dim)
        self.build_nn_modules()

    def build_nn_modules(self) -> None:
        self.embedding = torch.nn.Embedding(self.in_dim, self.out_dim)

    def mean(self, dim=0):  # the average embedding an eval render may use (fields/nerfacto_field.py:255-258)
        return self.embedding.weight.mean(dim)

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.embedding(in_tensor)
