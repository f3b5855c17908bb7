"""Batched dataclass of tensors: the container contract of the boundary.

Mirrors the behaviour of the reference's `TensorDataclass` (nerfstudio/utils/tensor_dataclass.py:27-331) that code
written against nerfstudio relies on for `RayBundle` / `RaySamples` / `Frustums`: every tensor field is
`[*batch, feature]`; construction broadcasts all fields (and nested containers, and tensors inside dict fields) to the
common batch shape as zero-copy views; `shape / size / ndim / len()`, indexing, `reshape`, `flatten`, `broadcast_to` and
`to` act on the batch dimensions only and return a new container. Fields listed in `_field_custom_dimensions` keep that
many trailing dimensions instead of one (cameras/rays.py has none; the reference's Cameras does).

Implementation: one recursive `_map_tensors(fn)` that rebuilds the container from its transformed fields; non-tensor
fields (callables, ints, `None`, ...) are carried over unchanged.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, Dict, Tuple

import numpy as np
import torch
from torch import Tensor


class TensorDataclass:
    """Base class; subclasses are `@dataclass`es whose tensor fields share leading batch dimensions."""

    _field_custom_dimensions: Dict[str, int] = {}

    # ------------------------------------------------------------------------------------------------- construction
    def __post_init__(self) -> None:
        if not dataclasses.is_dataclass(self):
            raise TypeError("TensorDataclass must be a dataclass")
        shapes = []
        self._collect_batch_shapes({f.name: getattr(self, f.name) for f in dataclasses.fields(self)}, shapes)
        if not shapes:
            raise ValueError("TensorDataclass must have at least one tensor")
        batch = tuple(torch.broadcast_shapes(*shapes))
        for f in dataclasses.fields(self):
            object.__setattr__(self, f.name, self._broadcast_value(f.name, getattr(self, f.name), batch))
        object.__setattr__(self, "_shape", batch)

    def _trailing(self, name: str) -> int:
        return int(self._field_custom_dimensions.get(name, 1)) if isinstance(self._field_custom_dimensions, dict) else 1

    def _collect_batch_shapes(self, items: Dict[str, Any], out: list) -> None:
        for name, v in items.items():
            if isinstance(v, Tensor):
                out.append(tuple(v.shape[: v.dim() - self._trailing(name)]))
            elif isinstance(v, TensorDataclass):
                out.append(tuple(v.shape))
            elif isinstance(v, dict):
                self._collect_batch_shapes(v, out)

    def _broadcast_value(self, name: str, v: Any, batch: Tuple[int, ...]) -> Any:
        if isinstance(v, Tensor):
            k = self._trailing(name)
            return v.broadcast_to((*batch, *v.shape[v.dim() - k:]))
        if isinstance(v, TensorDataclass):
            return v.broadcast_to(batch)
        if isinstance(v, dict):
            return {key: self._broadcast_value(key, x, batch) for key, x in v.items()}
        return v

    # ------------------------------------------------------------------------------------------------- shape queries
    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape

    @property
    def size(self) -> int:
        return int(np.prod(self._shape)) if len(self._shape) else 1

    @property
    def ndim(self) -> int:
        return len(self._shape)

    def __len__(self) -> int:
        if len(self._shape) == 0:
            raise TypeError("len() of a 0-d tensor")
        return self._shape[0]

    def __bool__(self) -> bool:
        if len(self) == 0:
            raise ValueError(f"The truth value of {self.__class__.__name__} when `len(x) == 0` is ambiguous.")
        return True

    # ------------------------------------------------------------------------------------------------- batch operations
    def _map_tensors(self, fn: Callable[[Tensor, int], Tensor], **overrides):
        """New container of the same class with `fn(tensor, trailing_dims)` applied to every tensor field (recursing
        into nested containers and dicts); `overrides` replace fields outright."""

        def walk(name: str, v: Any) -> Any:
            if isinstance(v, Tensor):
                return fn(v, self._trailing(name))
            if isinstance(v, TensorDataclass):
                return v._map_tensors(fn)
            if isinstance(v, dict):
                return {key: walk(key, x) for key, x in v.items()}
            return v

        kw = {f.name: walk(f.name, getattr(self, f.name)) for f in dataclasses.fields(self)}
        kw.update(overrides)
        return dataclasses.replace(self, **kw)

    def __getitem__(self, indices):
        if isinstance(indices, (Tensor, int, slice, type(Ellipsis))):
            indices = (indices,)
        if not isinstance(indices, tuple):
            indices = (indices,)

        def take(t: Tensor, k: int) -> Tensor:
            return t[indices + (slice(None),) * k]

        return self._map_tensors(take)

    def __setitem__(self, indices, value) -> None:
        raise RuntimeError("Index assignment is not supported for TensorDataclass")

    def reshape(self, shape):
        if isinstance(shape, int):
            shape = (shape,)
        return self._map_tensors(lambda t, k: t.reshape((*shape, *t.shape[t.dim() - k:])))

    def flatten(self):
        return self.reshape((-1,))

    def broadcast_to(self, shape):
        return self._map_tensors(lambda t, k: t.broadcast_to((*shape, *t.shape[t.dim() - k:])))

    def to(self, device):
        return self._map_tensors(lambda t, k: t.to(device))

    def pin_memory(self):
        return self._map_tensors(lambda t, k: t.pin_memory())
