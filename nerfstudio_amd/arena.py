"""Flat parameter / gradient arena and the data-parallel gradient exchange.

Reference behaviour being replaced: `DistributedDataParallel(model, find_unused_parameters=True)` all-reducing every
parameter gradient in 25 MB buckets each backward (nerfstudio/pipelines/base_pipeline.py:279-282) followed by one
`torch.optim.Adam` per parameter group (engine/optimizers.py:74-193).

MI355X design: every trainable tensor of the path (hash tables, MLP weights, appearance embedding; 77.7 MB fp32 for
nerfacto) is a view into ONE contiguous fp32 buffer, and so is every gradient. One step then needs exactly
  * one memset of the gradient arena,
  * one RCCL all-reduce over xGMI of the whole arena (a single large message instead of DDP's bucket train; the mean
    is folded into the optimiser as grad_scale = 1/world_size, so no extra pass),
  * one fused Adam launch over the arena (csrc/misc.hip), 16 B per lane.
Rays shard by batch: each rank draws its own rays (seed + rank, as scripts/train.py:98) — no data-path collective.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch.nn import Parameter

from . import functional as F

_ALIGN = 64  # floats: every tensor starts on a 256-B boundary


class ParamArena:
    def __init__(self, params: Iterable[Parameter], lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-15) -> None:
        self.params: List[Parameter] = []
        seen = set()
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        self.offsets, total = [], 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            self.offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # autograd may have replaced .grad; re-point the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def all_reduce(self, group: Optional[dist.ProcessGroup] = None) -> float:
        """Sum the gradient arena over the ranks (RCCL when the tensors are on the GPU, gloo on CPU). Returns the scale
        that turns the sum into DDP's mean; it is applied inside the Adam kernel."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world == 1:
            return 1.0
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world

    def broadcast_params(self, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
        """Replicated model: every rank starts from rank `src`'s parameters (DDP does this at wrap time)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat, src=src, group=group)

    def step(self, grad_scale: float = 1.0, lr: Optional[float] = None, hyper_dev: Optional[torch.Tensor] = None) -> None:
        """One Adam update of the whole arena. With `hyper_dev` (device floats from functional.adam_hyper) the launch
        carries no step-dependent host value and can be replayed from a captured hipGraph."""
        self.step_count += 1
        F.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, lr if lr is not None else self.lr,
                    self.betas, self.eps, grad_scale, hyper_dev)
