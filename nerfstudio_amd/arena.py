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

from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import os

import torch
import torch.distributed as dist
from torch.nn import Parameter

from . import functional as F
from .utils import profiler

_COALESCE = [os.environ.get("NSAMD_COALESCE_ALLREDUCE", "1") == "1"]  # one collective call per optimiser group (see all_reduce_group)
_ALIGN = 64  # floats: every tensor starts on a 256-B boundary
_GROUP_ALIGN = _ALIGN * 840  # floats: lcm(1..8) aligned shards per optimiser group (<= 215 KB of zero padding per group)


def _rows_gather(view: torch.Tensor, idx: torch.Tensor, buf: torch.Tensor) -> None:
    """buf[i] = view[idx[i]]: one nsamd kernel on the device (no torch op on the exchange path); torch on CPU (tests)."""
    if view.is_cuda:
        from . import _native as N

        N.check(N.load().nsamd_rows_gather(N.ptr(view), N.ptr(idx), idx.numel(), view.shape[1], N.ptr(buf), N.stream()),
                "rows_gather")
    else:
        torch.index_select(view, 0, idx, out=buf)


def _rows_scatter(view: torch.Tensor, idx: torch.Tensor, buf: torch.Tensor) -> None:
    if view.is_cuda:
        from . import _native as N

        N.check(N.load().nsamd_rows_scatter(N.ptr(view), N.ptr(idx), idx.numel(), view.shape[1], N.ptr(buf), N.stream()),
                "rows_scatter")
    else:
        view.index_copy_(0, idx, buf)


class _GroupHandle:
    """wait() for every collective of a group exchange, then put the compact rows back into the gradient."""

    def __init__(self, handles, post) -> None:
        self.handles, self.post = handles, post
        if not handles:  # synchronous exchange: the collectives are done
            self._finish()

    def _finish(self) -> None:
        for view, idx, buf in self.post:
            _rows_scatter(view, idx, buf)
        self.post = []

    def wait(self) -> None:
        for h in self.handles:
            if h is not None:
                h.wait()
        self.handles = []
        self._finish()


def _single(group=None) -> bool:
    """One rank: the collectives are skipped — unless NSAMD_FORCE_COLLECTIVES=1 asks for them anyway (a one-rank RCCL
    communicator: the way to run the data-parallel path's collectives, streams and compact exchange on a single-GPU box)."""
    return dist.get_world_size(group) == 1 and os.environ.get("NSAMD_FORCE_COLLECTIVES") != "1"


class ParamArena:
    """`params` is either an iterable of Parameters (one group "all") or an ordered dict {group name: parameters} — the
    reference's optimiser groups (models/nerfacto.py:255-260: "fields", "proposal_networks"). Groups are laid out one
    after the other, so each is one contiguous slice for the all-reduce and for Adam, and each has its own step counter:
    the reference steps a group's Adam only on iterations where the group received gradients
    (engine/optimizers.py:160-172 with zero_grad(set_to_none=True)), i.e. the proposal networks' parameters and moments
    stay untouched on the steps where the sampler runs them under no_grad (ray_samplers.py:590,604-609)."""

    def __init__(self, params: Union[Iterable[Parameter], Dict[str, Iterable[Parameter]]], lr: float = 1e-2,
                 betas=(0.9, 0.999), eps: float = 1e-15, bind_grads: bool = True) -> None:
        """bind_grads=False: `param.grad` stays None — the gradients live in the arena only (`grad_lookup`). For a trainer
        whose own optimisers must find nothing to step because the arena's fused Adam does (pipeline.HipPipeline under the
        reference's Trainer: engine/optimizers.py:160-172 steps a group only when a `.grad` is not None)."""
        groups = params if isinstance(params, dict) else {"all": params}
        self.bind_grads = bind_grads
        self.params: List[Parameter] = []
        self.group_params: Dict[str, List[Parameter]] = {}
        seen = set()
        for name, plist in groups.items():
            mine = []
            for p in plist:
                if id(p) not in seen and p.requires_grad:
                    seen.add(id(p))
                    mine.append(p)
            self.group_params[name] = mine
            self.params += mine
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        # groups one after the other, each starting AND ending on a multiple of _GROUP_ALIGN floats: a group's slice then
        # splits into equal 256-B-aligned shards for every world size up to 8 (reduce-scatter / sharded Adam / all-gather)
        self.offsets, total = [], 0
        spans = {}
        for name, plist in self.group_params.items():
            start = total
            for p in plist:
                assert p.dtype == torch.float32 and p.device == dev
                self.offsets.append(total)
                total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            total = (total + _GROUP_ALIGN - 1) // _GROUP_ALIGN * _GROUP_ALIGN
            if plist:
                spans[name] = (start, total)
        self.numel = total
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape) if bind_grads else None
        self.groups: Dict[str, Tuple[int, int]] = spans  # (the padding holds zeros for ever: zero gradient, zero update)
        self.step_counts: Dict[str, int] = {n: 0 for n in self.groups}
        self.lr, self.betas, self.eps = lr, betas, eps

    @property
    def step_count(self) -> int:
        return max(self.step_counts.values())

    def zero_grad(self, groups: Optional[Sequence[str]] = None, skip: Optional[Iterable[Parameter]] = None) -> None:
        """Zero the gradient arena, or only the contiguous slices of the named optimiser groups. `skip`: parameters
        whose gradient the next backward WRITES rather than accumulates (nsamd_hashgrid_encode_bwd_set) — their 67 MB
        need no zero-fill."""
        spans = [(0, self.numel)] if groups is None else [self.groups[name] for name in groups]
        if skip:
            ids = {id(p) for p in skip}
            holes = sorted((off, off + p.numel()) for p, off in zip(self.params, self.offsets) if id(p) in ids)
            cut = []
            for a, b in spans:
                for ha, hb in holes:
                    if hb <= a or ha >= b:
                        continue
                    if ha > a:
                        cut.append((a, ha))
                    a = max(a, hb)
                if a < b:
                    cut.append((a, b))
            spans = cut
        for a, b in spans:
            self.grad[a:b].zero_()
        if not self.bind_grads:
            return
        for p, off in zip(self.params, self.offsets):  # autograd may have replaced .grad; re-point the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def grad_view(self, param: Parameter) -> torch.Tensor:
        """The arena's gradient view of `param` (what `param.grad` is with bind_grads)."""
        off = next(o for p, o in zip(self.params, self.offsets) if p is param)
        return self.grad[off:off + param.numel()].view(param.shape)

    def grad_lookup(self) -> Dict[int, torch.Tensor]:
        """{id(parameter): gradient view}: where the kernel schedules write (train_step.NerfactoTrainStep.grad_lookup)."""
        return {id(p): self.grad[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)}

    def span(self, params: Iterable[Parameter]):
        """(start, end) float offsets of the contiguous arena range holding `params` (they must be adjacent, which is
        how callers build the arena: one parameter group after the other)."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), "parameters of a span must be adjacent in the arena"
        start = self.offsets[idx[0]]
        end = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else self.numel
        return start, end

    def all_reduce_span(self, start: int, end: int, async_op: bool = False, group: Optional[dist.ProcessGroup] = None):
        """Sum one contiguous slice of the gradient arena over the ranks. With async_op the collective runs on the
        communication stream (RCCL) while the caller keeps launching compute; `.wait()` the returned handle before the
        optimiser reads the slice. Returns None when there is nothing to do (single process)."""
        if not (dist.is_available() and dist.is_initialized()) or _single(group):
            return None
        return dist.all_reduce(self.grad[start:end], op=dist.ReduceOp.SUM, group=group, async_op=async_op)

    # ---- compact exchange of hash-table prefixes ---------------------------------------------------------------------
    def register_compact(self, param: Parameter, prefix_rows: int, index: Tensor) -> None:
        """Declare that of the first `prefix_rows` rows of `param` ([rows, F]) only the rows in `index` can ever carry a
        gradient (functional.HashGridSpec.reachable_prefix). all_reduce_group then exchanges those rows as one compact
        buffer instead of the whole prefix (nerfacto main table: 2.7 MB instead of 21 MB of the 67 MB)."""
        off = next(o for p, o in zip(self.params, self.offsets) if p is param)
        feat = param.shape[-1]
        assert param.dim() == 2 and 0 < prefix_rows <= param.shape[0] and index.numel() > 0
        assert int(index.max()) < prefix_rows
        idx = index.to(self.grad.device)
        self._compact = getattr(self, "_compact", {})
        self._compact[id(param)] = (off, prefix_rows, feat, idx, torch.zeros((idx.numel(), feat), device=self.grad.device))

    @profiler.time_function
    def all_reduce_group(self, name: str, async_op: bool = False, group: Optional[dist.ProcessGroup] = None):
        """Sum one optimiser group's gradients over the ranks: dense spans as they lie, registered table prefixes
        through their compact buffers. Returns a handle whose wait() also scatters the reduced rows back (None when
        there is nothing to do)."""
        if not (dist.is_available() and dist.is_initialized()) or _single(group):
            return None
        a, b = self.groups[name]
        compact = sorted((c for c in getattr(self, "_compact", {}).values() if a <= c[0] < b), key=lambda c: c[0])
        pieces, post = [], []
        cursor = a
        for off, rows, feat, idx, buf in compact:
            if off > cursor:
                pieces.append(self.grad[cursor:off])
            view = self.grad[off:off + rows * feat].view(rows, feat)
            _rows_gather(view, idx, buf)
            pieces.append(buf)
            post.append((view, idx, buf))
            cursor = off + rows * feat
        if cursor < b:
            pieces.append(self.grad[cursor:b])
        # ONE collective call for all pieces of the group (RCCL: one grouped launch instead of one kernel + one stream
        # hand-over per piece — on the one-rank rehearsal every collective launch is ~15-25 us of exposed latency)
        handles = None
        if len(pieces) > 1 and hasattr(dist, "all_reduce_coalesced") and _COALESCE[0]:
            import warnings

            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")  # (deprecation notice of the list form; the semantics are what is wanted)
                    handles = [dist.all_reduce_coalesced(pieces, op=dist.ReduceOp.SUM, group=group, async_op=async_op)]
            except (RuntimeError, NotImplementedError, TypeError) as e:
                # a backend without the coalesced form (only ever exercised over gloo and a one-rank RCCL communicator
                # here): one call per piece from now on — raised BEFORE anything was enqueued, so nothing is reduced twice
                _COALESCE[0] = False
                print(f"[nerfstudio_amd.arena] all_reduce_coalesced unavailable ({type(e).__name__}: {e}); "
                      "falling back to one all_reduce per piece", flush=True)
        if handles is None:
            handles = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op) for t in pieces]
        return _GroupHandle(handles if async_op else [], post)

    # ---- sharded optimiser: reduce-scatter -> Adam on this rank's 1/N of the group -> all-gather -----------------------
    # (SURVEY.md §8e "direct RS+AG" / ZeRO-1: the reference steps a replicated torch.optim.Adam on every rank after
    # DDP's all-reduce, engine/optimizers.py:74-193 + pipelines/base_pipeline.py:279-282. Adam is elementwise, so updating
    # only the rank's shard of (p, m, v) from the rank's shard of the summed gradient and gathering the updated parameters
    # gives every rank exactly the parameters replicated Adam gives — with 1/N of the optimiser's HBM traffic, 544 MB per
    # step and rank replicated. Moments outside the rank's shard are never read or written.)
    def shard_span(self, name: str, group: Optional[dist.ProcessGroup] = None) -> Tuple[int, int]:
        """This rank's contiguous (start, end) share of optimiser group `name` (equal shards in rank order)."""
        a, b = self.groups[name]
        if not (dist.is_available() and dist.is_initialized()):
            return a, b
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        assert (b - a) % world == 0, f"group {name}: {b - a} floats do not split into {world} shards"
        n = (b - a) // world
        return a + rank * n, a + (rank + 1) * n

    @profiler.time_function
    def reduce_scatter_group(self, name: str, async_op: bool = False, group: Optional[dist.ProcessGroup] = None):
        """Sum the group's gradients over the ranks INTO this rank's shard of the gradient arena (in place: the output is
        the rank's slice of the input, RCCL's in-place reduce-scatter). Afterwards only `shard_span(name)` of the gradient
        holds the sum; the rest of the slice is stale. Returns the async handle (None: synchronous / nothing to do)."""
        if not (dist.is_available() and dist.is_initialized()) or _single(group):
            return None
        a, b = self.groups[name]
        sa, sb = self.shard_span(name, group)
        return dist.reduce_scatter_tensor(self.grad[sa:sb], self.grad[a:b], op=dist.ReduceOp.SUM, group=group,
                                          async_op=async_op)

    @profiler.time_function
    def all_gather_group(self, name: str, async_op: bool = False, group: Optional[dist.ProcessGroup] = None):
        """Every rank's updated parameter shard -> every rank's full parameter slice (in place)."""
        if not (dist.is_available() and dist.is_initialized()) or _single(group):
            return None
        a, b = self.groups[name]
        sa, sb = self.shard_span(name, group)
        return dist.all_gather_into_tensor(self.flat[a:b], self.flat[sa:sb], group=group, async_op=async_op)

    @profiler.time_function
    def step_shard(self, name: str, grad_scale: float = 1.0, lr: Optional[float] = None,
                   hyper_dev: Optional[Dict[str, torch.Tensor]] = None, group: Optional[dist.ProcessGroup] = None) -> None:
        """Adam on this rank's shard of group `name` only (after reduce_scatter_group, before all_gather_group). The
        group's step counter advances as in `step`: bias corrections are identical on every rank."""
        sa, sb = self.shard_span(name, group)
        self.step_counts[name] += 1
        hd = hyper_dev.get(name) if isinstance(hyper_dev, dict) else hyper_dev
        F.adam_step(self.flat[sa:sb], self.grad[sa:sb], self.exp_avg[sa:sb], self.exp_avg_sq[sa:sb],
                    self.step_counts[name], lr if lr is not None else self.lr, self.betas, self.eps, grad_scale, hd)

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

    @profiler.time_function
    def step(self, grad_scale: float = 1.0, lr: Optional[float] = None, groups: Optional[Sequence[str]] = None,
             hyper_dev: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """One Adam update (csrc/misc.hip) per selected group (default: all), each with its own bias-correction step.
        `hyper_dev[name]` (device floats from functional.adam_hyper) removes every step-dependent host value from the
        launch so it can be replayed from a captured hipGraph."""
        for name in (groups if groups is not None else self.groups):
            a, b = self.groups[name]
            self.step_counts[name] += 1
            hd = hyper_dev.get(name) if isinstance(hyper_dev, dict) else hyper_dev
            F.adam_step(self.flat[a:b], self.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.step_counts[name],
                        lr if lr is not None else self.lr, self.betas, self.eps, grad_scale, hd)
