"""torch.autograd glue over the nsamd C ABI (include/nsamd.h).

Each public function here is one stage of the nerfacto hot path running as hand-written gfx950
kernels. PyTorch only provides device memory, the stream and the autograd graph — no arithmetic of the path is done
by torch ops. Reference citations are relative to /root/reference/nerfstudio/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _native as N


_GRID_CACHE: dict = {}

# When True, the backward kernels of the fused fields accumulate straight into `param.grad` whenever that buffer
# already exists (e.g. the views of arena.ParamArena) and autograd receives None for those inputs. This removes a
# zero-fill, a read-modify-write and an AccumulateGrad pass over every parameter (67 MB for the main hash table) per
# step. Semantics are those of gradient accumulation into a pre-zeroed .grad; off by default.
DIRECT_GRAD = False


def _grad_target(param: Tensor, needed: bool):
    """-> (buffer to accumulate into, value to return to autograd)."""
    if not needed:
        return None, None
    if DIRECT_GRAD and getattr(param, "grad", None) is not None and param.grad.is_contiguous():
        return param.grad, None
    buf = torch.zeros_like(param)
    return buf, buf


# ---------------------------------------------------------------------------------------------------------------
# configuration records
# ---------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class HashGridSpec:
    """Hyper-parameters of one multiresolution hash grid (HashEncoding ctor, field_components/encodings.py:321-347)."""

    num_levels: int
    min_res: int
    max_res: int
    log2_hashmap_size: int
    features_per_level: int = 2

    def __post_init__(self):
        if self.features_per_level != 2:
            raise ValueError("nerfstudio_amd hash grids are built for features_per_level=2 (every nerfacto config)")
        if not (0 < self.num_levels <= N.MAX_LEVELS):
            raise ValueError(f"num_levels must be in [1, {N.MAX_LEVELS}]")

    @property
    def table_size(self) -> int:
        return 2**self.log2_hashmap_size

    @property
    def out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    @property
    def growth_factor(self) -> float:
        if self.num_levels > 1:
            return math.exp((math.log(self.max_res) - math.log(self.min_res)) / (self.num_levels - 1))
        return 1.0

    def scalings(self) -> Tensor:
        """floor(min_res * growth**level) evaluated in fp32 on the host, exactly like encodings.py:342-344 (where
        `np.float64 ** LongTensor` dispatches to torch's __rpow__ and yields fp32)."""
        levels = torch.arange(self.num_levels)
        return torch.floor(self.min_res * self.growth_factor**levels).to(torch.float32)

    def reachable_prefix(self):
        """(rows, index): the leading `rows` rows of the `[L*T, F]` table belong to the coarse levels whose lattice
        (res + 1)^3 is smaller than the table, and `index` (int64, sorted) lists the rows of that prefix a position in
        [0, 1]^3 can ever touch. The torch path hashes every level (encodings.py:398-415), so on those levels all other
        rows keep a zero gradient for ever — SURVEY.md 8a: 332 k of the first 2.6 M rows of the nerfacto main table —
        and a data-parallel exchange only needs the listed ones. Host arithmetic in wrap-around uint32, as the kernels."""
        import numpy as np

        T = self.table_size
        scal = self.scalings().tolist()
        parts, levels = [], 0
        for lvl, s_ in enumerate(scal):
            res = int(s_)
            if (res + 1) ** 3 >= T:
                break
            c = np.arange(res + 1, dtype=np.uint32)
            with np.errstate(over="ignore"):
                h = c[:, None, None] ^ (c[None, :, None] * np.uint32(2654435761)) ^ (c[None, None, :] * np.uint32(805459861))
            parts.append(np.unique((h & np.uint32(T - 1)).astype(np.int64)) + lvl * T)
            levels += 1
        index = torch.from_numpy(np.concatenate(parts)) if parts else torch.zeros(0, dtype=torch.int64)
        return levels * T, index

    def native(self) -> N.Grid:
        g = _GRID_CACHE.get(self)
        if g is None:
            g = N.make_grid(self.num_levels, self.log2_hashmap_size, self.scalings().tolist())
            _GRID_CACHE[self] = g
        return g


@dataclass
class PointSpec:
    """Sample points either as explicit `[M,3]` positions or as rays + bin edges (never materialised)."""

    positions: Optional[Tensor] = None
    origins: Optional[Tensor] = None
    directions: Optional[Tensor] = None
    t_bins: Optional[Tensor] = None

    @property
    def ray_mode(self) -> bool:
        return self.positions is None

    @property
    def num_points(self) -> int:
        if self.positions is not None:
            return self.positions.shape[0]
        return self.t_bins.shape[0] * (self.t_bins.shape[1] - 1)

    @property
    def samples_per_ray(self) -> int:
        return 0 if self.positions is not None else self.t_bins.shape[1] - 1

    def tensors(self) -> Tuple[Optional[Tensor], ...]:
        return (self.positions, self.origins, self.directions, self.t_bins)

    def native(self) -> N.Points:
        N.require_cuda(*self.tensors())
        return N.make_points(self.positions, self.origins, self.directions, self.t_bins, self.samples_per_ray)

    def needs_position_grad(self) -> bool:
        if self.positions is not None:
            return self.positions.requires_grad
        return bool(self.origins.requires_grad or self.directions.requires_grad)


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _spec_from_flat(positions, origins, directions, t_bins) -> PointSpec:
    return PointSpec(_f32c(positions), _f32c(origins), _f32c(directions), _f32c(t_bins))


_SCATTER_WS: dict = {}
_FIELD_WS: dict = {}


def field_bwd_workspace(device) -> Tuple[Tensor, int]:
    """Scratch for the per-workgroup weight-gradient partials of nsamd_field_mlp_bwd (<= 1024 workgroups x 12544)."""
    ws = _FIELD_WS.get(str(device))
    if ws is None:
        ws = torch.empty(1024 * 12544, device=device, dtype=torch.float32)
        _FIELD_WS[str(device)] = ws
    return ws, ws.numel()


_SCATTER_WS_MAX = 8  # cached workspaces (least recently used ones are dropped: a main-table workspace is ~0.75 GB)


DENSITY_WS_FLOATS = 1024 * 1156  # <= 1024 workgroups x (64 x 16 + 2 x 64 + 1, padded to 16 B) partial sums


def density_bwd_workspace(device, slot: int = 0) -> Tensor:
    """Scratch rows for the weight-gradient partials of nsamd_density_mlp_bwd[_gated] (summed in a fixed order). One
    buffer per (device, slot): calls that may overlap on different streams take different slots."""
    key = (str(device), "density", slot)
    ws = _FIELD_WS.get(key)
    if ws is None:
        ws = torch.empty(DENSITY_WS_FLOATS, device=device, dtype=torch.float32)
        _FIELD_WS[key] = ws
    return ws


_SCATTER_BUCKET = 1 << 16  # workspaces are sized for M rounded up to a multiple of this many points


def _scatter_workspace(grid: HashGridSpec, device, num_points: int = 0, write_only: bool = False) -> Tuple[Optional[Tensor], int]:
    """Device scratch for the table-gradient scatter (csrc/scatter.hip): per (level, tile) a queue of 16-B records, the
    per-workgroup segment counts, a spill list. The library says how many words it wants
    (nsamd_hashgrid_encode_bwd_workspace) and how many leading words must start as zero (..._workspace_state: header +
    cursors, a few KB that depend on the grid only — the kernels leave them at zero); the bulk is never initialised.
    Cached per (grid, M rounded up to 64 k points, device): the packed instant-ngp path changes M every step
    (occupancy-grid sampling, DynamicBatch), and a cache keyed by the exact M would allocate a fresh ~GB workspace per
    step. A cached workspace serves every M of its bucket — the kernels lay the buffer out for the M of
    the call — and is re-allocated only if the library asks for more words than it holds. At most _SCATTER_WS_MAX
    entries; evicting one waits for the device first (a side-stream scatter may still be reading it). One workspace
    serves one call at a time: callers that overlap scatters on different streams must use different (grid, bucket)
    keys or serialise (train_step does)."""
    bucket = max(1, -(-int(num_points) // _SCATTER_BUCKET)) * _SCATTER_BUCKET
    key = (grid, bucket, str(device), write_only)
    lib = N.load()
    ws = _SCATTER_WS.get(key)
    if ws is not None:
        need = int(lib.nsamd_hashgrid_encode_bwd_workspace(grid.native(), num_points, 1 if write_only else 0)) \
            if num_points != bucket else ws.numel()
        if 0 < need <= ws.numel():
            _SCATTER_WS[key] = _SCATTER_WS.pop(key)  # most recently used last
            return ws, ws.numel()
        if need <= 0:
            return None, 0
        torch.cuda.synchronize(device)  # (never seen: the plan grows with M) replace it by a larger one
        _SCATTER_WS.pop(key)
    words = max(int(lib.nsamd_hashgrid_encode_bwd_workspace(grid.native(), m, 1 if write_only else 0))
                for m in {bucket, max(int(num_points), 1)})
    if words <= 0:
        return None, 0
    ws = torch.empty(words, device=device, dtype=torch.float32)
    state = int(lib.nsamd_hashgrid_encode_bwd_workspace_state(grid.native(), bucket))
    ws[:state].zero_()
    while len(_SCATTER_WS) >= _SCATTER_WS_MAX:
        torch.cuda.synchronize(device)  # no kernel on any stream may still be using the evicted buffer
        _SCATTER_WS.pop(next(iter(_SCATTER_WS)))
    _SCATTER_WS[key] = ws
    return ws, ws.numel()


def _producer_scatter_workspace(grid: HashGridSpec, device, num_points: int) -> Tuple[Optional[Tensor], int]:
    """Scratch of nsamd_field_mlp_bwd_scatter (the main field's backward emits the scatter's records itself): one static
    segment per (MLP workgroup, table tile) + dynamic areas + the worst-case spill list, ~1.4 GB for the nerfacto main table at
    196 608 points — untouched bulk, only the leading state words are ever initialised. Cached like `_scatter_workspace`."""
    bucket = max(1, -(-int(num_points) // _SCATTER_BUCKET)) * _SCATTER_BUCKET
    key = (grid, bucket, str(device), "producer")
    lib = N.load()
    ws = _SCATTER_WS.get(key)
    state = C.c_int64(0)
    if ws is not None:
        need = int(lib.nsamd_field_mlp_bwd_scatter_workspace(grid.native(), num_points, C.byref(state)))
        if 0 < need <= ws.numel():
            _SCATTER_WS[key] = _SCATTER_WS.pop(key)
            return ws, ws.numel()
        if need <= 0:
            return None, 0
        torch.cuda.synchronize(device)
        _SCATTER_WS.pop(key)
    words, states = 0, 0
    for m in {bucket, max(int(num_points), 1)}:
        w = int(lib.nsamd_field_mlp_bwd_scatter_workspace(grid.native(), m, C.byref(state)))
        words, states = max(words, w), max(states, int(state.value))
    if words <= 0:
        return None, 0
    ws = torch.empty(words, device=device, dtype=torch.float32)
    ws[:states].zero_()
    while len(_SCATTER_WS) >= _SCATTER_WS_MAX:
        torch.cuda.synchronize(device)
        _SCATTER_WS.pop(next(iter(_SCATTER_WS)))
    _SCATTER_WS[key] = ws
    return ws, ws.numel()


def scatter_events(ws: Tensor) -> Tuple[int, int, int]:
    """(spilled, unordered, lost) record counts of a scatter workspace since it was created
    (nsamd_hashgrid_scatter_events): `unordered` > 0 means some call was exact but not bit-reproducible."""
    ev = (C.c_uint32 * 3)()
    N.check(N.load().nsamd_hashgrid_scatter_events(N.ptr(ws), C.cast(ev, C.c_void_p), N.stream()), "scatter_events")
    return int(ev[0]), int(ev[1]), int(ev[2])


def _position_grads(spec: PointSpec, dpos: Tensor):
    """dL/dpositions [M,3] -> gradients of whatever the spec was built from."""
    if not spec.ray_mode:
        return dpos, None, None
    n, s1 = spec.t_bins.shape
    d = dpos.view(n, s1 - 1, 3)
    mid = ((spec.t_bins[:, :-1] + spec.t_bins[:, 1:]) / 2)[..., None]
    return None, d.sum(dim=1), (d * mid).sum(dim=1)


# ---------------------------------------------------------------------------------------------------------------
# a8  stand-alone hash encoding (Encoding API: [M,3] in [0,1] -> [M, 2L])
# ---------------------------------------------------------------------------------------------------------------
class _HashEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, table: Tensor, grid: HashGridSpec):
        N.require_cuda(x, table)
        x = _f32c(x)
        M = x.shape[0]
        out = torch.empty((M, grid.out_dim), device=x.device, dtype=torch.float32)
        pts = N.make_points(positions=x)
        N.check(
            N.load().nsamd_hashgrid_encode_fwd(pts, M, N.XFORM_NONE, N.Aabb(), N.ptr(table), grid.native(), N.ptr(out),
                                              grid.out_dim, 1, None, N.stream()),
            "hashgrid_encode_fwd",
        )
        ctx.save_for_backward(x, table)
        ctx.grid = grid
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        x, table = ctx.saved_tensors
        grid: HashGridSpec = ctx.grid
        gout = _f32c(gout)
        M = x.shape[0]
        dtable = torch.zeros_like(table) if ctx.needs_input_grad[1] else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if dtable is not None or dx is not None:
            pts = N.make_points(positions=x)
            ws, ws_n = _scatter_workspace(grid, x.device, M)
            N.check(
                N.load().nsamd_hashgrid_encode_bwd(pts, M, N.XFORM_NONE, N.Aabb(), N.ptr(table), grid.native(),
                                                  N.ptr(gout), grid.out_dim, 1, N.ptr(dtable), N.ptr(dx), N.ptr(ws),
                                                  ws_n, N.stream()),
                "hashgrid_encode_bwd",
            )
        return dx, dtable, None


def hashgrid_encode(x: Tensor, table: Tensor, grid: HashGridSpec) -> Tensor:
    """HashEncoding.pytorch_fwd semantics (encodings.py:417-458) on `[*bs,3]` -> `[*bs, 2L]`."""
    shape = x.shape[:-1]
    return _HashEncodeFn.apply(x.reshape(-1, 3), table, grid).view(*shape, grid.out_dim)


class _SpecEncodeFn(torch.autograd.Function):
    """Hash features of sample points given as a PointSpec (positions or rays + bins), with the fields' position
    normalisation (aabb or L-inf contraction) and selector fused into the gather kernel. -> (enc [M, 2L], selector [M])."""

    @staticmethod
    def forward(ctx, positions, origins, directions, t_bins, table, grid: HashGridSpec, transform: int, aabb):
        spec = _spec_from_flat(positions, origins, directions, t_bins)
        N.require_cuda(table)
        M = spec.num_points
        enc = torch.empty((M, grid.out_dim), device=table.device, dtype=torch.float32)
        sel = torch.empty((M,), device=table.device, dtype=torch.float32)
        box = aabb if isinstance(aabb, N.Aabb) else N.make_aabb(aabb)
        N.check(N.load().nsamd_hashgrid_encode_fwd(spec.native(), M, transform, box, N.ptr(table), grid.native(), N.ptr(enc),
                                                   grid.out_dim, 1, N.ptr(sel), N.stream()), "hashgrid_encode_fwd")
        ctx.spec, ctx.grid, ctx.transform, ctx.box, ctx.table_ref = spec, grid, transform, box, table
        ctx.save_for_backward(table)
        ctx.mark_non_differentiable(sel)
        return enc, sel

    @staticmethod
    def backward(ctx, genc: Tensor, _gsel):
        (table,) = ctx.saved_tensors
        spec: PointSpec = ctx.spec
        M = spec.num_points
        genc = _f32c(genc)
        need_pos = any(ctx.needs_input_grad[:3])
        ttable, rtable = _grad_target(ctx.table_ref, ctx.needs_input_grad[4])
        dpos = torch.empty((M, 3), device=table.device, dtype=torch.float32) if need_pos else None
        if ttable is not None or dpos is not None:
            ws, ws_n = _scatter_workspace(ctx.grid, table.device, M)
            N.check(N.load().nsamd_hashgrid_encode_bwd(spec.native(), M, ctx.transform, ctx.box, N.ptr(table),
                                                       ctx.grid.native(), N.ptr(genc), ctx.grid.out_dim, 1, N.ptr(ttable),
                                                       N.ptr(dpos), N.ptr(ws), ws_n, N.stream()), "hashgrid_encode_bwd")
        gp, go, gd = _position_grads(spec, dpos) if dpos is not None else (None, None, None)
        return gp, go, gd, None, rtable, None, None, None


def spec_encode(spec: PointSpec, table: Tensor, grid: HashGridSpec, transform: int, aabb) -> Tuple[Tensor, Tensor]:
    """(hash features `[M, 2L]`, selector `[M]`) of the points of `spec` after the field's position normalisation
    (fields/density_fields.py:95-103 + encodings.py:417-458) — the encoding half of a field, for heads that are not one
    of the fused MLP shapes."""
    return _SpecEncodeFn.apply(spec.positions, spec.origins, spec.directions, spec.t_bins, table, grid, transform, aabb)


def sh4_encode(directions: Tensor) -> Tensor:
    """SHEncoding(levels=4).pytorch_fwd (encodings.py:791-794); no gradient (it is @torch.no_grad there)."""
    N.require_cuda(directions)
    shape = directions.shape[:-1]
    d = _f32c(directions.detach().reshape(-1, 3))
    out = torch.empty((d.shape[0], 16), device=d.device, dtype=torch.float32)
    N.check(N.load().nsamd_sh4_encode(N.ptr(d), d.shape[0], N.ptr(out), N.stream()), "sh4_encode")
    return out.view(*shape, 16)


_FREQS: dict = {}


def nerf_encode(spec: PointSpec, num_frequencies: int, min_freq_exp: float, max_freq_exp: float,
                include_input: bool = False) -> Tensor:
    """NeRFEncoding.forward (encodings.py:148-189, no covariances) on the points of `spec` -> `[M, 6 F (+3)]`. The
    frequencies are evaluated with torch on the host once (`2 ** linspace`), as the reference does per call. No gradient
    w.r.t. the points (vanilla-nerf has none to take: no camera optimiser, no learned warp)."""
    N.require_cuda(*spec.tensors())
    if any(t is not None and t.requires_grad for t in spec.tensors()):
        raise RuntimeError("nsamd NeRFEncoding has no backward: the encoded points must not require grad")
    dev = next(t for t in spec.tensors() if t is not None).device
    key = (num_frequencies, float(min_freq_exp), float(max_freq_exp), dev)
    if key not in _FREQS:
        _FREQS[key] = (2 ** torch.linspace(min_freq_exp, max_freq_exp, num_frequencies)).to(dev)
    spec = _spec_from_flat(*spec.tensors())
    M = spec.num_points
    out = torch.empty((M, 6 * num_frequencies + (3 if include_input else 0)), device=dev, dtype=torch.float32)
    N.check(N.load().nsamd_nerf_encode(spec.native(), M, N.ptr(_FREQS[key]), num_frequencies, int(bool(include_input)),
                                       N.ptr(out), N.stream()), "nerf_encode")
    return out


def contract_linf(x: Tensor) -> Tensor:
    """SceneContraction(order=inf).forward (spatial_distortions.py:66-69), forward only (the fused fields carry
    the Jacobian inside their own backward)."""
    N.require_cuda(x)
    shape = x.shape
    xf = _f32c(x.detach().reshape(-1, 3))
    out = torch.empty_like(xf)
    N.check(N.load().nsamd_contract_linf(N.ptr(xf), xf.shape[0], N.ptr(out), N.stream()), "contract_linf")
    return out.view(shape)


# ---------------------------------------------------------------------------------------------------------------
# a9  stand-alone dense layer / MLP of arbitrary width                              (field_components/mlp.py:160-179)
# ---------------------------------------------------------------------------------------------------------------
_ACT = {None: 0, "relu": 1, "sigmoid": 2, "softplus": 3}


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, b: Optional[Tensor], act: int):
        N.require_cuda(x, W, b)
        x = _f32c(x)
        M, K = x.shape
        Nout = W.shape[0]
        y = torch.empty((M, Nout), device=x.device, dtype=torch.float32)
        N.check(N.load().nsamd_linear_fwd(N.ptr(x), N.ptr(W), N.ptr(b), M, K, Nout, act, N.ptr(y), N.stream()), "linear_fwd")
        ctx.save_for_backward(x, W, y)
        ctx.act, ctx.has_bias = act, b is not None
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, W, y = ctx.saved_tensors
        gy = _f32c(gy)
        M, K = x.shape
        Nout = W.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.zeros_like(W) if ctx.needs_input_grad[1] else None
        db = torch.zeros((Nout,), device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if dW is None and db is not None:
            dW = torch.zeros_like(W)
        N.check(N.load().nsamd_linear_bwd(N.ptr(x), N.ptr(W), N.ptr(y), N.ptr(gy), M, K, Nout, ctx.act, N.ptr(dx), N.ptr(dW),
                                          N.ptr(db), N.stream()), "linear_bwd")
        return dx, (dW if ctx.needs_input_grad[1] else None), db, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], activation: Optional[str] = None) -> Tensor:
    """act(x W^T + b) on `[*bs, K]` with `weight [N,K]` (nn.Linear layout); activation None | "relu" | "sigmoid" |
    "softplus"."""
    shape = x.shape[:-1]
    y = _LinearFn.apply(x.reshape(-1, x.shape[-1]), weight, bias, _ACT[activation])
    return y.view(*shape, weight.shape[0])


# ---------------------------------------------------------------------------------------------------------------
# a6  proposal density field: contraction -> hash grid -> MLP -> trunc_exp           (fields/density_fields.py:94-117)
# ---------------------------------------------------------------------------------------------------------------
class _DensityFieldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, origins, directions, t_bins, table, W0, b0, W1, b1, grid: HashGridSpec,
                transform: int, aabb, avg_density: float):
        spec = _spec_from_flat(positions, origins, directions, t_bins)
        N.require_cuda(table, W0, b0, W1, b1)
        lib = N.load()
        M = spec.num_points
        dev = table.device
        enc = torch.empty((grid.out_dim, M), device=dev, dtype=torch.float32)  # feature-major
        sel = torch.empty((M,), device=dev, dtype=torch.float32)
        g = grid.native()
        box = aabb if isinstance(aabb, N.Aabb) else N.make_aabb(aabb)
        N.check(lib.nsamd_hashgrid_encode_fwd(spec.native(), M, transform, box, N.ptr(table), g, N.ptr(enc), 1, M,
                                              N.ptr(sel), N.stream()), "hashgrid_encode_fwd")
        mlp = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0], avg_density)
        density = torch.empty((M,), device=dev, dtype=torch.float32)
        pre = torch.empty((M,), device=dev, dtype=torch.float32)
        N.check(lib.nsamd_density_mlp_fwd(N.ptr(enc), N.ptr(sel), M, mlp, N.ptr(density), N.ptr(pre), N.stream()),
                "density_mlp_fwd")
        ctx.spec, ctx.grid, ctx.transform, ctx.box, ctx.avg = spec, grid, transform, box, avg_density
        ctx.param_refs = (table, W0, b0, W1, b1)
        ctx.save_for_backward(table, W0, b0, W1, b1, enc, sel, pre)
        return density

    @staticmethod
    def backward(ctx, gdens: Tensor):
        table, W0, b0, W1, b1, enc, sel, pre = ctx.saved_tensors
        spec: PointSpec = ctx.spec
        lib = N.load()
        M = spec.num_points
        gdens = _f32c(gdens)
        denc = torch.empty_like(enc)
        refs = ctx.param_refs
        (tW0, rW0), (tb0, rb0), (tW1, rW1), (tb1, rb1) = (_grad_target(r, True) for r in refs[1:])
        mlp = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0], ctx.avg)
        dws = density_bwd_workspace(table.device)
        N.check(lib.nsamd_density_mlp_bwd(N.ptr(enc), N.ptr(sel), N.ptr(pre), N.ptr(gdens), M, mlp, N.ptr(denc),
                                          N.ptr(tW0), N.ptr(tb0), N.ptr(tW1), N.ptr(tb1), N.ptr(dws), dws.numel(),
                                          N.stream()),
                "density_mlp_bwd")
        need_pos = any(ctx.needs_input_grad[:3])
        ttable, rtable = _grad_target(refs[0], ctx.needs_input_grad[4])
        dpos = torch.empty((M, 3), device=table.device, dtype=torch.float32) if need_pos else None
        if ttable is not None or dpos is not None:
            ws, ws_n = _scatter_workspace(ctx.grid, table.device, M)
            N.check(lib.nsamd_hashgrid_encode_bwd(spec.native(), M, ctx.transform, ctx.box, N.ptr(table),
                                                  ctx.grid.native(), N.ptr(denc), 1, M, N.ptr(ttable), N.ptr(dpos),
                                                  N.ptr(ws), ws_n, N.stream()), "hashgrid_encode_bwd")
        gp, go, gd = _position_grads(spec, dpos) if dpos is not None else (None, None, None)
        return gp, go, gd, None, rtable, rW0, rb0, rW1, rb1, None, None, None, None


def density_field(spec: PointSpec, table: Tensor, W0: Tensor, b0: Tensor, W1: Tensor, b1: Tensor, grid: HashGridSpec,
                  transform: int, aabb: Optional[Tensor], average_init_density: float) -> Tensor:
    """HashMLPDensityField.get_density on M points -> density `[M]`."""
    return _DensityFieldFn.apply(spec.positions, spec.origins, spec.directions, spec.t_bins, table, W0, b0, W1, b1,
                                 grid, transform, aabb, float(average_init_density))


# ---------------------------------------------------------------------------------------------------------------
# a10 + a12  nerfacto main field                                                   (fields/nerfacto_field.py:203-310)
# ---------------------------------------------------------------------------------------------------------------
class _NerfactoFieldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, origins, directions, t_bins, table, bW0, bb0, bW1, bb1, hW0, hb0, hW1, hb1, hW2, hb2,
                appearance, view_dirs, camera_indices, appearance_const, dir_group: int, grid: HashGridSpec,
                transform: int, aabb, avg_density: float):
        spec = _spec_from_flat(positions, origins, directions, t_bins)
        params = (bW0, bb0, bW1, bb1, hW0, hb0, hW1, hb1, hW2, hb2)
        N.require_cuda(table, *params, view_dirs, appearance, camera_indices, appearance_const)
        if tuple(bW0.shape) != (64, 32) or tuple(bW1.shape) != (16, 64) or tuple(hW1.shape) != (64, 64) or \
                tuple(hW2.shape) != (3, 64) or hW0.shape[0] != 64 or hW0.shape[1] not in (31, 63):
            raise RuntimeError(
                "nsamd main-field kernels are built for the nerfacto shape: L=16,F=2 -> 64 -> 16, head 31|63 -> 64 -> 64 "
                f"-> 3; got base {tuple(bW0.shape)}/{tuple(bW1.shape)}, head {tuple(hW0.shape)}/{tuple(hW1.shape)}/"
                f"{tuple(hW2.shape)}")
        lib = N.load()
        M = spec.num_points
        dev = table.device
        view_dirs = _f32c(view_dirs)
        cams = camera_indices.contiguous().to(torch.int64) if camera_indices is not None else None
        enc = torch.empty((grid.out_dim, M), device=dev, dtype=torch.float32)
        sel = torch.empty((M,), device=dev, dtype=torch.float32)
        box = aabb if isinstance(aabb, N.Aabb) else N.make_aabb(aabb)
        N.check(lib.nsamd_hashgrid_encode_fwd(spec.native(), M, transform, box, N.ptr(table), grid.native(),
                                              N.ptr(enc), 1, M, N.ptr(sel), N.stream()), "hashgrid_encode_fwd")
        mlp = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(appearance),
                         appearance.shape[0] if appearance is not None else 0, avg_density)
        density = torch.empty((M,), device=dev, dtype=torch.float32)
        rgb = torch.empty((M, 3), device=dev, dtype=torch.float32)
        # samples grouped by ray, a whole number of 16-sample tiles per ray: head layer 0's share of the 48 per-ray inputs once
        # per ray (include/nsamd.h, nsamd_field_mlp.ray_terms)
        ctx.ray_terms = ctx.ray_inputs = None
        if dir_group % 16 == 0 and M % dir_group == 0 and M > 0 and os.environ.get("NSAMD_RAY_TERMS", "1") == "1":
            rays = M // dir_group
            has_app = cams is not None or appearance_const is not None
            ctx.ray_terms = torch.empty((rays, 64), device=dev, dtype=torch.float32)
            ctx.ray_inputs = torch.empty((rays, 48 if has_app else 16), device=dev, dtype=torch.float32)
            N.check(lib.nsamd_field_ray_terms(N.ptr(view_dirs), N.ptr(cams), N.ptr(appearance_const), rays, mlp,
                                              N.ptr(ctx.ray_terms), N.ptr(ctx.ray_inputs), N.stream()), "field_ray_terms")
            mlp.ray_terms, mlp.ray_inputs = N.ptr(ctx.ray_terms), N.ptr(ctx.ray_inputs)
        N.check(lib.nsamd_field_mlp_fwd(N.ptr(enc), N.ptr(sel), N.ptr(view_dirs), N.ptr(cams),
                                        N.ptr(appearance_const), dir_group, M, mlp, N.ptr(density), N.ptr(rgb),
                                        N.stream()), "field_mlp_fwd")
        ctx.spec, ctx.grid, ctx.transform, ctx.box, ctx.avg, ctx.dir_group = spec, grid, transform, box, avg_density, dir_group
        ctx.cams, ctx.app_const, ctx.has_app = cams, appearance_const, appearance is not None
        ctx.param_refs = (table, *params, appearance)
        ctx.save_for_backward(table, *params, appearance if appearance is not None else table.new_empty(0), enc, sel,
                              view_dirs)
        return density, rgb

    @staticmethod
    def backward(ctx, gdens: Tensor, grgb: Tensor):
        saved = ctx.saved_tensors
        table, params, appearance, enc, sel, view_dirs = saved[0], saved[1:11], saved[11], saved[12], saved[13], saved[14]
        if not ctx.has_app:
            appearance = None
        spec: PointSpec = ctx.spec
        lib = N.load()
        M = spec.num_points
        gdens = _f32c(gdens) if gdens is not None else torch.zeros((M,), device=table.device)
        grgb = _f32c(grgb) if grgb is not None else torch.zeros((M, 3), device=table.device)
        denc = torch.empty_like(enc)
        refs = ctx.param_refs
        targets = [_grad_target(r, True) for r in refs[1:11]]
        tparams, gparams = [t for t, _ in targets], [r for _, r in targets]
        tapp, gapp = _grad_target(refs[11], True) if (appearance is not None and ctx.cams is not None) else (None, None)
        mlp = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(appearance),
                         appearance.shape[0] if appearance is not None else 0, ctx.avg)
        if ctx.ray_terms is not None:
            mlp.ray_terms, mlp.ray_inputs = N.ptr(ctx.ray_terms), N.ptr(ctx.ray_inputs)
        grads = N.FieldMlpGrads(*(N.ptr(g) for g in tparams), N.ptr(tapp))
        fws, fws_n = field_bwd_workspace(table.device)
        N.check(lib.nsamd_field_mlp_bwd(N.ptr(enc), N.ptr(sel), N.ptr(view_dirs), N.ptr(ctx.cams),
                                        N.ptr(ctx.app_const), ctx.dir_group, M, mlp, N.ptr(gdens), N.ptr(grgb),
                                        N.ptr(denc), grads, N.ptr(fws), fws_n, N.stream()), "field_mlp_bwd")
        need_pos = any(ctx.needs_input_grad[:3])
        ttable, dtable = _grad_target(refs[0], ctx.needs_input_grad[4])
        dpos = torch.empty((M, 3), device=table.device, dtype=torch.float32) if need_pos else None
        if ttable is not None or dpos is not None:
            ws, ws_n = _scatter_workspace(ctx.grid, table.device, M)
            N.check(lib.nsamd_hashgrid_encode_bwd(spec.native(), M, ctx.transform, ctx.box, N.ptr(table),
                                                  ctx.grid.native(), N.ptr(denc), 1, M, N.ptr(ttable), N.ptr(dpos),
                                                  N.ptr(ws), ws_n, N.stream()), "hashgrid_encode_bwd")
        gp, go, gd = _position_grads(spec, dpos) if dpos is not None else (None, None, None)
        return (gp, go, gd, None, dtable, *gparams, gapp, None, None, None, None, None, None, None, None)


def nerfacto_field(spec: PointSpec, table: Tensor, base_params: Sequence[Tensor], head_params: Sequence[Tensor],
                   appearance: Optional[Tensor], view_dirs: Tensor, camera_indices: Optional[Tensor],
                   appearance_const: Optional[Tensor], dir_group: int, grid: HashGridSpec, transform: int,
                   aabb: Optional[Tensor], average_init_density: float) -> Tuple[Tensor, Tensor]:
    """NerfactoField.forward on M points -> (density `[M]`, rgb `[M,3]`).

    view_dirs `[num_dirs,3]` / camera_indices `[num_dirs]`: point p uses row p // dir_group.
    camera_indices None -> every point uses `appearance_const` `[32]` (eval: mean or zeros, nerfacto_field.py:253-261);
    both None -> the field has no appearance embedding.
    """
    return _NerfactoFieldFn.apply(spec.positions, spec.origins, spec.directions, spec.t_bins, table, *base_params,
                                  *head_params, appearance, view_dirs, camera_indices, appearance_const, int(dir_group),
                                  grid, transform, aabb, float(average_init_density))


# ---------------------------------------------------------------------------------------------------------------
# a4 / a13 / a14  samplers
# ---------------------------------------------------------------------------------------------------------------
_LINSPACE_CACHE: dict = {}


def _linspace(kind: str, num_samples: int, device) -> Tensor:
    """Host-evaluated torch.linspace tables (so the fp32 values are bit-identical to the reference's CPU path)."""
    key = (kind, num_samples, str(device))
    t = _LINSPACE_CACHE.get(key)
    if t is None:
        if kind == "edges":  # ray_samplers.py:100
            t = torch.linspace(0.0, 1.0, num_samples + 1)
        else:  # "u": ray_samplers.py:317 / :326
            nb = num_samples + 1
            t = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
        t = t.to(device)
        _LINSPACE_CACHE[key] = t
    return t


@torch.no_grad()
def piecewise_bins(nears: Tensor, fars: Tensor, num_samples: int, jitter: Optional[Tensor], spacing: int = 0
                   ) -> Tuple[Tensor, Tensor]:
    """UniformLinDispPiecewiseSampler (ray_samplers.py:78-128, 225-248). nears/fars `[N]` or `[N,1]`;
    jitter = the U[0,1) draws: `[N,1]` (single_jitter), `[N,S+1]` (one per bin edge) or None (eval). Returns
    (s_bins, t_bins) `[N, S+1]`."""
    N.require_cuda(nears, fars, jitter)
    nears, fars = _f32c(nears.reshape(-1)), _f32c(fars.reshape(-1))
    n = nears.shape[0]
    per_edge = _jitter_layout(jitter, n, num_samples + 1)
    jitter = _f32c(jitter.reshape(-1)) if jitter is not None else None
    s_bins = torch.empty((n, num_samples + 1), device=nears.device, dtype=torch.float32)
    t_bins = torch.empty_like(s_bins)
    edges = _linspace("edges", num_samples, nears.device)
    N.check(N.load().nsamd_piecewise_bins(N.ptr(nears), N.ptr(fars), N.ptr(edges), N.ptr(jitter), per_edge, n, num_samples,
                                          int(spacing), N.ptr(s_bins), N.ptr(t_bins), N.stream()), "piecewise_bins")
    return s_bins, t_bins


def _jitter_layout(jitter: Optional[Tensor], num_rays: int, num_edges: int) -> int:
    """0: one draw per ray (or none), 1: one per bin edge; anything else is a caller error."""
    if jitter is None or jitter.numel() == num_rays:
        return 0
    if jitter.numel() != num_rays * num_edges:
        raise ValueError(f"jitter must hold one draw per ray ({num_rays}) or per bin edge ({num_rays} x {num_edges}), "
                         f"got {tuple(jitter.shape)}")
    return 1


class _WeightsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t_bins: Tensor, density: Tensor):
        N.require_cuda(t_bins, density)
        t_bins, density = _f32c(t_bins), _f32c(density)
        n, s = density.shape
        w = torch.empty_like(density)
        N.check(N.load().nsamd_weights_fwd(N.ptr(t_bins), N.ptr(density), n, s, N.ptr(w), N.stream()), "weights_fwd")
        ctx.save_for_backward(t_bins, density)
        return w

    @staticmethod
    def backward(ctx, gw: Tensor):
        t_bins, density = ctx.saved_tensors
        n, s = density.shape
        gw = _f32c(gw)
        gd = torch.empty_like(density)
        N.check(N.load().nsamd_weights_bwd(N.ptr(t_bins), N.ptr(density), N.ptr(gw), n, s, N.ptr(gd), N.stream()),
                "weights_bwd")
        return None, gd


def weights_from_density(t_bins: Tensor, density: Tensor) -> Tensor:
    """RaySamples.get_weights (cameras/rays.py:129-152): `[N,S+1]`, `[N,S]` -> `[N,S]`."""
    return _WeightsFn.apply(t_bins, density)


@torch.no_grad()
def pdf_resample(s_bins_prev: Tensor, weights: Tensor, num_samples: int, jitter: Optional[Tensor], nears: Tensor,
                 fars: Tensor, anneal: float = 1.0, histogram_padding: float = 0.01, eps: float = 1e-5,
                 return_indices: bool = False, anneal_dev: Optional[Tensor] = None, spacing: int = 0,
                 include_original: bool = False):
    """PDFSampler.generate_ray_samples (ray_samplers.py:276-372) incl. the weight anneal (ray_samplers.py:601). Returns
    (s_bins, t_bins[, inds]): `[N, S+1]` edges, or `[N, S_prev+S+2]` with include_original (the new edges merged into the
    existing ones, :356-357); inds int32 `[N, S+1]` (the searchsorted result of the new edges). jitter: `[N,1]`
    (single_jitter), `[N,S+1]` (one per new edge) or None (eval)."""
    N.require_cuda(s_bins_prev, weights, nears, fars, jitter)
    s_bins_prev, weights = _f32c(s_bins_prev), _f32c(weights.detach())
    nears, fars = _f32c(nears.reshape(-1)), _f32c(fars.reshape(-1))
    n, s_prev = weights.shape
    nb = num_samples + 1
    per_edge = _jitter_layout(jitter, n, nb)
    jitter = _f32c(jitter.reshape(-1)) if jitter is not None else None
    dev = weights.device
    s_bins = torch.empty((n, nb + (s_prev + 1 if include_original else 0)), device=dev, dtype=torch.float32)
    t_bins = torch.empty_like(s_bins)
    inds = torch.empty((n, nb), device=dev, dtype=torch.int32) if return_indices else None
    u_base = _linspace("u", num_samples, dev)
    N.check(N.load().nsamd_pdf_resample(N.ptr(s_bins_prev), N.ptr(weights), s_prev, N.ptr(u_base), N.ptr(jitter),
                                        N.ptr(nears), N.ptr(fars), float(anneal), N.ptr(anneal_dev),
                                        float(histogram_padding), float(eps),
                                        1.0 / (2 * nb), int(spacing), per_edge, int(bool(include_original)), n, num_samples,
                                        N.ptr(s_bins), N.ptr(t_bins), N.ptr(inds), N.stream()), "pdf_resample")
    return (s_bins, t_bins, inds) if return_indices else (s_bins, t_bins)


class _DistanceGradientScaleFn(torch.autograd.Function):
    """scale_gradients_by_distance_squared (model_components/losses.py:534-569): identity forward; the gradient is multiplied
    by clamp(((start + end) / 2)^2, 0, 1) per sample (nsamd_distance_gradient_scale)."""

    @staticmethod
    def forward(ctx, density: Tensor, rgb: Tensor, t_bins: Tensor):
        ctx.save_for_backward(t_bins)
        return density.view_as(density), rgb.view_as(rgb)

    @staticmethod
    def backward(ctx, g_density, g_rgb):
        (t_bins,) = ctx.saved_tensors
        n, s1 = t_bins.shape
        gd = _f32c(g_density).clone() if g_density is not None else None
        gr = _f32c(g_rgb).clone() if g_rgb is not None else None
        N.check(N.load().nsamd_distance_gradient_scale(N.ptr(t_bins), n, s1 - 1, N.ptr(gd), N.ptr(gr), N.stream()),
                "distance_gradient_scale")
        return gd, gr, None


def scale_gradients_by_distance_squared(density: Tensor, rgb: Tensor, t_bins: Tensor) -> Tuple[Tensor, Tensor]:
    """density `[N,S,1]`, rgb `[N,S,3]` of a ray batch with euclidean bin edges t_bins `[N,S+1]` -> the same values whose
    gradients are scaled by the squared distance (NerfactoModelConfig.use_gradient_scaling)."""
    N.require_cuda(density, rgb, t_bins)
    return _DistanceGradientScaleFn.apply(density, rgb, _f32c(t_bins))


# ---------------------------------------------------------------------------------------------------------------
# a16-a18  compositing
# ---------------------------------------------------------------------------------------------------------------
def _bg_args(background, device):
    """-> (mode, ctypes float[3] or None)."""
    if background is None or (isinstance(background, str) and background in ("random", "none")):
        return N.BG_NONE, None
    if isinstance(background, str):
        if background == "last_sample":
            return N.BG_LAST_SAMPLE, None
        named = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0)}
        if background not in named:
            raise ValueError(f"unsupported background colour {background!r}")
        vals = named[background]
    else:
        vals = [float(v) for v in torch.as_tensor(background).reshape(3).tolist()]
    return N.BG_CONSTANT, (C.c_float * 3)(*vals)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb: Tensor, weights: Tensor, t_bins: Optional[Tensor], bg_mode: int, bg_vals, want_depth: bool):
        N.require_cuda(rgb, weights, t_bins)
        rgb, weights = _f32c(rgb), _f32c(weights)
        t_bins = _f32c(t_bins) if t_bins is not None else None
        n, s = weights.shape
        dev = weights.device
        out = torch.empty((n, 3), device=dev, dtype=torch.float32)
        acc = torch.empty((n,), device=dev, dtype=torch.float32)
        depth = torch.empty((n,), device=dev, dtype=torch.float32) if want_depth else None
        ws = torch.empty((2 + 2 * ((n + 3) // 4),), device=dev, dtype=torch.float32) if want_depth else None
        N.check(N.load().nsamd_composite_fwd(N.ptr(rgb), N.ptr(weights), N.ptr(t_bins) if want_depth else None, n, s,
                                             bg_mode, bg_vals, 0, N.ptr(out), N.ptr(acc), N.ptr(depth), None, None,
                                             N.ptr(ws), N.stream()), "composite_fwd")
        ctx.save_for_backward(rgb, weights, t_bins if want_depth else None, ws)
        ctx.bg_mode, ctx.bg_vals, ctx.want_depth = bg_mode, bg_vals, want_depth
        if want_depth:
            return out, acc, depth
        return out, acc, None

    @staticmethod
    def backward(ctx, g_out, g_acc, g_depth):
        rgb, weights, t_bins, ws = ctx.saved_tensors
        n, s = weights.shape
        g_out = _f32c(g_out) if g_out is not None else None
        g_acc = _f32c(g_acc) if g_acc is not None else None
        g_depth = _f32c(g_depth) if (g_depth is not None and ctx.want_depth) else None
        d_rgb = torch.empty_like(rgb)
        d_w = torch.empty_like(weights)
        N.check(N.load().nsamd_composite_bwd(N.ptr(rgb), N.ptr(weights), N.ptr(t_bins), n, s, ctx.bg_mode, ctx.bg_vals,
                                             N.ptr(g_out), N.ptr(g_acc), N.ptr(g_depth), N.ptr(ws), None, N.ptr(d_rgb),
                                             N.ptr(d_w), N.stream()), "composite_bwd")
        return d_rgb, d_w, None, None, None, None


def composite(rgb: Tensor, weights: Tensor, t_bins: Optional[Tensor] = None, background="last_sample",
              expected_depth: bool = True):
    """Training-mode RGB + accumulation (+ expected depth) in one launch, differentiable w.r.t. rgb and weights.
    rgb `[N,S,3]`, weights `[N,S]`, t_bins `[N,S+1]` -> (rgb `[N,3]`, accumulation `[N]`, depth `[N]` or None)."""
    mode, vals = _bg_args(background, weights.device)
    return _CompositeFn.apply(rgb, weights, t_bins, mode, vals, bool(expected_depth and t_bins is not None))


@torch.no_grad()
def composite_eval(rgb: Tensor, weights: Tensor, t_bins: Tensor, background="last_sample"):
    """Eval-mode compositing (nan_to_num on the samples, clamp to [0,1]; renderers.py:225-231):
    -> rgb `[N,3]`, accumulation `[N]`, expected depth `[N]`, median depth `[N]`."""
    N.require_cuda(rgb, weights, t_bins)
    rgb, weights, t_bins = _f32c(rgb), _f32c(weights), _f32c(t_bins)
    n, s = weights.shape
    dev = weights.device
    mode, vals = _bg_args(background, dev)
    out = torch.empty((n, 3), device=dev, dtype=torch.float32)
    acc = torch.empty((n,), device=dev, dtype=torch.float32)
    dexp = torch.empty((n,), device=dev, dtype=torch.float32)
    dmed = torch.empty((n,), device=dev, dtype=torch.float32)
    ws = torch.empty((2 + 2 * ((n + 3) // 4),), device=dev, dtype=torch.float32)
    N.check(N.load().nsamd_composite_fwd(N.ptr(rgb), N.ptr(weights), N.ptr(t_bins), n, s, mode, vals, 1, N.ptr(out),
                                         N.ptr(acc), N.ptr(dexp), N.ptr(dmed), None, N.ptr(ws), N.stream()),
            "composite_fwd")
    return out, acc, dexp, dmed


@torch.no_grad()
def depth_median(weights: Tensor, t_bins: Tensor, return_index: bool = False):
    """DepthRenderer(method="median") (renderers.py:354-364) -> `[N]` (and the int32 sample index)."""
    N.require_cuda(weights, t_bins)
    weights, t_bins = _f32c(weights.detach()), _f32c(t_bins)
    n, s = weights.shape
    d = torch.empty((n,), device=weights.device, dtype=torch.float32)
    idx = torch.empty((n,), device=weights.device, dtype=torch.int32) if return_index else None
    N.check(N.load().nsamd_composite_fwd(None, N.ptr(weights), N.ptr(t_bins), n, s, N.BG_NONE, None, 0, None, None,
                                         None, N.ptr(d), N.ptr(idx), None, N.stream()), "composite_fwd(median)")
    return (d, idx) if return_index else d


@torch.no_grad()
def accumulation(weights: Tensor) -> Tensor:
    """AccumulationRenderer (renderers.py:293-317) without gradient -> `[N]` (use `composite` when training)."""
    N.require_cuda(weights)
    weights = _f32c(weights)
    n, s = weights.shape
    acc = torch.empty((n,), device=weights.device, dtype=torch.float32)
    N.check(N.load().nsamd_composite_fwd(None, N.ptr(weights), None, n, s, N.BG_NONE, None, 0, None, N.ptr(acc), None,
                                         None, None, None, N.stream()), "composite_fwd(acc)")
    return acc


# ---------------------------------------------------------------------------------------------------------------
# a19  proposal losses
# ---------------------------------------------------------------------------------------------------------------
class _InterlevelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_prop: Tensor, s_bins_prop: Tensor, w_fine: Tensor, s_bins_fine: Tensor):
        N.require_cuda(w_prop, s_bins_prop, w_fine, s_bins_fine)
        w_prop, s_bins_prop = _f32c(w_prop), _f32c(s_bins_prop)
        w_fine, s_bins_fine = _f32c(w_fine.detach()), _f32c(s_bins_fine.detach())
        n, sp = w_prop.shape
        sf = w_fine.shape[1]
        per_ray = torch.empty((n,), device=w_prop.device, dtype=torch.float32)
        dw = torch.empty_like(w_prop)
        N.check(N.load().nsamd_interlevel_loss(N.ptr(s_bins_fine), N.ptr(w_fine), sf, N.ptr(s_bins_prop), N.ptr(w_prop),
                                               sp, n, 1.0 / (n * sf), N.ptr(per_ray), N.ptr(dw), N.stream()),
                "interlevel_loss")
        ctx.save_for_backward(dw)
        return per_ray.sum() / (n * sf)  # torch.mean over [N, S_fine]  (losses.py:128)

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return dw * g, None, None, None


def interlevel_loss(weights_list: Sequence[Tensor], s_bins_list: Sequence[Tensor]) -> Tensor:
    """losses.py:113-131 (weights `[N,S_i]`, spacing bins `[N,S_i+1]`, last entry = the nerf level, detached)."""
    total = None
    for w, b in zip(weights_list[:-1], s_bins_list[:-1]):
        term = _InterlevelFn.apply(w, b, weights_list[-1], s_bins_list[-1])
        total = term if total is None else total + term
    return total


class _DistortionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights: Tensor, s_bins: Tensor):
        N.require_cuda(weights, s_bins)
        weights, s_bins = _f32c(weights), _f32c(s_bins)
        n, s = weights.shape
        per_ray = torch.empty((n,), device=weights.device, dtype=torch.float32)
        dw = torch.empty_like(weights)
        N.check(N.load().nsamd_distortion_loss(N.ptr(s_bins), N.ptr(weights), s, n, 1.0 / n, N.ptr(per_ray), N.ptr(dw),
                                               N.stream()), "distortion_loss")
        ctx.save_for_backward(dw)
        return per_ray.sum() / n  # torch.mean over rays (losses.py:153)

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return dw * g, None


def distortion_loss(weights: Tensor, s_bins: Tensor) -> Tensor:
    """losses.py:149-154 on the final level."""
    return _DistortionFn.apply(weights, s_bins)


# ---------------------------------------------------------------------------------------------------------------
# a3  camera-pose corrections of the rays
# ---------------------------------------------------------------------------------------------------------------
CAMERA_MODES = {"SO3xR3": 1, "SE3": 2}


class _CameraRaysFn(torch.autograd.Function):
    """CameraOptimizer.apply_to_raybundle (cameras/camera_optimizers.py:148-153) on the pose parameter: one launch forward
    (nsamd_camera_apply), one launch backward (nsamd_camera_backward: per-camera sums of dL/d(origins, directions) in double
    in a fixed order, then the exponential map's closed-form backward) instead of the index / exp-map / bmm chain of torch
    kernels and their autograd nodes. The regulariser (:179-185) stays with the caller's loss dict."""

    @staticmethod
    def forward(ctx, pose: Tensor, origins: Tensor, directions: Tensor, cams: Tensor, mode: int):
        N.require_cuda(pose, origins, directions, cams)
        raw_o, raw_d = _f32c(origins.reshape(-1, 3)), _f32c(directions.reshape(-1, 3))
        idx = cams.reshape(-1).contiguous().to(torch.int64)
        p = _f32c(pose)
        n = raw_o.shape[0]
        o, d = torch.empty_like(raw_o), torch.empty_like(raw_d)
        N.check(N.load().nsamd_camera_apply(N.ptr(p), mode, p.shape[0], N.ptr(raw_o), N.ptr(raw_d), N.ptr(idx), n, N.ptr(o),
                                            N.ptr(d), N.stream()), "camera_apply")
        ctx.save_for_backward(p, raw_d, idx)
        ctx.mode = mode
        ctx.pose_param = pose
        return o.reshape(origins.shape), d.reshape(directions.shape)

    @staticmethod
    def backward(ctx, g_o: Optional[Tensor], g_d: Optional[Tensor]):
        p, raw_d, idx = ctx.saved_tensors
        n = raw_d.shape[0]
        g_o = torch.zeros_like(raw_d) if g_o is None else _f32c(g_o.reshape(-1, 3))
        g_d = torch.zeros_like(raw_d) if g_d is None else _f32c(g_d.reshape(-1, 3))
        buf, ret = _grad_target(ctx.pose_param, True)
        up = N.RayGrads()
        up.d_origins[0], up.d_directions[0], up.count = N.ptr(g_o), N.ptr(g_d), 1
        N.check(N.load().nsamd_camera_backward(N.ptr(p), ctx.mode, p.shape[0], N.ptr(raw_d), N.ptr(idx), n, up, 0.0, 0.0,
                                               N.ptr(buf), None, N.stream()), "camera_backward")
        return ret, None, None, None, None


def camera_correct_rays(pose: Tensor, mode: str, origins: Tensor, directions: Tensor, camera_indices: Tensor):
    """origins + t(c), R(c) directions for the camera c of every ray; `pose` [num_cameras, 6] = (translation, rotation
    vector), mode "SO3xR3" / "SE3" (cameras/lie_groups.py:25-117). The rays themselves are treated as constants."""
    return _CameraRaysFn.apply(pose, origins, directions, camera_indices, CAMERA_MODES[mode])


# ---------------------------------------------------------------------------------------------------------------
# a1  pinhole ray generation
# ---------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def raygen_pinhole(ray_indices: Tensor, c2w: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor):
    """RayGenerator.forward for perspective cameras (ray_generators.py:41-56; cameras.py:598-634, 781-787, 887-909).
    -> origins `[N,3]`, directions `[N,3]`, pixel_area `[N,1]`, directions_norm `[N,1]`."""
    N.require_cuda(ray_indices, c2w, fx, fy, cx, cy)
    idx = ray_indices.contiguous().to(torch.int64)
    c2w = _f32c(c2w.reshape(-1, 3, 4))
    fx, fy, cx, cy = (_f32c(t.reshape(-1)) for t in (fx, fy, cx, cy))
    n = idx.shape[0]
    dev = idx.device
    o = torch.empty((n, 3), device=dev, dtype=torch.float32)
    d = torch.empty((n, 3), device=dev, dtype=torch.float32)
    pa = torch.empty((n, 1), device=dev, dtype=torch.float32)
    dn = torch.empty((n, 1), device=dev, dtype=torch.float32)
    N.check(N.load().nsamd_raygen_pinhole(N.ptr(idx), N.ptr(c2w), N.ptr(fx), N.ptr(fy), N.ptr(cx), N.ptr(cy), n,
                                          c2w.shape[0], N.ptr(o), N.ptr(d), N.ptr(pa), N.ptr(dn), N.stream()),
            "raygen_pinhole")
    return o, d, pa, dn


@torch.no_grad()
def adam_hyper(step: int, lr: float, betas=(0.9, 0.999)) -> Tuple[float, float]:
    """(lr / (1 - b1^step), sqrt(1 - b2^step)) — the two step-dependent scalars of Adam, in double as torch/optim/adam.py
    evaluates them (step_size, bias_correction2_sqrt); they are rounded to fp32 once, when they reach the kernel."""
    return lr / (1.0 - betas[0] ** step), (1.0 - betas[1] ** step) ** 0.5


def adam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-15, grad_scale: float = 1.0, hyper_dev: Optional[Tensor] = None) -> None:
    """torch.optim.Adam step over a flat fp32 arena (engine/optimizers.py:74-193; AdamOptimizerConfig eps=1e-15)."""
    N.require_cuda(params, grads, exp_avg, exp_avg_sq)
    N.check(N.load().nsamd_adam_step(N.ptr(params), N.ptr(grads), N.ptr(exp_avg), N.ptr(exp_avg_sq), params.numel(),
                                     float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                                     float(grad_scale), N.ptr(hyper_dev), N.stream()), "adam_step")


# ---------------------------------------------------------------------------------------------------------------
# a21 / f4  packed-sample path of instant-ngp (csrc/packed.hip): occupancy-grid marching, packed transmittance scan with
# early termination, compaction, packed compositing            (ray_samplers.py:385-519, models/instant_ngp.py:172-217)
# ---------------------------------------------------------------------------------------------------------------
def _occgrid_native(binaries: Tensor, roi_aabb: Sequence[float], coarse: Optional[Tensor] = None) -> N.OccGrid:
    g = N.OccGrid()
    g.binaries = N.ptr(binaries)
    g.levels, g.resolution = int(binaries.shape[0]), int(binaries.shape[1])
    for i, v in enumerate(roi_aabb):
        g.aabb[i] = float(v)
    g.coarse = N.ptr(coarse) if coarse is not None and coarse.numel() else None
    return g


def occgrid_coarse_words(levels: int, resolution: int) -> int:
    """Words of the coarse (4x4x4-block) occupancy bitfield for a grid shape; 0 = the marcher takes none."""
    return int(N.load().nsamd_occgrid_coarse_words(int(levels), int(resolution)))


@torch.no_grad()
def occgrid_cell_positions(cells: Optional[Tensor], num: int, binaries: Tensor, roi_aabb: Sequence[float], jitter: Tensor) -> Tensor:
    """Positions `[num,3]` inside the grid cells `cells` (flat int64 indices; None = cells 0..num-1) at the fractional
    offsets `jitter [num,3]` (nsamd_occgrid_cell_positions)."""
    N.require_cuda(binaries, jitter, cells)
    x = torch.empty((num, 3), device=jitter.device, dtype=torch.float32)
    N.check(N.load().nsamd_occgrid_cell_positions(N.ptr(cells), num, _occgrid_native(binaries, roi_aabb), N.ptr(_f32c(jitter)),
                                                  N.ptr(x), N.stream()), "occgrid_cell_positions")
    return x


@torch.no_grad()
def occgrid_update(occs: Tensor, cells: Optional[Tensor], occ_new: Tensor, ema_decay: float, scratch: Tensor) -> None:
    """occs[c] = max(occs[c] * decay, new estimates of c) for the listed cells, in place (nsamd_occgrid_update)."""
    N.require_cuda(occs, occ_new, scratch, cells)
    assert scratch.numel() >= occs.numel() and occs.is_contiguous()
    N.check(N.load().nsamd_occgrid_update(N.ptr(occs), occs.numel(), N.ptr(cells), N.ptr(_f32c(occ_new.reshape(-1))),
                                          occ_new.numel(), float(ema_decay), N.ptr(scratch), N.stream()), "occgrid_update")


@torch.no_grad()
def occgrid_binarise(occs: Tensor, binaries: Tensor, coarse: Optional[Tensor], occ_thre: float, scratch: Tensor,
                     threshold_out: Optional[Tensor] = None) -> None:
    """binaries = occs > min(mean(occs), occ_thre) and its coarse bitfield (nsamd_occgrid_binarise). scratch: >= 1024
    float64; threshold_out: optional `[2]` fp32 (threshold, mean)."""
    N.require_cuda(occs, binaries, scratch, coarse, threshold_out)
    assert scratch.dtype == torch.float64 and scratch.numel() >= 1024
    N.check(N.load().nsamd_occgrid_binarise(N.ptr(occs), int(binaries.shape[0]), int(binaries.shape[1]), float(occ_thre),
                                            N.ptr(binaries), N.ptr(coarse) if coarse is not None and coarse.numel() else None,
                                            N.ptr(scratch), N.ptr(threshold_out), N.stream()), "occgrid_binarise")


@torch.no_grad()
def packed_info_from_counts(counts: Tensor) -> Tuple[Tensor, int]:
    """nerfacc.pack_info from per-ray counts (int32 `[N]`) -> (`[N,2]` int64 (start, count), total). The total is read
    back to the host (the packed arrays are allocated to size, as nerfacc does)."""
    N.require_cuda(counts)
    n = counts.shape[0]
    info = torch.empty((n, 2), device=counts.device, dtype=torch.int64)
    total = torch.zeros((1,), device=counts.device, dtype=torch.int64)
    N.check(N.load().nsamd_packed_info(N.ptr(counts), n, N.ptr(info), N.ptr(total), N.stream()), "packed_info")
    return info, int(total.item())


@torch.no_grad()
def occgrid_march(origins: Tensor, directions: Tensor, binaries: Tensor, roi_aabb: Sequence[float], step_size: float,
                  near_plane: float = 0.0, far_plane: float = 1e10, t_min: Optional[Tensor] = None,
                  t_max: Optional[Tensor] = None, cone_angle: float = 0.0, jitter: Optional[Tensor] = None,
                  coarse: Optional[Tensor] = None, stash_cap: int = 128):
    """Ray marching through a multi-level occupancy grid (`binaries [levels,R,R,R]` uint8) -> (ray_indices int64 `[n]`,
    t_starts, t_ends fp32 `[n]`, packed_info `[N,2]`). Count pass, prefix, write pass. `coarse`: the grid's 4x4x4-block
    bitfield (occgrid_binarise) — empty-space skipping, same samples. `stash_cap` > 0: the count pass leaves every ray's first
    `stash_cap` kept steps in a scratch and the write pass copies them (a ray is marched once; same values), 0: march twice."""
    N.require_cuda(origins, directions, binaries)
    o, d = _f32c(origins), _f32c(directions)
    assert binaries.dtype == torch.uint8 and binaries.is_contiguous() and binaries.dim() == 4
    n = o.shape[0]
    dev = o.device
    grid = _occgrid_native(binaries, roi_aabb, coarse)
    tmin = None if t_min is None else _f32c(t_min.reshape(-1))
    tmax = None if t_max is None else _f32c(t_max.reshape(-1))
    jit = None if jitter is None else _f32c(jitter.reshape(-1))
    counts = torch.empty((n,), device=dev, dtype=torch.int32)
    lib = N.load()
    far = min(float(far_plane), 3.0e38)
    stash = torch.empty((n, int(stash_cap), 2), device=dev, dtype=torch.float32) if stash_cap > 0 else None
    N.check(lib.nsamd_occgrid_march_count_stash(N.ptr(o), N.ptr(d), N.ptr(tmin), N.ptr(tmax), n, float(near_plane), far, grid,
                                                float(step_size), float(cone_angle), N.ptr(jit), N.ptr(counts), N.ptr(stash),
                                                int(stash_cap), N.stream()), "occgrid_march_count_stash")
    info, total = packed_info_from_counts(counts)
    ray_indices = torch.empty((total,), device=dev, dtype=torch.int64)
    t_starts = torch.empty((total,), device=dev, dtype=torch.float32)
    t_ends = torch.empty((total,), device=dev, dtype=torch.float32)
    if total:
        N.check(lib.nsamd_occgrid_march_write_stashed(N.ptr(o), N.ptr(d), N.ptr(tmin), N.ptr(tmax), n, float(near_plane), far,
                                                      grid, float(step_size), float(cone_angle), N.ptr(jit), N.ptr(info),
                                                      N.ptr(stash), int(stash_cap), N.ptr(ray_indices), N.ptr(t_starts),
                                                      N.ptr(t_ends), N.stream()), "occgrid_march_write_stashed")
    return ray_indices, t_starts, t_ends, info


@torch.no_grad()
def packed_positions(origins: Tensor, directions: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor) -> Tensor:
    """o[ray] + d[ray] (t_start + t_end) / 2 for packed samples -> `[n,3]`."""
    N.require_cuda(origins, directions, ray_indices, t_starts, t_ends)
    out = torch.empty((ray_indices.shape[0], 3), device=origins.device, dtype=torch.float32)
    N.check(N.load().nsamd_packed_positions(N.ptr(_f32c(origins)), N.ptr(_f32c(directions)), N.ptr(ray_indices),
                                            N.ptr(t_starts), N.ptr(t_ends), ray_indices.shape[0], N.ptr(out), N.stream()),
            "packed_positions")
    return out


@torch.no_grad()
def packed_visibility_compact(ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Tensor,
                              early_stop_eps: float = 1e-4, alpha_thre: float = 0.0):
    """nerfacc.render_visibility_from_density + masking (OccGridEstimator.sampling): keep the samples whose transmittance
    is still >= early_stop_eps and whose alpha reaches alpha_thre -> (ray_indices, t_starts, t_ends, packed_info, mask)."""
    N.require_cuda(ray_indices, t_starts, t_ends, sigmas, packed_info)
    n_rays = packed_info.shape[0]
    dev = t_starts.device
    mask = torch.empty((t_starts.shape[0],), device=dev, dtype=torch.uint8)
    kept = torch.empty((n_rays,), device=dev, dtype=torch.int32)
    lib = N.load()
    N.check(lib.nsamd_packed_visibility(N.ptr(t_starts), N.ptr(t_ends), N.ptr(_f32c(sigmas)), N.ptr(packed_info), n_rays,
                                        float(early_stop_eps), float(alpha_thre), N.ptr(mask), N.ptr(kept), N.stream()),
            "packed_visibility")
    info2, total = packed_info_from_counts(kept)
    ri = torch.empty((total,), device=dev, dtype=torch.int64)
    ts = torch.empty((total,), device=dev, dtype=torch.float32)
    te = torch.empty((total,), device=dev, dtype=torch.float32)
    if total:
        N.check(lib.nsamd_packed_compact(N.ptr(mask), N.ptr(packed_info), N.ptr(info2), n_rays, N.ptr(t_starts), N.ptr(t_ends),
                                         N.ptr(ri), N.ptr(ts), N.ptr(te), N.stream()), "packed_compact")
    return ri, ts, te, info2, mask


class _PackedWeightsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas: Tensor, t_starts: Tensor, t_ends: Tensor, packed_info: Tensor):
        N.require_cuda(sigmas, t_starts, t_ends, packed_info)
        sigmas = _f32c(sigmas)
        w = torch.empty_like(sigmas)
        N.check(N.load().nsamd_packed_weights_fwd(N.ptr(t_starts), N.ptr(t_ends), N.ptr(sigmas), N.ptr(packed_info),
                                                  packed_info.shape[0], N.ptr(w), None, N.stream()), "packed_weights_fwd")
        ctx.save_for_backward(sigmas, t_starts, t_ends, packed_info)
        return w

    @staticmethod
    def backward(ctx, gw: Tensor):
        sigmas, t_starts, t_ends, info = ctx.saved_tensors
        ds = torch.empty_like(sigmas)
        N.check(N.load().nsamd_packed_weights_bwd(N.ptr(t_starts), N.ptr(t_ends), N.ptr(sigmas), N.ptr(_f32c(gw)), N.ptr(info),
                                                  info.shape[0], N.ptr(ds), N.stream()), "packed_weights_bwd")
        return ds, None, None, None


def packed_weights(sigmas: Tensor, t_starts: Tensor, t_ends: Tensor, packed_info: Tensor) -> Tensor:
    """nerfacc.render_weight_from_density(...)[0] on packed samples `[n]` (differentiable w.r.t. sigmas)."""
    return _PackedWeightsFn.apply(sigmas, t_starts, t_ends, packed_info)


def _packed_bg(background) -> Tuple[int, Optional[object]]:
    if isinstance(background, str):
        if background == "last_sample":
            raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")  # renderers.py:95-96
        if background == "random":
            return 0, None
        vals = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0)}[background]
    else:
        vals = tuple(float(v) for v in background.reshape(-1)[:3].tolist())
    return 1, (C.c_float * 3)(*vals)


class _PackedCompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb: Tensor, weights: Tensor, ray_indices: Tensor, packed_info: Tensor, t_starts, t_ends, bg_mode: int,
                bg_vals, eval_mode: bool):
        N.require_cuda(rgb, weights, ray_indices, packed_info)
        rgb, weights = _f32c(rgb), _f32c(weights)
        n_rays = packed_info.shape[0]
        dev = rgb.device
        out = torch.empty((n_rays, 3), device=dev, dtype=torch.float32)
        acc = torch.empty((n_rays,), device=dev, dtype=torch.float32)
        depth = torch.empty((n_rays,), device=dev, dtype=torch.float32) if t_starts is not None else None
        N.check(N.load().nsamd_packed_composite_fwd(N.ptr(rgb), N.ptr(weights), N.ptr(t_starts), N.ptr(t_ends),
                                                    N.ptr(packed_info), n_rays, bg_mode, bg_vals, 1 if eval_mode else 0,
                                                    N.ptr(out), N.ptr(acc), N.ptr(depth), N.stream()), "packed_composite_fwd")
        ctx.save_for_backward(rgb, weights, ray_indices)
        ctx.bg = (bg_mode, bg_vals)
        ctx.mark_non_differentiable(*([depth] if depth is not None else []))
        return out, acc, depth

    @staticmethod
    def backward(ctx, g_rgb, g_acc, _g_depth):
        rgb, weights, ray_indices = ctx.saved_tensors
        n = weights.shape[0]
        d_rgb = torch.empty_like(rgb) if ctx.needs_input_grad[0] else None
        d_w = torch.empty_like(weights)
        g_rgb = _f32c(g_rgb) if g_rgb is not None else torch.zeros((int(ray_indices.max()) + 1 if n else 0, 3), device=rgb.device)
        N.check(N.load().nsamd_packed_composite_bwd(N.ptr(rgb), N.ptr(weights), N.ptr(ray_indices), n, ctx.bg[0], ctx.bg[1],
                                                    N.ptr(g_rgb), N.ptr(_f32c(g_acc) if g_acc is not None else None),
                                                    N.ptr(d_rgb), N.ptr(d_w), N.stream()), "packed_composite_bwd")
        return d_rgb, d_w, None, None, None, None, None, None, None


def packed_composite(rgb: Tensor, weights: Tensor, ray_indices: Tensor, packed_info: Tensor, t_starts: Optional[Tensor] = None,
                     t_ends: Optional[Tensor] = None, background="random", eval_mode: bool = False):
    """accumulate_along_rays compositing of packed samples -> (rgb `[N,3]`, accumulation `[N]`, depth `[N]` or None)."""
    mode, vals = _packed_bg(background)
    return _PackedCompositeFn.apply(rgb, weights, ray_indices, packed_info, t_starts, t_ends, mode, vals, bool(eval_mode))
