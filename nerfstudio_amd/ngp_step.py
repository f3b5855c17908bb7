"""One instant-ngp training iteration as an explicit kernel schedule over capacity-sized buffers (BASELINE configs[3],
SURVEY.md §8 a21 / f4) — what train_step.NerfactoTrainStep is to the nerfacto path.

Reference call sequence (models/instant_ngp.py:172-217 `get_outputs`, :233-250 `get_loss_dict`; sampler
model_components/ray_samplers.py:437-519 -> nerfacc OccGridEstimator.sampling; pipelines/base_pipeline.py:290-303):

    march through the occupancy grid  ->  density of the candidates (field.density_fn)  ->  visibility scan + compaction
    ->  NerfactoField on the survivors  ->  packed weights  ->  rgb / accumulation / expected depth  ->  MSE against the
    target blended with the batch's random background  ->  backward  ->  Adam

Through this package's nn.Module / autograd classes that is the same kernels plus ~40 torch glue ops (index gathers for
the frustums, elementwise position arithmetic, allocations of every intermediate, bincount, the loss) between TWO host
reads of a sample count, and the step is host-bound: 1.52 ms against 0.85 ms of kernel time (profiles/r03_final_bench_ngp*).
Here every intermediate lives in a buffer sized for a CAPACITY of candidates / kept samples (grown 1.5 x when a batch exceeds
it — the only allocation after warm-up), the kernels are launched back to back through the C ABI, and the two counts are the
only host reads (the packed arrays are data-dependent in size, as in nerfacc; the kernels take their sizes by value).

Same kernels, same order as the module path. What differs is rounding only: the positions of the kept samples come from
nsamd_packed_positions (an fma) instead of the torch expression `o + d * (t0 + t1) / 2`, and the gradient of the MSE is
written out by hand. tests/test_gpu_packed.py compares a step of the two routes (outputs, loss, every gradient).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _native as N
from . import functional as F
from .utils import profiler


# kept steps per ray the count pass of the marcher leaves behind for the write pass (nsamd_occgrid_march_count_stash): a ray
# that keeps more is marched a second time (the benchmark's rays keep ~42 of ~840 lattice steps)
STASH_CAP = 128
_TWO_PASS = os.environ.get("NSAMD_NGP_TWO_PASS", "0") == "1"


class NgpTrainStep:
    """forward() -> backward() on static buffers; the caller owns the optimiser (param.grad is accumulated into, the main
    table's gradient is WRITTEN — nsamd_hashgrid_encode_bwd_set — unless `accumulate_table`)."""

    def __init__(self, model, num_rays: int, device, cap_candidates: int = 0, cap_kept: int = 0) -> None:
        N.require_cuda(torch.empty(0, device=device))
        self.model, self.dev = model, device
        cfg = model.config
        self.cfg = cfg
        fld = model.field
        enc = fld.mlp_base.encoding
        self.grid = enc.spec
        self.L2 = enc.spec.out_dim
        if self.L2 != 32:
            raise RuntimeError("NgpTrainStep: the main-field kernels take 16 levels x 2 features")
        self.n, self.cap_n, self._ray = 0, 0, {}
        self._set_num_rays(int(num_rays))
        n = self.n
        self.totals = torch.zeros(2, device=device, dtype=torch.int64)
        self.totals_host = torch.zeros(2, dtype=torch.int64).pin_memory()
        self.loss_sum = torch.zeros(1, device=device)
        self.view0 = torch.zeros(1, 3, device=device)  # density_fn: no direction, a constant appearance row
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        self.app0 = torch.zeros(emb.shape[1], device=device) if emb is not None else None
        self.cap_c = self.cap_k = 0
        self._grow_candidates(int(cap_candidates) if cap_candidates > 0 else 64 * n)  # (grown on demand)
        self._grow_kept(int(cap_kept) if cap_kept > 0 else 32 * n)
        self.num_candidates = self.num_kept = 0
        # {id(parameter): gradient buffer} of a caller that keeps the gradients outside `param.grad` (arena.ParamArena.grad_lookup:
        # pipeline.NgpEngine under a trainer whose own optimiser must find `.grad` None); None: `param.grad`
        self.grad_lookup = None
        self.accumulate_table = False
        self.fuse_route = os.environ.get("NSAMD_NGP_FUSE_ROUTE", "1") == "1"  # the field backward emits the scatter's records (backward)
        self.has_bounds = False
        bgc = model.renderer_rgb.background_color
        if isinstance(bgc, str) and bgc == "last_sample":
            raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")  # renderers.py:95-96
        self.random_bg = isinstance(bgc, str) and bgc == "random"
        self.bg_mode, self.bg_vals = F._packed_bg(bgc)

    # ---- buffers -----------------------------------------------------------------------------------------------------
    # per-ray arrays: name -> (trailing shape, dtype). The attributes of these names are the first `n` rows of capacity-sized
    # buffers: DynamicBatchPipeline changes the number of rays of a batch every step (pipelines/dynamic_batch.py:71-95), and
    # a change within the capacity costs nothing (grown 1.5 x beyond it).
    _RAY_BUFFERS = (("origins", (3,), torch.float32), ("directions", (3,), torch.float32), ("target", (3,), torch.float32),
                    ("cams", (), torch.int64), ("jitter", (), torch.float32), ("counts", (), torch.int32),
                    ("t_min", (), torch.float32), ("t_max", (), torch.float32), ("stash", (STASH_CAP, 2), torch.float32),
                    ("kept", (), torch.int32), ("info", (2,), torch.int64), ("info2", (2,), torch.int64),
                    ("rgb", (3,), torch.float32), ("acc", (), torch.float32), ("depth", (), torch.float32),
                    ("bg", (3,), torch.float32), ("pred", (3,), torch.float32), ("g_rgb", (3,), torch.float32),
                    ("g_acc", (), torch.float32))

    def _set_num_rays(self, n: int) -> None:
        if n <= 0:
            raise ValueError("NgpTrainStep: a batch needs at least one ray")
        if n > self.cap_n:
            self.cap_n = max(n, int(1.5 * self.cap_n))
            self._ray = {name: torch.zeros((self.cap_n, *shape), device=self.dev, dtype=dtype) for name, shape, dtype in self._RAY_BUFFERS}
            self.n = 0
        if n != self.n:
            for name, _, _ in self._RAY_BUFFERS:
                setattr(self, name, self._ray[name][:n])
            self.n = n

    def _grow_candidates(self, cap: int) -> None:
        if cap <= self.cap_c:
            return
        dev, f32 = self.dev, torch.float32
        self.cap_c = cap
        self.c_ri = torch.empty(cap, device=dev, dtype=torch.int64)
        self.c_ts, self.c_te = torch.empty(cap, device=dev, dtype=f32), torch.empty(cap, device=dev, dtype=f32)
        self.c_pos = torch.empty(cap, 3, device=dev, dtype=f32)
        self.c_enc = torch.empty(self.L2 * cap, device=dev, dtype=f32)
        self.c_sel, self.c_sigma = torch.empty(cap, device=dev, dtype=f32), torch.empty(cap, device=dev, dtype=f32)
        self.c_rgb = torch.empty(cap, 3, device=dev, dtype=f32)
        self.c_mask = torch.empty(cap, device=dev, dtype=torch.uint8)

    def _grow_kept(self, cap: int) -> None:
        if cap <= self.cap_k:
            return
        dev, f32 = self.dev, torch.float32

        def b(*shape, dtype=f32):
            return torch.empty(shape, device=dev, dtype=dtype)

        self.cap_k = cap
        self.k_ri, self.k_cams = b(cap, dtype=torch.int64), b(cap, dtype=torch.int64)
        self.k_ts, self.k_te = b(cap), b(cap)
        self.k_pos, self.k_dirs = b(cap, 3), b(cap, 3)
        self.k_enc, self.k_denc = b(self.L2 * cap), b(self.L2 * cap)
        self.k_sel, self.k_dens, self.k_w = b(cap), b(cap), b(cap)
        self.k_rgb, self.k_drgb = b(cap, 3), b(cap, 3)
        self.k_dw, self.k_dsigma = b(cap), b(cap)
        self.k_mid = b(cap)

    def _read_total(self, slot: int) -> int:
        """The one host read of a sample count (pinned buffer, the launch stream's own copy + wait)."""
        self.totals_host[slot:slot + 1].copy_(self.totals[slot:slot + 1], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return int(self.totals_host[slot])

    # ---- batch ---------------------------------------------------------------------------------------------------------
    def set_batch(self, origins: Tensor, directions: Tensor, camera_indices: Optional[Tensor], target: Optional[Tensor] = None,
                  nears: Optional[Tensor] = None, fars: Optional[Tensor] = None) -> None:
        """A batch of any number of rays (see `_set_num_rays`). `nears` / `fars`: the per-ray interval a collider put on the
        bundle — VolumetricSampler hands them to the marcher as t_min / t_max (ray_samplers.py:470-476), as the module path
        does (this schedule used to march the global near / far planes whatever the bundle carried)."""
        self._set_num_rays(int(origins.reshape(-1, 3).shape[0]))
        self.has_bounds = nears is not None and fars is not None
        if self.has_bounds:
            self.t_min.copy_(nears.reshape(-1))
            self.t_max.copy_(fars.reshape(-1))
        self.origins.copy_(origins.reshape(-1, 3))
        self.directions.copy_(directions.reshape(-1, 3))
        if camera_indices is not None:
            self.cams.copy_(camera_indices.reshape(-1))
        elif self.app0 is not None and self.model.training:
            raise AttributeError("Camera indices are not provided.")  # fields/nerfacto_field.py:240-241
        if target is not None:
            self.target.copy_(target.reshape(-1, 3))

    def _field_mlp(self) -> N.FieldMlp:
        fld = self.model.field
        params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        return N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(emb), emb.shape[0] if emb is not None else 0,
                          float(fld.average_init_density))

    # ---- forward -------------------------------------------------------------------------------------------------------
    @profiler.time_function
    @torch.no_grad()
    def forward(self, jitter: Optional[Tensor] = None) -> None:
        """Sampling + field + compositing. `jitter [n]`: injected lattice offsets (tests); None draws them on the device."""
        m, cfg, n = self.model, self.cfg, self.n
        lib, st, ck = N.load(), N.stream(), N.check
        grid = m.occupancy_grid
        grid.ensure_derived()
        og = F._occgrid_native(grid.binaries, grid._roi, grid._coarse)
        if jitter is None:
            self.jitter.uniform_()  # stratified training (ray_samplers.py:489); same draws as torch.rand(n)
        else:
            self.jitter.copy_(jitter.reshape(-1))
        o, d = N.ptr(self.origins), N.ptr(self.directions)
        near, far, step, cone = float(cfg.near_plane), min(float(cfg.far_plane), 3.0e38), float(cfg.render_step_size), float(cfg.cone_angle)
        # -- candidates: count -> prefix -> (host read) -> write
        tmin, tmax = (N.ptr(self.t_min), N.ptr(self.t_max)) if self.has_bounds else (None, None)
        stash = None if _TWO_PASS else N.ptr(self.stash)  # (NSAMD_NGP_TWO_PASS=1: march twice, full head on the candidates — A/B)
        ck(lib.nsamd_occgrid_march_count_stash(o, d, tmin, tmax, n, near, far, og, step, cone, N.ptr(self.jitter), N.ptr(self.counts),
                                               stash, 0 if _TWO_PASS else STASH_CAP, st), "occgrid_march_count_stash")
        ck(lib.nsamd_packed_info(N.ptr(self.counts), n, N.ptr(self.info), N.ptr(self.totals[0:1]), st), "packed_info")
        mc = self._read_total(0)
        self.num_candidates = mc
        fld = m.field
        enc = fld.mlp_base.encoding
        table = enc.hash_table
        fm = self._field_mlp()
        mk = 0
        if mc:
            if mc > self.cap_c:
                self._grow_candidates(int(1.5 * mc))
            ck(lib.nsamd_occgrid_march_write_stashed(o, d, tmin, tmax, n, near, far, og, step, cone, N.ptr(self.jitter), N.ptr(self.info),
                                                     stash, 0 if _TWO_PASS else STASH_CAP, N.ptr(self.c_ri), N.ptr(self.c_ts), N.ptr(self.c_te), st),
               "occgrid_march_write_stashed")
            # -- sigma_fn (ray_samplers.py:420-429): density of the candidates; no direction, a constant appearance row
            ck(lib.nsamd_packed_positions(o, d, N.ptr(self.c_ri), N.ptr(self.c_ts), N.ptr(self.c_te), mc, N.ptr(self.c_pos), st),
               "packed_positions")
            ck(lib.nsamd_hashgrid_encode_fwd(N.make_points(positions=self.c_pos), mc, fld._transform, fld._box, N.ptr(table),
                                             self.grid.native(), N.ptr(self.c_enc), 1, mc, N.ptr(self.c_sel), st), "hashgrid_encode_fwd")
            ck(lib.nsamd_field_mlp_fwd(N.ptr(self.c_enc), N.ptr(self.c_sel), N.ptr(self.view0), None, N.ptr(self.app0), mc, mc, fm,
                                       N.ptr(self.c_sigma), N.ptr(self.c_rgb) if _TWO_PASS else None, st), "field_mlp_fwd")  # (rgb NULL: density only)
            # -- visibility-ordered early termination + alpha threshold, then compaction (OccGridEstimator.sampling)
            alpha = float(cfg.alpha_thre)
            if alpha > 0.0:
                alpha = min(alpha, grid._occ_mean)
            ck(lib.nsamd_packed_visibility(N.ptr(self.c_ts), N.ptr(self.c_te), N.ptr(self.c_sigma), N.ptr(self.info), n, 1e-4, alpha,
                                           N.ptr(self.c_mask), N.ptr(self.kept), st), "packed_visibility")
            ck(lib.nsamd_packed_info(N.ptr(self.kept), n, N.ptr(self.info2), N.ptr(self.totals[1:2]), st), "packed_info")
            mk = self._read_total(1)
            if mk:
                if mk > self.cap_k:
                    self._grow_kept(int(1.5 * mk))
                ck(lib.nsamd_packed_compact(N.ptr(self.c_mask), N.ptr(self.info), N.ptr(self.info2), n, N.ptr(self.c_ts),
                                            N.ptr(self.c_te), N.ptr(self.k_ri), N.ptr(self.k_ts), N.ptr(self.k_te), st), "packed_compact")
        if mk == 0:
            # a single fake sample (ray 0, [1, 1]) keeps every downstream shape valid (ray_samplers.py:494-500)
            mk = 1
            self.k_ri[:1].zero_()
            self.k_ts[:1].fill_(1.0)
            self.k_te[:1].fill_(1.0)
            self.info2.zero_()
            self.info2[0, 1] = 1
        self.num_kept = mk
        grid.last_packed_info = self.info2
        # -- the field on the survivors: per-sample direction and camera
        torch.index_select(self.directions, 0, self.k_ri[:mk], out=self.k_dirs[:mk])
        train_app = self.app0 is not None and m.training
        if train_app:
            torch.index_select(self.cams, 0, self.k_ri[:mk], out=self.k_cams[:mk])
        ck(lib.nsamd_packed_positions(o, d, N.ptr(self.k_ri), N.ptr(self.k_ts), N.ptr(self.k_te), mk, N.ptr(self.k_pos), st),
           "packed_positions")
        ck(lib.nsamd_hashgrid_encode_fwd(N.make_points(positions=self.k_pos), mk, fld._transform, fld._box, N.ptr(table),
                                         self.grid.native(), N.ptr(self.k_enc), 1, mk, N.ptr(self.k_sel), st), "hashgrid_encode_fwd")
        self._app_const = None
        if self.app0 is not None and not train_app:  # eval semantics of the embedding (nerfacto_field.py:253-261)
            emb = fld.embedding_appearance.embedding.weight
            self._app_const = (emb.mean(dim=0) if fld.use_average_appearance_embedding else torch.zeros_like(emb[0])).contiguous()
        ck(lib.nsamd_field_mlp_fwd(N.ptr(self.k_enc), N.ptr(self.k_sel), N.ptr(self.k_dirs), N.ptr(self.k_cams) if train_app else None,
                                   N.ptr(self._app_const), 1, mk, fm, N.ptr(self.k_dens), N.ptr(self.k_rgb), st), "field_mlp_fwd")
        self._train_app = train_app
        # -- packed weights and the three renderers in one launch (models/instant_ngp.py:191-214)
        ck(lib.nsamd_packed_weights_fwd(N.ptr(self.k_ts), N.ptr(self.k_te), N.ptr(self.k_dens), N.ptr(self.info2), n, N.ptr(self.k_w),
                                        None, st), "packed_weights_fwd")
        ck(lib.nsamd_packed_composite_fwd(N.ptr(self.k_rgb), N.ptr(self.k_w), N.ptr(self.k_ts), N.ptr(self.k_te), N.ptr(self.info2), n,
                                          self.bg_mode, self.bg_vals, 0, N.ptr(self.rgb), N.ptr(self.acc), N.ptr(self.depth), st),
           "packed_composite_fwd")
        # expected depth clipped to the batch's range of sample midpoints (renderers.py:381-383)
        torch.add(self.k_ts[:mk], self.k_te[:mk], out=self.k_mid[:mk])
        self.k_mid[:mk].mul_(0.5)
        torch.clamp(self.depth, min=self.k_mid[:mk].min(), max=self.k_mid[:mk].max(), out=self.depth)

    def outputs(self) -> Dict[str, Tensor]:
        """The reference's output dict (models/instant_ngp.py:209-216), views of the static buffers."""
        return {"rgb": self.rgb, "accumulation": self.acc[:, None], "depth": self.depth[:, None],
                "num_samples_per_ray": self.info2[:, 1]}

    # ---- loss + backward -----------------------------------------------------------------------------------------------
    @profiler.time_function
    @torch.no_grad()
    def loss(self, background: Optional[Tensor] = None) -> Tensor:
        """MSE against the target blended as RGBRenderer.blend_background_for_loss_computation does (renderers.py:175-199):
        with background "random" a fresh colour per ray is added to the prediction as bg (1 - accumulation) (the target is
        RGB here: RGBA targets are blended by the caller). -> the loss (a 0-dim view of a static buffer); the gradients with
        respect to the rendered rgb / accumulation are left in g_rgb / g_acc."""
        n = self.n
        if self.random_bg:
            if background is None:
                self.bg.uniform_()  # torch.rand_like(pred_image)
            else:
                self.bg.copy_(background)
            torch.neg(self.acc, out=self.g_acc)
            self.g_acc.add_(1.0)                                    # (1 - accumulation)
            torch.mul(self.bg, self.g_acc[:, None], out=self.pred)
            self.pred.add_(self.rgb)                                # pred_image + background * (1 - accumulation)
        else:
            self.pred.copy_(self.rgb)
        self.loss_sum.zero_()
        N.check(N.load().nsamd_mse_loss(N.ptr(self.pred), N.ptr(self.target), 3 * n, 1.0 / (3 * n), N.ptr(self.loss_sum),
                                        N.ptr(self.g_rgb), N.stream()), "mse_loss")
        if self.random_bg:  # d pred / d accumulation = -bg
            torch.sum(self.g_rgb * self.bg, dim=-1, out=self.g_acc)
            self.g_acc.neg_()
        return self.loss_sum[0] / (3 * n)

    def prepare_grads(self) -> None:
        """Gradient buffers for every field parameter: fresh ones zero-filled (the kernels accumulate), except the main
        table's, which the scatter writes."""
        fld = self.model.field
        table = fld.mlp_base.encoding.hash_table
        if self.grad_lookup is not None:
            return
        for p in fld.parameters():
            if p.grad is None:
                p.grad = torch.empty_like(p) if (p is table and not self.accumulate_table) else torch.zeros_like(p)

    @profiler.time_function
    @torch.no_grad()
    def backward(self) -> None:
        m, n, mk = self.model, self.n, self.num_kept
        lib, st, ck = N.load(), N.stream(), N.check
        fld = m.field
        enc = fld.mlp_base.encoding
        table = enc.hash_table
        self.prepare_grads()
        ck(lib.nsamd_packed_composite_bwd(N.ptr(self.k_rgb), N.ptr(self.k_w), N.ptr(self.k_ri), mk, self.bg_mode, self.bg_vals,
                                          N.ptr(self.g_rgb), N.ptr(self.g_acc) if self.random_bg else None, N.ptr(self.k_drgb),
                                          N.ptr(self.k_dw), st), "packed_composite_bwd")
        ck(lib.nsamd_packed_weights_bwd(N.ptr(self.k_ts), N.ptr(self.k_te), N.ptr(self.k_dens), N.ptr(self.k_dw), N.ptr(self.info2), n,
                                        N.ptr(self.k_dsigma), st), "packed_weights_bwd")
        params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        gl = self.grad_lookup
        grad_of = (lambda p: gl[id(p)]) if gl is not None else (lambda p: p.grad)  # noqa: E731
        grads = N.FieldMlpGrads(*(N.ptr(grad_of(p)) for p in params), N.ptr(grad_of(emb)) if (emb is not None and self._train_app) else None)
        fws, fws_n = F.field_bwd_workspace(self.dev)
        write_only = not self.accumulate_table
        if write_only and self.fuse_route and self.grid.num_levels == 16:
            # as the nerfacto schedule does (train_step.backward_field_and_table): the field backward emits the table scatter's
            # pass-1 records from its registers — no `denc` round trip, no route launch over the kept samples
            # (NSAMD_NGP_FUSE_ROUTE=0: the two entry points, A/B)
            sws, sws_n = F._producer_scatter_workspace(self.grid, self.dev, mk)
            if sws is not None:
                ck(lib.nsamd_field_mlp_bwd_scatter(
                    N.make_points(positions=self.k_pos), fld._transform, fld._box, self.grid.native(), N.ptr(self.k_enc),
                    N.ptr(self.k_sel), N.ptr(self.k_dirs), N.ptr(self.k_cams) if self._train_app else None, N.ptr(self._app_const), 1,
                    mk, self._field_mlp(), N.ptr(self.k_dsigma), N.ptr(self.k_drgb), None, grads, N.ptr(fws), fws_n,
                    N.ptr(grad_of(table)), N.ptr(sws), sws_n, st), "field_mlp_bwd_scatter")
                return
        ck(lib.nsamd_field_mlp_bwd(N.ptr(self.k_enc), N.ptr(self.k_sel), N.ptr(self.k_dirs), N.ptr(self.k_cams) if self._train_app else None,
                                   N.ptr(self._app_const), 1, mk, self._field_mlp(), N.ptr(self.k_dsigma), N.ptr(self.k_drgb),
                                   N.ptr(self.k_denc), grads, N.ptr(fws), fws_n, st), "field_mlp_bwd")
        ws, ws_n = F._scatter_workspace(self.grid, self.dev, mk, write_only=write_only)
        fn = lib.nsamd_hashgrid_encode_bwd_set if write_only else lib.nsamd_hashgrid_encode_bwd
        ck(fn(N.make_points(positions=self.k_pos), mk, fld._transform, fld._box, N.ptr(table), self.grid.native(), N.ptr(self.k_denc), 1, mk,
              N.ptr(grad_of(table)), None, N.ptr(ws), ws_n, st), "hashgrid_encode_bwd")


# ---------------------------------------------------------------------------------------------------------------------
# behind the Model API (NGPModel.get_outputs / get_loss_dict), as fused_step.FusedTrainStep does for nerfacto
# ---------------------------------------------------------------------------------------------------------------------
class _NgpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor: Tensor, step: "NgpFusedStep", value: Tensor):  # noqa: D102
        ctx.step = step
        return value.clone()

    @staticmethod
    def backward(ctx, g):  # noqa: D102
        step: "NgpFusedStep" = ctx.step
        step.backward_calls += 1
        if step.checks_left > 0 or step.backward_calls % 64 == 0:  # the kernels hold d(rgb_loss): unit upstream gradient only
            step.checks_left = max(step.checks_left - 1, 0)
            if float(g) != 1.0:
                raise RuntimeError("NgpFusedStep: rgb_loss must reach backward() with unit weight; use the module path otherwise")
        step.runner.accumulate_table = step.model.field.mlp_base.encoding.hash_table.grad is not None
        step.runner.backward()
        return None, None, None


class NgpFusedStep:
    """NGPModel's training iteration on the explicit schedule behind the Model API (config.fused_train_step)."""

    def __init__(self, model) -> None:
        self.model, self.runner = model, None
        self.checks_left, self.backward_calls = 3, 0

    def get_outputs(self, ray_bundle, jitter: Optional[Tensor] = None) -> Dict[str, object]:
        o = ray_bundle.origins.reshape(-1, 3)
        if self.runner is None:
            self.runner = NgpTrainStep(self.model, o.shape[0], o.device)
        r = self.runner
        cams = ray_bundle.camera_indices
        r.set_batch(o, ray_bundle.directions.reshape(-1, 3), None if cams is None else cams.reshape(-1),
                    nears=getattr(ray_bundle, "nears", None), fars=getattr(ray_bundle, "fars", None))
        r.forward(jitter)
        out = r.outputs()
        out["ngp_step"] = self
        return out

    def get_loss_dict(self, outputs, batch) -> Dict[str, Tensor]:
        assert outputs.get("ngp_step") is self, "outputs of another forward"
        r = self.runner
        image = batch["image"].to(r.target.device)
        if image.shape[-1] == 4:  # RGBA: blended with the loss's own background, as renderers.py:175-199 does
            raise NotImplementedError("NgpFusedStep takes RGB targets; RGBA goes through the module path")
        r.target.copy_(image.reshape(-1, 3))
        value = r.loss()
        anchor = self.model.field.mlp_base.encoding.hash_table
        return {"rgb_loss": _NgpLoss.apply(anchor, self, value)}
