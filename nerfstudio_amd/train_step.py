"""One nerfacto training iteration as an explicit kernel schedule (no autograd graph, no torch arithmetic).

What the reference does per step (engine/trainer.py:487-531 -> pipelines/base_pipeline.py:290-303 ->
models/nerfacto.py:298-392): collider, ProposalNetworkSampler (3 levels), NerfactoField, weights, renderers, MSE +
interlevel + distortion losses, `backward()`, optimiser. `NerfactoModel` (nerfacto.py) reproduces that through the
module API with autograd; this runner issues THE SAME kernels in a fixed order over buffers allocated once:

  forward : piecewise_bins -> [hash_fwd -> density_mlp_fwd -> weights_fwd -> pdf_resample] x 2 proposal levels
            -> hash_fwd(main) -> field_mlp_fwd -> weights_fwd -> composite_fwd (+ median depths)
  backward: mse_loss, distortion_loss, interlevel_loss x 2 (value + gradient in one pass each)
            -> composite_bwd -> weights_bwd -> field_mlp_bwd -> hash_bwd(main)
            -> (only on proposal-update steps, ray_samplers.py:590) weights_bwd -> density_mlp_bwd -> hash_bwd x 2

~25 launches, every one a libnsamd kernel (round 5: batch selection + initial bins are one launch, the main field's
weight-gradient reduce rides the table scatter's apply pass; the fully merged forms of the per-ray stages were bit-identical and
measured slower — csrc/experiments/rounds2to5_opt_in_variants.patch); parameter gradients
accumulate directly into `param.grad` (the views of arena.ParamArena). Nothing here depends on host-side values that change from step to step (the jitter is drawn on the
device, the anneal exponent lives in device memory), so the whole iteration can be captured in a hipGraph once per
schedule variant and replayed. Numerically identical to the autograd path (tests/test_gpu_kernels.py compares them).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _native as N
from . import functional as F
from .nerfacto import NerfactoModel
from .utils import profiler


class NerfactoTrainStep:
    def __init__(self, model: NerfactoModel, num_rays: int, device, compute_depths: bool = True,
                 forward_only: bool = False) -> None:
        """forward_only: no gradient / saved-activation buffers (eval_render.EvalRenderer: at 32 768 rays per chunk they
        are ~1.5 GB an eval render never touches); `forward_backward` and the backward methods must not be called."""
        self.model = model
        self.forward_only = bool(forward_only)
        cfg = model.config
        self.cfg = cfg
        self.n = n = int(num_rays)
        self.counts = (*cfg.num_proposal_samples_per_ray, cfg.num_nerf_samples_per_ray)
        self.n_prop = len(cfg.num_proposal_samples_per_ray)
        self.compute_depths = compute_depths
        if cfg.background_color not in ("last_sample", "black", "white", "random"):
            raise ValueError(cfg.background_color)
        # use_single_jitter=False (ray_samplers.py:104-107, 318-322): one draw per bin EDGE instead of one per ray — the
        # per-level jitter buffers become [n, S+1] and the resampling goes through the unfused kernels
        self.single_jitter = bool(getattr(cfg, "use_single_jitter", True))
        self.gradient_scaling = bool(getattr(cfg, "use_gradient_scaling", False))
        f32 = dict(device=device, dtype=torch.float32)
        e = lambda *shape: torch.empty(shape, **f32)  # noqa: E731
        # (training-only buffers: one element each in a forward-only runner)
        t = (lambda *shape: torch.empty((1,) * len(shape), **f32)) if forward_only else e  # noqa: E731
        if cfg.background_color == "random":
            # the rendered colour carries no background; the loss blends `rand_like(pred) * (1 - accumulation)` into the
            # prediction (renderers.py:112-115, 194-196; models/nerfacto.py:377-381) — a per-ray colour the kernels read
            # (zero-initialised: callers that inject the jitter — `draw_jitter=False` — and no background of their own
            # composite against black, never against uninitialised memory)
            self.bg_mode, self.bg_vals, self.bg_rays = 3, None, torch.zeros((n, 3), **f32)
        else:
            self.bg_mode, self.bg_vals = F._bg_args(cfg.background_color, device)
            self.bg_rays = None
        # ---- per-step inputs (static addresses; the caller fills them) ----
        self.origins, self.directions = e(n, 3), e(n, 3)
        self.camera_indices = torch.zeros((n,), device=device, dtype=torch.int64)
        self.target = e(n, 3)
        self.nears = torch.full((n,), float(cfg.near_plane), **f32)  # NearFarCollider, training mode
        self.fars = torch.full((n,), float(cfg.far_plane), **f32)
        self.jitter = e(self.n_prop + 1, n)  # single jitter: one draw per level and ray
        self.jitter_edges = None if self.single_jitter else [e(n, s + 1) for s in self.counts]  # per edge: [n, S+1] per level
        self.anneal_dev = torch.ones((1,), **f32)
        # ---- sampler state ----
        self.s_bins = [e(n, s + 1) for s in self.counts]
        self.t_bins = [e(n, s + 1) for s in self.counts]
        self.weights = [e(n, s) for s in self.counts]
        # ---- fields ----
        self.props = list(model.proposal_networks) if not cfg.use_same_proposal_network else \
            [model.proposal_networks[0]] * self.n_prop
        self.p_enc, self.p_sel, self.p_pre, self.p_dens = [], [], [], []
        for lvl in range(self.n_prop):
            m = n * self.counts[lvl]
            self.p_enc.append(t(self.props[lvl].encoding.get_out_dim(), m))  # (forward-only: see `ensure_proposal_features`)
            self.p_sel.append(t(m))
            self.p_pre.append(t(m))
            self.p_dens.append(e(m))
        self.m_main = mm = n * self.counts[-1]
        fld = model.field
        self.f_enc, self.f_sel, self.f_dens, self.f_rgb = e(fld.mlp_base.encoding.get_out_dim(), mm), e(mm), e(mm), e(mm, 3)
        # ---- outputs ----
        self.rgb, self.acc, self.depth_exp = e(n, 3), e(n), e(n)
        self.minmax_ws = e(2 + 2 * ((n + 3) // 4))
        self.depth_med = [e(n) for _ in self.counts]
        # ---- losses / gradients ----
        self.sq_err = e(n)  # per-ray sum of squared rgb errors
        self.dist_per_ray = e(n)
        self.inter_per_ray = [e(n) for _ in range(self.n_prop)]
        self.d_rgb_out = e(n, 3)
        self.dw_dist = t(n, self.counts[-1])
        self.dw_prop = [t(n, self.counts[lvl]) for lvl in range(self.n_prop)]
        self.d_rgb_s, self.d_dens_main = t(mm, 3), t(mm)
        # host arrays of device pointers for nsamd_proposal_losses (the buffers are static, so built once)
        def parr(ts):
            return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

        self._pl_s_bins = parr(self.s_bins[: self.n_prop])
        self._pl_weights = parr(self.weights[: self.n_prop])
        self._pl_per_ray = parr(self.inter_per_ray)
        self._pl_dw = parr(self.dw_prop)
        self._pl_S = (C.c_int32 * self.n_prop)(*self.counts[: self.n_prop])
        # Ray terms (include/nsamd.h, nsamd_field_mlp.ray_terms): head layer 0's share of the 48 per-ray inputs (SH of the view
        # direction, appearance row) once per ray instead of once per sample. NSAMD_RAY_TERMS=0: the plain kernels (A/B).
        self.ray_terms_on = os.environ.get("NSAMD_RAY_TERMS", "1") == "1" and self.counts[-1] % 16 == 0
        self.ray_terms = e(n, 64) if self.ray_terms_on else None
        self._ray_terms_ready = False
        self.ray_inputs = e(n, 16 + (32 if fld.embedding_appearance is not None else 0)) \
            if (self.ray_terms_on and not forward_only) else None  # (the backward's: weight gradient of the 48 per-ray columns)
        self.f_denc = t(*self.f_enc.shape)
        self.p_ddens = [t(*x.shape) for x in self.p_dens]
        self.p_denc = [t(*x.shape) for x in self.p_enc]
        self.field_ws = None if forward_only else F.field_bwd_workspace(device)[0]
        # one scratch per proposal level: their backward chains may run concurrently on different streams
        self.density_ws = [] if forward_only else [F.density_bwd_workspace(device, slot=lvl) for lvl in range(self.n_prop)]
        # Zero-gradient gating of the proposal chains (include/nsamd.h): one device flag per level, raised by the level's
        # weights backward when any ray carries interlevel-loss gradient; while it is clear the rest of the chain
        # (density MLP backward, table scatter, ray gradients) returns at once — the zero-filled gradients are the result.
        # NSAMD_GATE_PROPOSALS=0: the ungated entry points (A/B).
        self.gate_proposals = os.environ.get("NSAMD_GATE_PROPOSALS", "1") == "1"
        self.prop_gates = torch.zeros(max(self.n_prop, 1) * 4, device=device, dtype=torch.int32)  # 16 B apart
        self.gates_precleared = False  # True: the caller zeroes `prop_gates` before every proposal backward (trainer.HipTrainer)
        # ... and its per-ray form (1 = the ray carries gradient): levels that are only partly without gradient
        self.prop_ray_masks = [torch.zeros(n, device=device, dtype=torch.uint8) for _ in range(self.n_prop)]
        # The main field's backward emits the table scatter's pass-1 records itself (nsamd_field_mlp_bwd_scatter: no `denc`
        # round trip, no route launch); NSAMD_FUSE_ROUTE=0: the two entry points (A/B).
        self.fuse_route = os.environ.get("NSAMD_FUSE_ROUTE", "1") == "1"
        self.keep_denc = False  # True: the fused launch also stores the encoded-feature gradient in `f_denc` (tests read it)
        self.prop_mlp_inline = os.environ.get("NSAMD_PROP_MLP_INLINE", "0") == "1"
        # compute units the persistent main backward leaves to the proposal chains on the iterations that run them
        # (0, the default: none; -1: one more sweep of the persistent workgroups, 36 of 256 CUs at 196 608 points; n: n CUs.
        #  Measured round 6, profiles/r06_s22_*, r06_s27_*: -0.5 % window / -0.7 % long run for a field backward that is 16 %
        #  longer on those iterations — opt-in)
        self.bwd_reserve_cus = int(os.environ.get("NSAMD_BWD_RESERVE_CUS", "0"))
        # the same stage of the proposal levels' backward chains as one launch across the levels (0: level by level, A/B)
        self.merge_prop_levels = os.environ.get("NSAMD_MERGE_PROP_LEVELS", "1") == "1"
        # the iteration's loss values and training metrics, written by the losses launch's finishing pass (nsamd.h):
        # rgb_loss, interlevel_loss, distortion_loss, psnr, distortion, sum of the three losses
        self.loss_vals = torch.zeros(32, **f32)  # (8 results + the finishing pass's scratch: partial sums, ticket)
        self._loss_vals_fresh = False
        self.want_loss_vals = False  # the separate-launch path: also launch nsamd_train_loss_values (set by pipeline.TrainEngine)
        # (slot pointer, slots, pool) of a batch selection the caller leaves to `forward_proposals` (one launch with the initial
        # bins, nsamd_select_bins); NSAMD_FUSE_SELECT=0: the caller launches nsamd_select_batch itself (A/B)
        self.pending_select = None
        self._slot0 = None  # a device zero: `set_batch` as a one-slot batch selection
        self._outputs = None
        self.fuse_select = os.environ.get("NSAMD_FUSE_SELECT", "1") == "1"
        # Second stream for the proposal-network backward: the two backward chains are independent, and since the
        # scatter kernels were reworked (latency-bound phases, small workgroups) they overlap: 3.87 -> 4.02 M rays/s on
        # MI355X (profiles/). `side_stream = None` runs them back to back (the data-parallel path does: its proposal
        # chain is the cover for the main-field all-reduce).
        self.side_stream = torch.cuda.Stream(device=device)
        # ... and the proposal levels are independent of each other too: level i > 0 gets its own stream
        self.level_streams = [torch.cuda.Stream(device=device) for _ in range(max(self.n_prop - 1, 0))]
        self._fork, self._join = torch.cuda.Event(), torch.cuda.Event()
        # the field backward's weight-gradient reduce on its own stream, beside the table scatter (opt-in: NSAMD_SPLIT_REDUCE=1; measured neutral)
        self.split_reduce = os.environ.get("NSAMD_SPLIT_REDUCE", "0") == "1"
        self.reduce_stream = torch.cuda.Stream(device=device)
        self.rays_beside_apply = os.environ.get("NSAMD_RAYS_BESIDE_APPLY", "1") == "1"  # (camera optimiser, see backward_field_and_table)
        self._red_fork, self._red_join = torch.cuda.Event(), torch.cuda.Event()
        self._level_join = [torch.cuda.Event() for _ in self.level_streams]
        # Proposal levels may run their backward chains on separate streams only when they share nothing: a shared
        # network (use_same_proposal_network) means one gradient buffer, and equal (grid, sample count) means one
        # scatter workspace — concurrent launches would race on either.
        keys = [(id(self.props[lvl]), self.props[lvl].encoding.spec, n * self.counts[lvl]) for lvl in range(self.n_prop)]
        self.levels_independent = (len({k[0] for k in keys}) == self.n_prop and
                                   len({(k[1], k[2]) for k in keys}) == self.n_prop)
        # Every level's chain on the ONE side stream by default (round 6, profiles/r06_s19_*: three alternating repeats, window
        # 0.6265 / 0.6304 / 0.6275 against 0.6494 / 0.6512 / 0.6278 ms with a stream per level, long run 0.689 against 0.695; the
        # chains in line on the main stream 0.674 / 0.736 — same bits in all three). NSAMD_LEVEL_STREAMS=1: a stream per level.
        if os.environ.get("NSAMD_LEVEL_STREAMS", "0") != "1":
            self.levels_independent = False
        # ---- camera optimiser (nerfstudio's nerfacto default is SO3xR3, the benchmark recipe is "off") ----
        # Host-side torch computes the corrected rays from `pose_adjustment` (a [num_cameras, 6] parameter); the kernels
        # return dL/d(origins, directions) per ray (nsamd_hashgrid_encode_bwd_rays, one buffer per sampling level so that
        # the levels' backward chains stay independent), and autograd carries it back to the parameter.
        co = getattr(model, "camera_optimizer", None)
        self.cam_opt = co if (co is not None and co.config.mode != "off") else None
        # The exponential map, the pose-to-ray arithmetic and their backward as two launches (nsamd_camera_apply /
        # nsamd_camera_backward) instead of ~100 eager torch kernels + autograd nodes: any optimiser with the reference's
        # parameterisation (a [num_cameras, 6] `pose_adjustment`, mode SO3xR3 / SE3, every camera trainable) — this
        # package's or the reference's own. NSAMD_CAMERA_KERNELS=0: the torch route (A/B).
        self.cam_kernels = False
        if self.cam_opt is not None:
            pose = getattr(co, "pose_adjustment", None)
            self.cam_kernels = (os.environ.get("NSAMD_CAMERA_KERNELS", "1") == "1" and pose is not None and pose.is_cuda
                                and pose.dtype == torch.float32 and pose.dim() == 2 and pose.shape[1] == 6
                                and co.config.mode in ("SO3xR3", "SE3")
                                and len(self.counts) <= 4  # nsamd_ray_grads has four level slots
                                and getattr(co, "non_trainable_camera_indices", None) is None)
            self.cam_mode = {"SO3xR3": 1, "SE3": 2}.get(co.config.mode, 0)
            self.raw_origins, self.raw_directions = e(n, 3), e(n, 3)
            self.d_origins = [e(n, 3) for _ in self.counts]
            self.d_directions = [e(n, 3) for _ in self.counts]
            self._corrected = None
            self.camera_reg = torch.zeros((), **f32)
        self.reg_in_backward = True  # backward_cameras also differentiates the pose regulariser
        # True: `backward_join` leaves the camera optimiser's share to the caller (`backward_cameras(updated, force=True)`
        # after a replayed graph: the exponential map's autograd graph is host-side torch and stays out of the capture)
        self.cameras_outside = False
        self.grad_lookup = None  # {id(parameter): gradient buffer} of an arena that does not bind `param.grad`
        self.main_table_write_only = True  # the main table's gradient is written, not accumulated (written_params)
        # True: backward_field_and_table leaves the table scatter to the caller (`backward_table`): bench.py's deferred
        # schedule runs it beside the NEXT iteration's proposal forward, from the copies `shadow_points` took
        self.defer_table = False
        self.sh_origins = self.sh_directions = self.sh_t_bins = None
        self.spacing = int(getattr(model.proposal_sampler.initial_sampler, "spacing", 0))
        # host-evaluated tables (bit-identical to the reference's CPU linspace)
        self.edges = F._linspace("edges", self.counts[0], device)
        self.u_base = [None] + [F._linspace("u", s, device) for s in self.counts[1:]]

    # -------------------------------------------------------------------------------------------------------------
    def set_batch(self, origins: Tensor, directions: Tensor, camera_indices: Tensor,
                  target_rgb: Optional[Tensor] = None) -> None:
        cams = camera_indices.reshape(-1)
        if (target_rgb is not None and origins.is_cuda and all(
                x.dtype == torch.float32 and x.is_contiguous() and x.shape == (self.n, 3) and x.device == self.origins.device
                for x in (origins, directions, target_rgb)) and cams.dtype == torch.int64 and cams.is_contiguous()
                and cams.shape[0] == self.n and cams.device == self.origins.device):
            # a datamanager's device-resident batch: ONE launch (a pool of one slot) instead of four copy kernels
            if self._slot0 is None:
                self._slot0 = torch.zeros(1, device=self.origins.device)
            o, d = (self.raw_origins, self.raw_directions) if self.cam_opt is not None else (self.origins, self.directions)
            N.check(N.load().nsamd_select_batch(N.ptr(self._slot0), 1, self.n, N.ptr(origins), N.ptr(directions), N.ptr(cams),
                                                N.ptr(target_rgb), N.ptr(o), N.ptr(d), N.ptr(self.camera_indices),
                                                N.ptr(self.target), N.stream()), "select_batch")
            return
        if self.cam_opt is not None:  # the kernels see the pose-corrected rays (apply_camera_corrections)
            self.raw_origins.copy_(origins)
            self.raw_directions.copy_(directions)
        else:
            self.origins.copy_(origins)
            self.directions.copy_(directions)
        self.camera_indices.copy_(camera_indices.reshape(-1))
        if target_rgb is not None:
            self.target.copy_(target_rgb)

    def prepare_grads(self, updated: bool) -> None:
        """For callers without a gradient arena (a trainer that runs `zero_grad(set_to_none=True)`, engine/optimizers.py:
        160-172): give every parameter this iteration produces a gradient for a buffer. Fresh buffers are zero-filled,
        except the main table's, which the scatter writes; a gradient that already exists is accumulated into
        (gradient accumulation), the main table's included."""
        fld = self.model.field
        table = fld.mlp_base.encoding.hash_table
        self.main_table_write_only = table.grad is None
        owners = [fld] + (list(self.model.proposal_networks) if updated else [])
        for mod in owners:
            for prm in mod.parameters():
                if prm.requires_grad and prm.grad is None:
                    prm.grad = torch.empty_like(prm) if prm is table else torch.zeros_like(prm)

    def ensure_proposal_features(self, lvl: int) -> None:
        """A forward-only runner whose proposal network the fused density kernel does not take (the two-kernel pair needs the
        level's encoded features and selector): allocate them on first use."""
        m = self.n * self.counts[lvl]
        if self.p_sel[lvl].numel() != m:
            dev = self.p_dens[lvl].device
            self.p_enc[lvl] = torch.empty((self.props[lvl].encoding.get_out_dim(), m), device=dev)
            self.p_sel[lvl] = torch.empty((m,), device=dev)

    def _grad(self, p: Tensor) -> Tensor:
        if self.grad_lookup is not None:  # an arena that keeps `param.grad` unset (arena.ParamArena(bind_grads=False))
            g = self.grad_lookup.get(id(p))
            if g is not None:
                return g
        assert p.grad is not None and p.grad.is_contiguous(), "parameters need preallocated .grad (use arena.ParamArena)"
        return p.grad

    def _points(self, lvl: int) -> N.Points:
        return N.make_points(None, self.origins, self.directions, self.t_bins[lvl], self.counts[lvl])

    # -------------------------------------------------------------------------------------------------------------
    def forward_backward(self, updated: bool, draw_jitter: bool = True) -> None:
        """One iteration up to (not including) the optimiser. `updated`: proposal networks receive gradient this step
        (ProposalNetworkSampler.updated_this_step()). Gradients ACCUMULATE into param.grad (zero them first)."""
        if self.forward_only:
            raise RuntimeError("NerfactoTrainStep(forward_only=True) has no gradient buffers")
        self.forward_and_losses(updated, draw_jitter)
        self.backward_all(updated)

    def proposal_branches(self):
        """[(stream, join event, proposal levels)]: how the proposal backward splits over streams. Levels that share a
        network or a scatter workspace stay on one stream; [] without a side stream."""
        if self.side_stream is None:
            return []
        if not self.levels_independent:
            return [(self.side_stream, self._join, None)]
        return [(self.side_stream, self._join, [0])] + [(ls, ev, [i + 1]) for i, (ls, ev) in
                                                         enumerate(zip(self.level_streams, self._level_join))]

    def backward_all(self, updated: bool) -> None:
        """Everything after the losses: the main backward chain, the proposal chains on the steps that update them
        (parallel streams where they share nothing; parallel branches inside a captured hipGraph), and the camera
        optimiser's share."""
        self.backward_fork(updated)
        self.backward_join(updated)

    def backward_fork(self, updated: bool) -> None:
        """First half of backward_all: the proposal chains go onto their side streams (not yet joined), the main chain runs
        on the current stream. A data-parallel caller starts the main-field gradient exchange between this and
        backward_join — the proposal chains then run beside the main chain AND beside the collective."""
        branches = self.proposal_branches() if updated else []
        if updated and os.environ.get("NSAMD_DIAG_SKIP_PROP_BWD") == "1":
            # timing diagnostic only (wrong training): update iterations without the proposal networks' backward chains — what
            # an update iteration would cost if those chains hid completely behind the main backward
            self._open_branches = []
            self.backward_main()
            return
        self._open_branches = branches
        if branches:
            # The backward chains are independent (disjoint gradients, separate scratch): fork the proposal chains onto
            # their own streams so that these latency-bound kernels overlap with the main chain.
            main = N.current_stream()
            side_stage = "all"
            if self.prop_mlp_inline:
                # NSAMD_PROP_MLP_INLINE=1 (experiment): the levels' weights + density-MLP backward run IN LINE ahead of the main
                # backward — beside it their 44-KiB workgroups delay the persistent main-backward workgroups at its start, and
                # what follows them starves until it ends — and only the table scatters are forked
                self.backward_proposals(stage="mlp")
                side_stage = "scatter"
            self._fork.record(main)
            for stream, join, levels in branches:
                stream.wait_event(self._fork)
                with N.on_stream(stream):
                    self.backward_proposals(levels=levels, stage=side_stage)
                    join.record(stream)
            self.backward_main(reserve=True)
        else:
            # (the reservation follows the KIND of iteration, not the streams: every schedule sums the same partials)
            self.backward_main(reserve=updated)
            if updated:
                self.backward_proposals()

    def backward_join(self, updated: bool) -> None:
        """Second half of backward_all: wait for the proposal chains, then the camera optimiser's share."""
        main = N.current_stream()
        for _, join, _ in getattr(self, "_open_branches", []):
            main.wait_event(join)
        self._open_branches = []
        self.backward_cameras(updated)

    def forward_backward_main(self, updated: bool, draw_jitter: bool = True) -> None:
        """Forward of everything, the losses, and the backward of the MAIN field (87 % of the gradient bytes). With data
        parallelism the all-reduce of the main-field gradients can start right after this while `backward_proposals`
        (interlevel-loss gradients of the proposal networks) still runs."""
        self.forward_and_losses(updated, draw_jitter)
        self.backward_main(reserve=updated)

    def written_params(self):
        """Parameters whose gradient this runner WRITES (hash tables, nsamd_hashgrid_encode_bwd_set): callers need not
        zero them (ParamArena.zero_grad(skip=...)). All other gradients accumulate and must be zeroed first."""
        # Only the 67 MB main table: for the 5 MB proposal tables the zero-fill is nothing and the accumulating call needs
        # no worst-case spill list.
        return [self.model.field.mlp_base.encoding.hash_table] if self.main_table_write_only else []

    def forward_and_losses(self, updated: bool, draw_jitter: bool = True) -> None:
        self.apply_camera_corrections()
        self.forward_proposals(draw_jitter, need_enc=updated)
        self.forward_main_and_losses(updated)

    # ---- camera optimiser -----------------------------------------------------------------------------------------
    def apply_camera_corrections(self) -> None:
        """origins + t, R @ directions with the current pose corrections (camera_optimizers.py:148-153) into the ray
        buffers the kernels read; the autograd graph of the tiny exponential map is kept for backward_cameras."""
        if self.cam_opt is None:
            return
        if self.cam_kernels:
            pose = self.cam_opt.pose_adjustment
            N.check(N.load().nsamd_camera_apply(N.ptr(pose), self.cam_mode, pose.shape[0], N.ptr(self.raw_origins),
                                                N.ptr(self.raw_directions), N.ptr(self.camera_indices), self.n,
                                                N.ptr(self.origins), N.ptr(self.directions), N.stream()), "camera_apply")
            return
        if hasattr(self.cam_opt, "corrected_rays"):
            o, d = self.cam_opt.corrected_rays(self.raw_origins, self.raw_directions, self.camera_indices)
        else:  # the reference's own CameraOptimizer: forward(indices) -> [n,3,4] corrections (camera_optimizers.py:107-153)
            c = self.cam_opt(self.camera_indices)
            o = self.raw_origins + c[:, :3, 3]
            d = torch.bmm(c[:, :3, :3], self.raw_directions[..., None]).squeeze(-1)
        self._corrected = (o, d)
        self.origins.copy_(o.detach())
        self.directions.copy_(d.detach())

    def _gate(self, lvl: int):
        """Device address of proposal level `lvl`'s gradient flag (None: gating off)."""
        return self.prop_gates.data_ptr() + 16 * lvl if self.gate_proposals else None

    def _rays_backward(self, lvl: int, net, denc: Tensor, gate=None, ray_mask=None) -> None:
        """dL/d(origins, directions) of sampling level `lvl` from its encoded-feature gradient (on the current stream)."""
        lib, m = N.load(), self.n * self.counts[lvl]
        enc = net.mlp_base.encoding if hasattr(net.mlp_base, "encoding") else net.encoding
        N.check(lib.nsamd_hashgrid_encode_bwd_rays_gated(
            self._points(lvl), m, net._transform, net._box, N.ptr(enc.hash_table), enc.spec.native(), N.ptr(denc), 1, m,
            N.ptr(self.d_origins[lvl]), N.ptr(self.d_directions[lvl]), 0, gate, ray_mask, N.stream()),
            "hashgrid_encode_bwd_rays")

    @profiler.time_function
    def backward_cameras(self, updated: bool, force: bool = False) -> None:
        """Per-ray gradients of every level that received one -> `pose_adjustment.grad` (plus the L2 regulariser of
        camera_optimizers.py:179-185, whose value is kept in `camera_reg`). Call after the backward chains have joined."""
        if self.cam_opt is None or (self.cameras_outside and not force):
            return
        L = self.n_prop
        if self.cam_kernels:
            pose = self.cam_opt.pose_adjustment
            if self.grad_lookup is not None and id(pose) in self.grad_lookup:
                g = self.grad_lookup[id(pose)]
            else:
                if pose.grad is None:
                    pose.grad = torch.zeros_like(pose)
                g = pose.grad
            levels = (list(range(L)) if updated else []) + [L]  # the proposal networks saw the rays too (interlevel loss)
            up = N.RayGrads()
            for k, lvl in enumerate(levels):
                up.d_origins[k], up.d_directions[k] = N.ptr(self.d_origins[lvl]), N.ptr(self.d_directions[lvl])
            up.count = len(levels)
            cfg = self.cam_opt.config
            reg = self.reg_in_backward  # (else the caller differentiates the regulariser itself — fused_step.FusedTrainStep)
            N.check(N.load().nsamd_camera_backward(
                N.ptr(pose), self.cam_mode, pose.shape[0], N.ptr(self.raw_directions), N.ptr(self.camera_indices), self.n, up,
                float(cfg.trans_l2_penalty) if reg else 0.0, float(cfg.rot_l2_penalty) if reg else 0.0, N.ptr(g),
                N.ptr(self.camera_reg) if reg else None, N.stream()), "camera_backward")
            return
        d_o, d_d = self.d_origins[L], self.d_directions[L]
        if updated:  # the proposal networks saw the rays too (interlevel loss)
            d_o = d_o + sum(self.d_origins[:L])
            d_d = d_d + sum(self.d_directions[:L])
        o, d = self._corrected
        outs, ups = [o, d], [d_o, d_d]
        if self.reg_in_backward:
            reg = {}
            self.cam_opt.get_loss_dict(reg)
            # a STATIC buffer: every captured schedule variant must leave the value where `loss_dict` reads it
            self.camera_reg.copy_(reg["camera_opt_regularizer"].detach())
            outs.append(reg["camera_opt_regularizer"])
            ups.append(torch.ones_like(self.camera_reg))
        # (else: the caller differentiates the regulariser itself — fused_step.FusedTrainStep)
        pose = [p for p in self.cam_opt.parameters() if p.requires_grad]
        if self.grad_lookup is not None and all(id(p) in self.grad_lookup for p in pose):
            for p, g in zip(pose, torch.autograd.grad(outs, pose, ups, allow_unused=True)):
                if g is not None:
                    self.grad_lookup[id(p)].add_(g)
        else:
            torch.autograd.backward(outs, ups)
        self._corrected = None

    @profiler.time_function
    def forward_proposals(self, draw_jitter: bool = True, need_enc: bool = True) -> None:
        """Initial bins and the proposal levels (density fields + resampling): reads only the proposal networks'
        parameters, so with data parallelism it can run while the main-field gradients of the previous step are still
        being all-reduced (bench.py). need_enc: keep the levels' encoded features / selector / pre-activation for
        backward_proposals (False on the steps where the proposal networks get no gradient, ray_samplers.py:590: the
        fused forward then writes nothing but the densities)."""
        lib, st, n = N.load(), N.stream(), self.n
        ck = N.check
        per_edge = not self.single_jitter
        if draw_jitter:
            if per_edge:
                for j in self.jitter_edges:
                    j.uniform_()
            else:
                self.jitter.uniform_()  # torch.rand per level and ray (ray_samplers.py:105, :322), drawn on the device
            if self.bg_rays is not None:
                self.bg_rays.uniform_()  # rand_like(pred) of the loss blend (renderers.py:195)
        S0 = self.counts[0]
        jit0 = self.jitter_edges[0] if per_edge else self.jitter[0]
        sel, self.pending_select = self.pending_select, None
        if sel is not None:
            # the step's batch out of the caller's pool of batches AND the initial bins in one launch (trainer.HipTrainer hands
            # the selection over instead of launching it: the bins need nears / fars / the draw, not the rays)
            slot, slots, pool = sel
            ck(lib.nsamd_select_bins(slot, slots, n, N.ptr(pool["origins"]), N.ptr(pool["directions"]), N.ptr(pool["cameras"]),
                                     N.ptr(pool["target"]), N.ptr(self.origins), N.ptr(self.directions),
                                     N.ptr(self.camera_indices), N.ptr(self.target), N.ptr(self.nears), N.ptr(self.fars),
                                     N.ptr(self.edges), N.ptr(jit0), int(per_edge), S0, self.spacing, N.ptr(self.s_bins[0]),
                                     N.ptr(self.t_bins[0]), st), "select_bins")
        else:
            ck(lib.nsamd_piecewise_bins(N.ptr(self.nears), N.ptr(self.fars), N.ptr(self.edges), N.ptr(jit0), int(per_edge), n,
                                        S0, self.spacing, N.ptr(self.s_bins[0]), N.ptr(self.t_bins[0]), st), "piecewise_bins")
        hook, self.after_bins = getattr(self, "after_bins", None), None
        if hook is not None:  # (trainer.HipTrainer: the point where the pending main-field Adam is forked off, NSAMD_FORK_AFTER_BINS)
            hook()
        # ---- proposal levels ----
        for lvl in range(self.n_prop):
            net = self.props[lvl]
            S, m = self.counts[lvl], n * self.counts[lvl]
            mlp = net.mlp_base[1]
            W0, b0, W1, b1 = mlp.param_tensors()
            dm = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0],
                              float(net.average_init_density))
            # hash grid + MLP + trunc_exp in one launch, features in registers (nsamd_density_field_fwd); networks the
            # fused kernel is not built for go through the two-kernel pair
            fused = lib.nsamd_density_field_fwd(
                self._points(lvl), m, net._transform, net._box, N.ptr(net.encoding.hash_table), net.encoding.spec.native(), dm,
                N.ptr(self.p_enc[lvl]) if need_enc else None, N.ptr(self.p_sel[lvl]) if need_enc else None,
                N.ptr(self.p_dens[lvl]), N.ptr(self.p_pre[lvl]) if need_enc else None, st)
            if fused == N.ERR_UNSUPPORTED:
                ck(lib.nsamd_hashgrid_encode_fwd(self._points(lvl), m, net._transform, net._box, N.ptr(net.encoding.hash_table),
                                                 net.encoding.spec.native(), N.ptr(self.p_enc[lvl]), 1, m,
                                                 N.ptr(self.p_sel[lvl]), st), "hashgrid_encode_fwd")
                ck(lib.nsamd_density_mlp_fwd(N.ptr(self.p_enc[lvl]), N.ptr(self.p_sel[lvl]), m, dm, N.ptr(self.p_dens[lvl]),
                                             N.ptr(self.p_pre[lvl]), st), "density_mlp_fwd")
            else:
                ck(fused, "density_field_fwd")
            S2 = self.counts[lvl + 1]
            if per_edge:
                # one draw per new bin edge: the fused launch below draws per ray, so this (non-default) configuration takes
                # its three constituents — same numbers (nsamd.h)
                ck(lib.nsamd_weights_fwd(N.ptr(self.t_bins[lvl]), N.ptr(self.p_dens[lvl]), n, S, N.ptr(self.weights[lvl]), st),
                   "weights_fwd")
                if self.compute_depths:
                    ck(lib.nsamd_composite_fwd(None, N.ptr(self.weights[lvl]), N.ptr(self.t_bins[lvl]), n, S, N.BG_NONE, None, 0,
                                               None, None, None, N.ptr(self.depth_med[lvl]), None, None, st), "composite_fwd(median)")
                ck(lib.nsamd_pdf_resample(N.ptr(self.s_bins[lvl]), N.ptr(self.weights[lvl]), S, N.ptr(self.u_base[lvl + 1]),
                                          N.ptr(self.jitter_edges[lvl + 1]), N.ptr(self.nears), N.ptr(self.fars), 1.0,
                                          N.ptr(self.anneal_dev), 0.01, 1e-5, 1.0 / (2 * (S2 + 1)), self.spacing, 1, 0, n, S2,
                                          N.ptr(self.s_bins[lvl + 1]), N.ptr(self.t_bins[lvl + 1]), None, st), "pdf_resample")
                continue
            # weights of this level, its median depth (prop_depth_i, models/nerfacto.py:346-347) and the PDF resampling
            ck(lib.nsamd_proposal_resample(N.ptr(self.t_bins[lvl]), N.ptr(self.s_bins[lvl]), N.ptr(self.p_dens[lvl]), S,
                                           N.ptr(self.u_base[lvl + 1]), N.ptr(self.jitter[lvl + 1]), N.ptr(self.nears),
                                           N.ptr(self.fars), 1.0, N.ptr(self.anneal_dev), 0.01, 1e-5,
                                           1.0 / (2 * (S2 + 1)), self.spacing, n, S2, N.ptr(self.weights[lvl]),
                                           N.ptr(self.depth_med[lvl]) if self.compute_depths else None,
                                           N.ptr(self.s_bins[lvl + 1]), N.ptr(self.t_bins[lvl + 1]), st),
               "proposal_resample")

    def forward_main_and_losses(self, updated: bool) -> None:
        """Main field on the final samples, compositing, and the three losses with their gradients."""
        self.forward_main()
        self.losses(updated)

    def ray_terms_launch(self) -> None:
        """nsamd_field_ray_terms for the batch in the static buffers: head layer 0's share of the 48 per-ray inputs. Needs the
        batch's (pose-corrected) directions and camera indices and the CURRENT head_W0 / head_b0 / appearance table — i.e. it
        runs after batch selection and after the main-field Adam of the previous iteration. On torch's current stream."""
        lib, st, n = N.load(), N.stream(), self.n
        fld = self.model.field
        params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        fm = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(emb), emb.shape[0] if emb is not None else 0,
                        float(fld.average_init_density))
        cams = N.ptr(self.camera_indices) if emb is not None else None
        N.check(lib.nsamd_field_ray_terms(N.ptr(self.directions), cams, None, n, fm, N.ptr(self.ray_terms), N.ptr(self.ray_inputs),
                                          st), "field_ray_terms")
        self._ray_terms_ready = True

    @profiler.time_function
    def forward_main(self) -> None:
        """Hash grid + MLPs of the main field on the final samples -> per-sample density and rgb."""
        lib, st, n, cfg = N.load(), N.stream(), self.n, self.cfg
        ck = N.check
        fld = self.model.field
        L = self.n_prop
        S, mm = self.counts[L], self.m_main
        enc = fld.mlp_base.encoding
        params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        fm = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(emb), emb.shape[0] if emb is not None else 0,
                        float(fld.average_init_density))
        cams = N.ptr(self.camera_indices) if emb is not None else None
        if self.ray_terms_on:
            if not self._ray_terms_ready:  # (a caller may have launched them already, off the critical path: trainer.HipTrainer)
                self.ray_terms_launch()
            self._ray_terms_ready = False
            fm.ray_terms, fm.ray_inputs = N.ptr(self.ray_terms), N.ptr(self.ray_inputs)
        ck(lib.nsamd_hashgrid_encode_fwd(self._points(L), mm, fld._transform, fld._box, N.ptr(enc.hash_table),
                                         enc.spec.native(), N.ptr(self.f_enc), 1, mm, N.ptr(self.f_sel), st),
           "hashgrid_encode_fwd")
        ck(lib.nsamd_field_mlp_fwd(N.ptr(self.f_enc), N.ptr(self.f_sel), N.ptr(self.directions), cams, None, S, mm, fm,
                                   N.ptr(self.f_dens), N.ptr(self.f_rgb), st), "field_mlp_fwd")

    @profiler.time_function
    def losses(self, updated: bool) -> None:
        """Weights, compositing (rgb / accumulation / depths: the model outputs) and the three losses with their gradients
        against `self.target`. Reads only what forward_main left behind, so it can be repeated with another target."""
        lib, st, n, cfg = N.load(), N.stream(), self.n, self.cfg
        ck = N.check
        L = self.n_prop
        S = self.counts[L]
        self._loss_vals_fresh = False
        # weights + compositing + MSE value/gradient in one launch (+ the global depth clip)
        ck(lib.nsamd_render_train(N.ptr(self.f_rgb), N.ptr(self.f_dens), N.ptr(self.t_bins[L]), n, S, self.bg_mode,
                                  self.bg_vals, N.ptr(self.target), 1.0 / (3 * n), N.ptr(self.weights[L]), N.ptr(self.rgb),
                                  N.ptr(self.acc), N.ptr(self.depth_exp),
                                  N.ptr(self.depth_med[L]) if self.compute_depths else None, N.ptr(self.minmax_ws),
                                  N.ptr(self.sq_err), N.ptr(self.d_rgb_out), N.ptr(self.bg_rays), st), "render_train")
        # ---- proposal losses: value + gradient, all levels in one launch (models/nerfacto.py:363-375) ----
        ck(lib.nsamd_proposal_losses(N.ptr(self.s_bins[L]), N.ptr(self.weights[L]), S, self.n_prop, self._pl_s_bins,
                                     self._pl_weights, self._pl_S, n, float(cfg.interlevel_loss_mult) / (n * S),
                                     float(cfg.distortion_loss_mult) / n, self._pl_per_ray,
                                     self._pl_dw if updated else None, N.ptr(self.dist_per_ray), N.ptr(self.dw_dist), st),
           "proposal_losses")
        if self.want_loss_vals:
            # a caller that reads the loss dictionary every iteration (pipeline.TrainEngine): the values and the training metrics
            # as five floats from one small launch instead of a dozen reductions issued by the host
            ck(lib.nsamd_train_loss_values(N.ptr(self.sq_err), N.ptr(self.dist_per_ray), self.n_prop, self._pl_per_ray, n, S,
                                           float(cfg.interlevel_loss_mult), float(cfg.distortion_loss_mult),
                                           N.ptr(self.loss_vals), st), "train_loss_values")
            self._loss_vals_fresh = True

    @profiler.time_function
    def backward_main(self, field: bool = True, reserve: bool = False) -> None:
        """composite -> weights -> field MLPs -> main hash table (MSE + distortion gradients). `field=False`: stop before
        the field's MLPs (the caller runs `backward_field_and_table`). `reserve`: an iteration that also runs the proposal
        levels' backward chains — the persistent field backward leaves `bwd_reserve_cus` compute units to them."""
        if reserve and field and self.bwd_reserve_cus != 0:
            lib = N.load()
            prev = lib.nsamd_field_mlp_bwd_reserve_cus(self.bwd_reserve_cus)
            try:
                self.backward_main(field=True)
            finally:
                lib.nsamd_field_mlp_bwd_reserve_cus(prev)
            return
        lib, st, n = N.load(), N.stream(), self.n
        ck = N.check
        L = self.n_prop
        S = self.counts[L]
        ck(lib.nsamd_render_train_bwd(N.ptr(self.f_rgb), N.ptr(self.weights[L]), N.ptr(self.f_dens), N.ptr(self.t_bins[L]),
                                      n, S, self.bg_mode, self.bg_vals, N.ptr(self.d_rgb_out), N.ptr(self.dw_dist),
                                      N.ptr(self.d_rgb_s), N.ptr(self.d_dens_main), N.ptr(self.bg_rays), st),
           "render_train_bwd")
        if self.gradient_scaling:  # scale_gradients_by_distance_squared on the field's outputs (models/nerfacto.py:321-322)
            ck(lib.nsamd_distance_gradient_scale(N.ptr(self.t_bins[L]), n, S, N.ptr(self.d_dens_main), N.ptr(self.d_rgb_s), st),
               "distance_gradient_scale")
        if field:
            self.backward_field_and_table()

    def backward_field_and_table(self) -> None:
        """Second half of backward_main: the main field's MLPs (from `d_dens_main`, `d_rgb_s`) and the table scatter. Separate
        so that tests can drive the field backward with upstream gradients of their own."""
        lib, st, n = N.load(), N.stream(), self.n
        ck = N.check
        fld = self.model.field
        L = self.n_prop
        S, mm = self.counts[L], self.m_main
        enc = fld.mlp_base.encoding
        params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        fm = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(emb), emb.shape[0] if emb is not None else 0,
                        float(fld.average_init_density))
        cams = N.ptr(self.camera_indices) if emb is not None else None
        if self.ray_terms_on:  # (what forward_main computed for this batch and these parameters)
            fm.ray_terms, fm.ray_inputs = N.ptr(self.ray_terms), N.ptr(self.ray_inputs)
        grads = N.FieldMlpGrads(*(N.ptr(self._grad(p)) for p in params), N.ptr(self._grad(emb)) if emb is not None else None)
        split = self.split_reduce and self.side_stream is not None
        if (self.fuse_route and self.main_table_write_only and not self.defer_table
                and enc.spec.num_levels == 16):
            sws, sws_n = F._producer_scatter_workspace(enc.spec, self.f_enc.device, mm)
            if sws is not None:
                want_denc = self.cam_opt is not None or self.keep_denc  # the camera optimiser's share needs the feature gradient as well
                args = (self._points(L), fld._transform, fld._box, enc.spec.native(), N.ptr(self.f_enc), N.ptr(self.f_sel),
                        N.ptr(self.directions), cams, None, S, mm, fm, N.ptr(self.d_dens_main), N.ptr(self.d_rgb_s),
                        N.ptr(self.f_denc) if want_denc else None, grads, N.ptr(self.field_ws), self.field_ws.numel(),
                        N.ptr(self._grad(enc.hash_table)), N.ptr(sws), sws_n)
                if N.PROFILE is not None:  # the per-kernel table (utils/roofline.py): one launch group at a time, same bits
                    for phase in (1, 2, 4):
                        ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, phase, st), "field_mlp_bwd_scatter_phase")
                elif split:
                    # the weight-gradient reduce (12.8 MB of partial rows, latency-bound) needs nothing the apply pass produces
                    # and vice versa: the reduce on its own stream BESIDE the apply pass (NSAMD_SPLIT_REDUCE=1; same bits)
                    ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, 1, st), "field_mlp_bwd_scatter_phase")
                    main = N.current_stream()
                    self._red_fork.record(main)
                    self.reduce_stream.wait_event(self._red_fork)
                    with N.on_stream(self.reduce_stream):
                        ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, 2, N.stream()), "field_mlp_bwd_scatter_phase")
                        self._red_join.record(self.reduce_stream)
                    ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, 4, st), "field_mlp_bwd_scatter_phase")
                    main.wait_event(self._red_join)
                elif os.environ.get("NSAMD_DIAG_SKIP_DW_REDUCE") == "1":
                    # timing diagnostic only (wrong training): no weight-gradient reduce at all — what the iteration would cost
                    # if the reduce were free
                    for phase in (1, 4):
                        ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, phase, st), "field_mlp_bwd_scatter_phase")
                elif self.cam_opt is not None and self.rays_beside_apply:
                    # camera optimiser: the rays' gradient through the main grid (a gather pass over the table, 96 us) needs the
                    # encoded-feature gradient the kernel has just written and nothing of the table scatter's apply pass (LDS
                    # atomics, latency-bound): beside it on the second stream instead of behind it (NSAMD_RAYS_BESIDE_APPLY=0: A/B)
                    ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, 1, st), "field_mlp_bwd_scatter_phase")
                    main = N.current_stream()
                    self._red_fork.record(main)
                    self.reduce_stream.wait_event(self._red_fork)
                    with N.on_stream(self.reduce_stream):
                        self._rays_backward(L, fld, self.f_denc)
                        self._red_join.record(self.reduce_stream)
                    ck(lib.nsamd_field_mlp_bwd_scatter_phase(*args, 6, st), "field_mlp_bwd_scatter_phase")  # (the reduce rides the apply pass)
                    main.wait_event(self._red_join)
                    return
                else:
                    ck(lib.nsamd_field_mlp_bwd_scatter(*args, st), "field_mlp_bwd_scatter")
                if self.cam_opt is not None:
                    self._rays_backward(L, fld, self.f_denc)
                return
        if split:
            # The sum of the per-workgroup weight-gradient partials needs nothing the table scatter produces and vice
            # versa: the gradient kernel here, the reduce on its own stream beside the scatter (joined at the end).
            args = (N.ptr(self.f_enc), N.ptr(self.f_sel), N.ptr(self.directions), cams, None, S, mm, fm,
                    N.ptr(self.d_dens_main), N.ptr(self.d_rgb_s), N.ptr(self.f_denc), grads, N.ptr(self.field_ws),
                    self.field_ws.numel())
            ck(lib.nsamd_field_mlp_bwd_phase(*args, 1, st), "field_mlp_bwd_phase")
            main = N.current_stream()
            self._red_fork.record(main)
            self.reduce_stream.wait_event(self._red_fork)
            with N.on_stream(self.reduce_stream):
                ck(lib.nsamd_field_mlp_bwd_phase(*args, 2, N.stream()), "field_mlp_bwd_phase")
                self._red_join.record(self.reduce_stream)
        else:
            ck(lib.nsamd_field_mlp_bwd(N.ptr(self.f_enc), N.ptr(self.f_sel), N.ptr(self.directions), cams, None, S, mm, fm,
                                       N.ptr(self.d_dens_main), N.ptr(self.d_rgb_s), N.ptr(self.f_denc), grads,
                                       N.ptr(self.field_ws), self.field_ws.numel(), st), "field_mlp_bwd")
        if self.cam_opt is not None:
            self._rays_backward(L, fld, self.f_denc)
        if not self.defer_table:
            self.backward_table()
        if split:
            N.current_stream().wait_event(self._red_join)

    def backward_table(self, shadow: bool = False) -> None:
        """The main table's gradient scatter from `f_denc`. `shadow`: the sample points come from the copies `shadow_points`
        took (a caller that runs this scatter AFTER the next batch has been selected: bench.py's deferred schedule)."""
        lib, st = N.load(), N.stream()
        ck = N.check
        fld = self.model.field
        L = self.n_prop
        mm = self.m_main
        enc = fld.mlp_base.encoding
        pts = N.make_points(None, self.sh_origins, self.sh_directions, self.sh_t_bins, self.counts[L]) if shadow else self._points(L)
        if self.main_table_write_only:
            ws, ws_n = F._scatter_workspace(enc.spec, self.f_enc.device, mm, write_only=True)
            ck(lib.nsamd_hashgrid_encode_bwd_set(pts, mm, fld._transform, fld._box, N.ptr(enc.hash_table),
                                                 enc.spec.native(), N.ptr(self.f_denc), 1, mm,
                                                 N.ptr(self._grad(enc.hash_table)), None, N.ptr(ws), ws_n, st),
               "hashgrid_encode_bwd")
        else:  # accumulate into an existing gradient (prepare_grads)
            ws, ws_n = F._scatter_workspace(enc.spec, self.f_enc.device, mm)
            ck(lib.nsamd_hashgrid_encode_bwd(pts, mm, fld._transform, fld._box, N.ptr(enc.hash_table),
                                             enc.spec.native(), N.ptr(self.f_denc), 1, mm,
                                             N.ptr(self._grad(enc.hash_table)), None, N.ptr(ws), ws_n, st),
               "hashgrid_encode_bwd")

    def shadow_points(self) -> None:
        """Copies of what defines the final samples (ray origins, directions, bin edges: 0.9 MB) for a table scatter that
        runs after the next batch has overwritten them (`backward_table(shadow=True)`)."""
        L = self.n_prop
        if self.sh_origins is None:
            self.sh_origins, self.sh_directions = torch.empty_like(self.origins), torch.empty_like(self.directions)
            self.sh_t_bins = torch.empty_like(self.t_bins[L])
        self.sh_origins.copy_(self.origins)
        self.sh_directions.copy_(self.directions)
        self.sh_t_bins.copy_(self.t_bins[L])

    @profiler.time_function
    def backward_proposals(self, levels=None, stage: str = "all") -> None:
        """Backward of the proposal networks (interlevel loss only; main-level weights are detached, losses.py:119-120).
        Needs the dw_prop written by forward_backward_main(updated=True). `levels`: subset of proposal levels (their
        chains share nothing, so they may run on different streams). `stage`: "mlp" = weights backward + density-MLP backward
        (its weight-gradient reduce included), "scatter" = ray gradients + table scatter, "all" = both in order."""
        lib, st, n = N.load(), N.stream(), self.n
        ck = N.check
        do_mlp, do_scatter = stage in ("all", "mlp"), stage in ("all", "scatter")
        lvls = list(range(self.n_prop) if levels is None else levels)
        if (self.merge_prop_levels and stage == "all" and len(lvls) >= 2 and self.gate_proposals and self.cam_opt is None
                and N.PROFILE is None):
            # every stage of the levels' chains as ONE launch across the levels (nsamd_proposal_levels_bwd; same bits)
            arr = self._proposal_level_structs(lvls)
            if arr is not None:
                ck(lib.nsamd_proposal_levels_bwd(arr, len(lvls), int(self.gates_precleared), st), "proposal_levels_bwd")
                return
        for _ in (0,):
            for lvl in lvls:
                net = self.props[lvl]
                S, m = self.counts[lvl], n * self.counts[lvl]
                mlp = net.mlp_base[1]
                W0, b0, W1, b1 = mlp.param_tensors()
                dm = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0],
                                  float(net.average_init_density))
                gate = self._gate(lvl)
                dws = self.density_ws[lvl]
                spec = net.encoding.spec
                ws, ws_n = F._scatter_workspace(spec, self.f_enc.device, m)
                grads = (N.ptr(self._grad(W0)), N.ptr(self._grad(b0)), N.ptr(self._grad(W1)), N.ptr(self._grad(b1)))
                if gate is None or ws is None:  # ungated chain (A/B switch, or no binned-scatter workspace for this shape)
                    if do_mlp:
                        ck(lib.nsamd_weights_bwd(N.ptr(self.t_bins[lvl]), N.ptr(self.p_dens[lvl]), N.ptr(self.dw_prop[lvl]),
                                                 n, S, N.ptr(self.p_ddens[lvl]), st), "weights_bwd")
                        ck(lib.nsamd_density_mlp_bwd(N.ptr(self.p_enc[lvl]), N.ptr(self.p_sel[lvl]), N.ptr(self.p_pre[lvl]),
                                                     N.ptr(self.p_ddens[lvl]), m, dm, N.ptr(self.p_denc[lvl]), *grads,
                                                     N.ptr(dws), dws.numel(), st), "density_mlp_bwd")
                    if do_scatter:
                        if self.cam_opt is not None:
                            self._rays_backward(lvl, net, self.p_denc[lvl])
                        ck(lib.nsamd_hashgrid_encode_bwd(self._points(lvl), m, net._transform, net._box,
                                                         N.ptr(net.encoding.hash_table), spec.native(), N.ptr(self.p_denc[lvl]),
                                                         1, m, N.ptr(self._grad(net.encoding.hash_table)), None, N.ptr(ws), ws_n,
                                                         st), "hashgrid_encode_bwd")
                    continue
                # the weights backward raises the level's flag when any ray carries interlevel gradient; the rest of the
                # chain returns at once while it is clear (the zero-filled gradients are then already the result)
                mask = N.ptr(self.prop_ray_masks[lvl])
                if do_mlp:
                    ck(lib.nsamd_weights_bwd_gate(N.ptr(self.t_bins[lvl]), N.ptr(self.p_dens[lvl]), N.ptr(self.dw_prop[lvl]),
                                                  n, S, N.ptr(self.p_ddens[lvl]), gate, mask, int(self.gates_precleared), st),
                       "weights_bwd_gate")
                    ck(lib.nsamd_density_mlp_bwd_gated(N.ptr(self.p_enc[lvl]), N.ptr(self.p_sel[lvl]), N.ptr(self.p_pre[lvl]),
                                                       N.ptr(self.p_ddens[lvl]), m, dm, N.ptr(self.p_denc[lvl]), *grads,
                                                       N.ptr(dws), dws.numel(), gate, mask, S, st), "density_mlp_bwd_gated")
                if do_scatter:
                    if self.cam_opt is not None:
                        self._rays_backward(lvl, net, self.p_denc[lvl], gate, mask)
                    ck(lib.nsamd_hashgrid_encode_bwd_gated(self._points(lvl), m, net._transform, net._box,
                                                           N.ptr(net.encoding.hash_table), spec.native(),
                                                           N.ptr(self.p_denc[lvl]), 1, m,
                                                           N.ptr(self._grad(net.encoding.hash_table)), N.ptr(ws), ws_n, gate,
                                                           mask, st), "hashgrid_encode_bwd_gated")

    def _proposal_level_structs(self, lvls):
        """ctypes array of nsamd_proposal_level_bwd for `lvls` (None: a level without a binned-scatter workspace). The structs
        hold raw addresses of this runner's static buffers and of the parameters / gradients, which do not move."""
        arr = (N.ProposalLevelBwd * len(lvls))()
        n = self.n
        for i, lvl in enumerate(lvls):
            net = self.props[lvl]
            S, m = self.counts[lvl], n * self.counts[lvl]
            W0, b0, W1, b1 = net.mlp_base[1].param_tensors()
            spec = net.encoding.spec
            ws, ws_n = F._scatter_workspace(spec, self.f_enc.device, m)
            if ws is None:
                return None
            dws = self.density_ws[lvl]
            e = arr[i]
            e.num_rays, e.samples_per_ray = n, S
            e.t_bins, e.density, e.dweights = N.ptr(self.t_bins[lvl]), N.ptr(self.p_dens[lvl]), N.ptr(self.dw_prop[lvl])
            e.ddensity, e.gate, e.ray_mask = N.ptr(self.p_ddens[lvl]), self._gate(lvl), N.ptr(self.prop_ray_masks[lvl])
            e.enc, e.selector, e.pre = N.ptr(self.p_enc[lvl]), N.ptr(self.p_sel[lvl]), N.ptr(self.p_pre[lvl])
            e.mlp = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0],
                                 float(net.average_init_density))
            e.denc = N.ptr(self.p_denc[lvl])
            e.dW0, e.db0, e.dW1, e.db1 = (N.ptr(self._grad(t)) for t in (W0, b0, W1, b1))
            e.mlp_workspace, e.mlp_workspace_floats = N.ptr(dws), dws.numel()
            e.origins, e.directions = N.ptr(self.origins), N.ptr(self.directions)
            e.transform, e.aabb = net._transform, net._box
            e.table, e.grid = N.ptr(net.encoding.hash_table), spec.native()
            e.dtable = N.ptr(self._grad(net.encoding.hash_table))
            e.scatter_workspace, e.scatter_workspace_floats = N.ptr(ws), ws_n
        return arr

    # -------------------------------------------------------------------------------------------------------------
    def loss_dict(self) -> Dict[str, Tensor]:
        """Loss values of the last iteration (models/nerfacto.py:363-375); a few tiny torch reductions, call on demand."""
        n, S = self.n, self.counts[-1]
        if self._loss_vals_fresh:  # what the losses launch left behind: no reduction launches
            # ONE clone: the buffer is overwritten by every iteration, and a caller that keeps the dictionaries of several
            # iterations (a trainer's logging history) must not see them all change to the latest values
            v = self.loss_vals[:3].clone()
            out = {"rgb_loss": v[0], "distortion_loss": v[2], "interlevel_loss": v[1]}
            if self.cam_opt is not None:
                out["camera_opt_regularizer"] = self.camera_reg
            return out
        out = {"rgb_loss": self.sq_err.sum() / (3 * n),
               "distortion_loss": self.cfg.distortion_loss_mult * self.dist_per_ray.sum() / n}
        inter = sum(p.sum() for p in self.inter_per_ray) / (n * S)
        out["interlevel_loss"] = self.cfg.interlevel_loss_mult * inter
        if self.cam_opt is not None:
            out["camera_opt_regularizer"] = self.camera_reg
        return out

    def outputs(self) -> Dict[str, Tensor]:
        """The tensors NerfactoModel.get_outputs returns, as views of the static buffers (built once: the buffers never move, and a
        trainer asks for them every iteration — a dozen view constructions were a measurable share of the host's work per step)."""
        if self._outputs is None:
            out = {"rgb": self.rgb, "accumulation": self.acc[:, None], "expected_depth": self.depth_exp[:, None],
                   "weights_list": [w[..., None] for w in self.weights]}
            if self.compute_depths:
                out["depth"] = self.depth_med[-1][:, None]
                for i in range(self.n_prop):
                    out[f"prop_depth_{i}"] = self.depth_med[i][:, None]
            self._outputs = out
        return dict(self._outputs)
