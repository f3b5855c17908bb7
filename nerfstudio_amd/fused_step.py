"""The explicit kernel schedule (train_step.NerfactoTrainStep) behind the reference's Model API.

A nerfstudio trainer drives a model through `get_outputs(ray_bundle)` -> `get_metrics_dict` -> `get_loss_dict` ->
`loss.backward()` -> optimiser (pipelines/base_pipeline.py:290-303, engine/trainer.py:487-531). Through the nn.Module /
autograd classes of this package that is ~60 autograd nodes and as many small host calls per iteration: the step is
host-bound (2.1 ms eager on MI355X against 0.9 ms for the same kernels launched by the runner,
profiles/r02_module_path.txt). `FusedTrainStep` keeps the Model API and runs the runner underneath:

  get_outputs(ray_bundle)      rays -> [camera corrections] -> proposal levels -> main field -> compositing; returns the
                               reference's output dict (rgb, accumulation, depth, expected_depth, prop_depth_i,
                               weights_list) as views of the runner's static buffers
  get_loss_dict(outputs, batch) the three losses against batch["image"] (one more launch of the fused
                               compositing + MSE kernel with the real target, and the proposal-loss kernel); the values
                               come back as outputs of ONE autograd node whose backward launches the backward chains,
                               which write the parameter gradients straight into `param.grad`
  loss.backward()              -> the runner's backward_all; parameters whose `.grad` is None get a buffer
                               (engine/optimizers.py:160-172 zeroes with set_to_none=True), existing gradients are
                               accumulated into

Contract: the loss terms are summed with unit weights before `backward()` (the reference's trainer does:
engine/trainer.py:514 `functools.reduce(torch.add, loss_dict.values())`; mixed precision is off for this method) — the
kernels produce the gradient of that sum; any other upstream gradient raises. Same kernels in the same order as the
module path, so outputs, losses and gradients agree to rounding (tests/test_gpu_kernels.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor

_LOSS_KEYS = ("rgb_loss", "interlevel_loss", "distortion_loss")


def ddp_reason() -> Optional[str]:
    """The fused steps write parameter gradients straight into `param.grad`: DistributedDataParallel's reducer, which hooks
    autograd's gradient accumulation (pipelines/base_pipeline.py:279-282 wraps the model in DDP when world_size > 1), would
    never see them and the ranks would silently diverge. More than one rank takes the module path under DDP, or bench.py's
    arena exchange (arena.ParamArena + dp_schedule.PipelinedExchange)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return "world size > 1 under DistributedDataParallel"
    return None


class _FusedLosses(torch.autograd.Function):
    """forward: the loss values the kernels already computed; backward: the runner's backward chains."""

    @staticmethod
    def forward(ctx, anchor: Tensor, step: "FusedTrainStep"):  # noqa: D102
        ctx.step = step
        ctx.updated = step.updated
        ld = step.runner.loss_dict()
        return tuple(ld[k].clone() for k in _LOSS_KEYS)

    @staticmethod
    def backward(ctx, *grads):  # noqa: D102
        step: "FusedTrainStep" = ctx.step
        # the kernels hold d(sum of the terms): verify the caller asked for exactly that — on the first iterations and
        # then on every 64th backward (a host sync each; a GradScaler switched on at resume, a changed loss coefficient or a
        # caller that back-propagates a single term must not pass silently for long)
        step.backward_calls += 1
        if step.checks_left > 0 or step.backward_calls % 64 == 0:
            step.checks_left = max(step.checks_left - 1, 0)
            for g in grads:
                if g is None or float(g) != 1.0:
                    raise RuntimeError("FusedTrainStep: the loss terms must be summed with unit weights before backward() "
                                       "(engine/trainer.py:514); use the module path for anything else")
        r = step.runner
        r.prepare_grads(ctx.updated)
        r.backward_all(ctx.updated)
        return None, None


class FusedTrainStep:
    def __init__(self, model) -> None:
        self.model = model
        self.runner = None
        self.updated = False
        self.checks_left = 3  # upstream-gradient checks cost a host sync each: the first iterations, then every 64th
        self.backward_calls = 0

    def supported(self) -> Optional[str]:
        """None, or the reason this model has to stay on the module path."""
        cfg = self.model.config
        if getattr(cfg, "predict_normals", False):
            return "predict_normals"
        return ddp_reason()

    def _runner_for(self, num_rays: int, device):
        from .train_step import NerfactoTrainStep

        if self.runner is None or self.runner.n != num_rays:
            from . import _native as N

            N.require_cuda(torch.empty(0, device=device))  # the kernel wrappers' loud error, before any device object is built
            self.runner = NerfactoTrainStep(self.model, num_rays, device)
            self.runner.reg_in_backward = False  # get_loss_dict hands the regulariser to autograd
        return self.runner

    # --- Model.get_outputs (models/nerfacto.py:298-360), training mode -----------------------------------------------------
    def get_outputs(self, ray_bundle, jitters: Optional[List[Tensor]] = None) -> Dict[str, object]:
        m = self.model
        o = ray_bundle.origins.reshape(-1, 3)
        r = self._runner_for(o.shape[0], o.device)
        ps = m.proposal_sampler
        forced = getattr(ps, "force_updated", None)  # a caller that replays captured schedule variants decides itself
        self.updated = bool(ps.updated_this_step() if forced is None else forced)
        r.set_batch(o, ray_bundle.directions.reshape(-1, 3), ray_bundle.camera_indices.reshape(-1))
        r.anneal_dev.fill_(float(ps._anneal))  # BEFORE_TRAIN_ITERATION callback's value (models/nerfacto.py:270-280)
        if jitters is not None:  # parity tests inject the draws of the module path
            for lvl, j in enumerate(jitters):
                if getattr(r, "jitter_edges", None) is not None:  # use_single_jitter=False: one draw per bin edge
                    r.jitter_edges[lvl].copy_(j.reshape(r.jitter_edges[lvl].shape))
                else:
                    r.jitter[lvl].copy_(j.reshape(-1))
        r.apply_camera_corrections()
        r.forward_proposals(draw_jitter=jitters is None, need_enc=self.updated)
        r.forward_main()
        r.losses(self.updated)  # compositing = the model outputs (its loss half is redone once the target is known)
        if self.updated and forced is None:
            ps.mark_updated()  # ray_samplers.py:606-607
        out = r.outputs()
        out["fused_step"] = self
        return out

    # --- Model.get_loss_dict (models/nerfacto.py:363-392) ----------------------------------------------------------------
    def get_loss_dict(self, outputs, batch) -> Dict[str, Tensor]:
        assert outputs.get("fused_step") is self, "outputs of another forward"
        r = self.runner
        # RGBA targets are blended with the background the PREDICTION is blended with (models/nerfacto.py:377-381 ->
        # renderers.py:175-199): "last_sample" counts as black, a named / tensor colour as itself, and "random" is the
        # per-ray colour this iteration drew for the prediction (`bg_rays`, which the loss kernel adds as bg (1 - acc)) —
        # not black, which would pull transparent pixels towards black against a random-coloured prediction.
        # RGB targets pass through unchanged.
        image = batch["image"].to(r.target.device)
        if image.shape[-1] == 4:
            image = image.reshape(-1, 4)
            image = self.model.renderer_rgb.blend_background(image, background_color=r.bg_rays if r.bg_rays is not None else None)
        r.target.copy_(image.reshape(-1, 3))
        r.losses(self.updated)
        anchor = self.model.field.mlp_base.encoding.hash_table  # any parameter: makes autograd call backward
        loss_dict = dict(zip(_LOSS_KEYS, _FusedLosses.apply(anchor, self)))
        if hasattr(self.model, "camera_optimizer"):
            # L2 regulariser on the pose corrections: plain autograd on the [num_cameras, 6] parameter; the runner adds
            # only the rays' share of that parameter's gradient (reg_in_backward = False)
            self.model.camera_optimizer.get_loss_dict(loss_dict)
        return loss_dict

    def get_metrics_dict(self, outputs, batch) -> Dict[str, Tensor]:
        """psnr against the (background-blended) target and the distortion term (models/nerfacto.py:352-361)."""
        m = self.model
        gt = m.renderer_rgb.blend_background(batch["image"].to(outputs["rgb"].device))
        mse = torch.mean((outputs["rgb"].detach() - gt.reshape(-1, 3)) ** 2)
        metrics = {"psnr": -10.0 * torch.log10(mse), "distortion": self.runner.dist_per_ray.sum() / self.runner.n}
        if hasattr(m, "camera_optimizer"):
            m.camera_optimizer.get_metrics_dict(metrics)
        return metrics
