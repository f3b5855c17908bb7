"""`HipPipeline`: the `nerfacto-hip` method's pipeline — the reference's `VanillaPipeline` with the training iteration on
the captured kernel schedule (trainer.HipTrainer) and, for more than one rank, the arena's pipelined gradient exchange
INSTEAD OF DistributedDataParallel.

Reference seam (nothing of nerfstudio is edited):
    Trainer.train_iteration (engine/trainer.py:487-531)
        optimizers.zero_grad_some            -> `param.grad` is None throughout: the gradients live in the arena
        pipeline.get_train_loss_dict(step)   -> THIS: datamanager.next_train -> static buffers -> ONE graph replay (forward,
                                                losses, backward, fused Adam of the arena) -> loss / metrics dictionaries
        loss.backward()                      -> a no-op node (the backward already ran inside the replay)
        optimizers.optimizer_scaler_step_some-> finds no gradient and steps nothing (engine/optimizers.py:160-172)
        optimizers.scheduler_step_all        -> the reference's schedulers keep computing the learning rates; the arena's
                                                Adam reads them from the torch optimisers' `param_groups` every iteration
    base_pipeline.py:279-282 wraps the model in DDP(find_unused_parameters=True) when world_size > 1; DDP never sees
    gradients that kernels write into an arena, so this pipeline does not wrap: rank-local rays, replicated model,
    dp_schedule.PipelinedExchange over arena.ParamArena (all-reduce of contiguous slices, pipelined across steps).

The optimiser state is shared, not duplicated: `Optimizers.optimizers[group].state[p]` holds VIEWS of the arena's moments, so
the reference's checkpoint code (engine/trainer.py:456-478) saves what the fused Adam produced, and a resumed run's
`load_optimizers` state is copied into the arena before the first iteration.

Anything the captured schedule does not cover (predict_normals, a non-Adam optimiser, weight decay, gradient clipping,
gradient accumulation, mixed precision, RGBA targets over a random background, CPU tensors) takes the reference's own
`get_train_loss_dict` body over the module path — with one rank; with more the pipeline refuses loudly, because neither DDP
nor the arena would then reduce the gradients.

`HipDynamicBatchPipeline` is the same seam for `instant-ngp-hip`: the reference's `DynamicBatchPipeline`
(pipelines/dynamic_batch.py:40-108 — it resizes the ray batch after every iteration from the number of samples the last one
kept) with `get_train_loss_dict` on the explicit packed-sample schedule (ngp_trainer.NgpTrainer over ngp_step.NgpTrainStep:
march -> candidates' density -> visibility scan + compaction -> field -> packed compositing -> MSE -> backward -> the arena's
fused Adam) instead of the module path's ~40 torch glue operations per iteration. The batch size changes from step to step
and the packed arrays are data-dependent in size (as in nerfacc), so this iteration is a sequence of eager launches over
capacity-sized buffers with the two sample counts as its host reads — the second of which IS the number the pipeline's
feedback needs, so `int(metrics["num_samples_per_batch"])` (dynamic_batch.py:91) costs no further synchronisation.

nerfstudio itself is imported lazily (`pipeline_classes()`, `ngp_pipeline_classes()`), as in plugin.py.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch

_LOSS_KEYS = ("rgb_loss", "interlevel_loss", "distortion_loss")


class _AlreadyBackpropagated(torch.autograd.Function):
    """Loss values of an iteration whose backward (and optimiser step) already ran inside the replayed graph: the trainer's
    `loss.backward()` (engine/trainer.py:515) finds a node and nothing to do."""

    @staticmethod
    def forward(ctx, anchor, *values):  # noqa: D102
        ctx.n_inputs = 1 + len(values)
        return tuple(v.clone() for v in values)

    @staticmethod
    def backward(ctx, *grads):  # noqa: D102
        return (None,) * ctx.n_inputs


class _LossValuesBackpropagated(torch.autograd.Function):
    """The same for the runner's loss-value buffer (train_step.loss_vals, five floats: rgb, interlevel, distortion, psnr, the
    distortion metric): ONE clone, handed out as five outputs of this node, so that the trainer's backward never meets a
    select node (whose backward launches a kernel)."""

    @staticmethod
    def forward(ctx, anchor, loss_vals):  # noqa: D102
        assert loss_vals.dim() == 1 and loss_vals.numel() >= 5
        out = loss_vals.clone()
        return out[0], out[1], out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, *grads):  # noqa: D102
        return None, None


def unsupported_model_reason(model) -> Optional[str]:
    """Configuration-level reasons the captured schedule cannot run this model (None: it can). use_gradient_scaling and
    per-edge jitter are covered by the explicit schedule (train_step.py); the normals options are module path only."""
    return "predict_normals" if getattr(model.config, "predict_normals", False) else None


class TrainEngine:
    """Owns the arena, the HipTrainer and the bookkeeping that ties them to a trainer's `Optimizers`."""

    EAGER_ITERATIONS = 2

    def __init__(self, pipeline, optimizers, trainer=None, runner_factory=None, on_build=None) -> None:
        self.pipeline, self.optimizers, self.host_trainer = pipeline, optimizers, trainer
        self.runner_factory = runner_factory  # tests: a CPU stand-in for train_step.NerfactoTrainStep
        self.on_build = on_build  # tests: called with the HipTrainer right after it was built (inject jitter draws)
        self.trainer = None  # trainer.HipTrainer, built on the first batch
        self.arena = None
        self.reason: Optional[str] = None
        self._lr_hist: Dict[str, Dict[int, float]] = {}
        self._anchor = None

    # ---- what the reference's Optimizers say ---------------------------------------------------------------------------
    def _optimizer_reason(self) -> Optional[str]:
        groups = self.optimizers.optimizers
        hyper = None
        for name, opt in groups.items():
            if type(opt) is not torch.optim.Adam:
                return f"optimizer of '{name}' is {type(opt).__name__}, not torch.optim.Adam"
            pg = opt.param_groups
            if len(pg) != 1:
                return f"'{name}' has {len(pg)} param_groups"
            g = pg[0]
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                return f"'{name}': weight decay / amsgrad / maximize"
            cfg = self.optimizers.config.get(name, {}).get("optimizer", None)
            if getattr(cfg, "max_norm", None) is not None:
                return f"'{name}': gradient clipping (max_norm)"
            h = (tuple(g["betas"]), float(g["eps"]))
            if hyper is not None and h != hyper:
                return "optimiser groups with different betas / eps"
            hyper = h
        ht = self.host_trainer
        if ht is not None:
            if getattr(ht, "mixed_precision", False):
                return "mixed precision"
            # engine/trainer.py:513-516: with an enabled scaler the trainer calls grad_scaler.scale(loss).backward() and
            # grad_scaler.update(), which asserts that an optimiser recorded inf checks — none does when the gradients live
            # in the arena
            scaler = getattr(ht, "grad_scaler", None)
            if getattr(ht, "use_grad_scaler", False) or (scaler is not None and scaler.is_enabled()):
                return "gradient scaler"
            if getattr(getattr(ht, "config", None), "log_gradients", False):
                return "log_gradients (param.grad stays None: the gradients live in the arena)"
            gas = getattr(ht, "gradient_accumulation_steps", None)
            if gas is not None and any(gas[k] != 1 for k in groups):
                return "gradient accumulation"
        return None

    def _lr(self, group: str, iteration: int) -> float:
        """Learning rate of `group` at `iteration`: what the reference's scheduler had set when that iteration ran (the
        deferred main-field Adam of iteration k-1 runs inside iteration k)."""
        hist = self._lr_hist.get(group, {})
        if iteration in hist:
            return hist[iteration]
        return float(self.optimizers.optimizers[group].param_groups[0]["lr"])

    def _record_lrs(self, step: int) -> None:
        for name, opt in self.optimizers.optimizers.items():
            hist = self._lr_hist.setdefault(name, {})
            hist[step] = float(opt.param_groups[0]["lr"])
            for old in [k for k in hist if k < step - 3]:
                del hist[old]

    # ---- construction on the first batch ---------------------------------------------------------------------------------
    def build(self, ray_bundle, batch) -> Optional[str]:
        from .arena import ParamArena

        model = self.pipeline.model
        reason = unsupported_model_reason(model) or self._optimizer_reason()
        o = ray_bundle.origins
        if reason is None and not o.is_cuda and self.runner_factory is None:
            reason = "rays on the CPU"
        if reason is None and batch["image"].shape[-1] == 4 and model.config.background_color == "random":
            reason = "RGBA targets over a random background"
        if reason is not None:
            self.reason = reason
            return reason
        params = self.optimizers.parameters
        order = [g for g in ("fields", "proposal_networks", "camera_opt") if g in params]
        extra = [g for g in params if g not in order]
        if extra:
            self.reason = f"parameter groups {extra} the schedule does not train"
            return self.reason
        first = self.optimizers.optimizers[order[0]].param_groups[0]
        self.arena = arena = ParamArena({g: list(params[g]) for g in order}, lr=float(first["lr"]), betas=tuple(first["betas"]),
                                        eps=float(first["eps"]), bind_grads=False)
        if self.pipeline.world_size > 1:
            arena.broadcast_params()  # what DDP does at wrap time
        self._adopt_optimizer_state()
        self._anchor = torch.zeros((), device=o.device, requires_grad=True)
        self.build_trainer_only(ray_bundle, batch)
        return None

    def _target(self, batch) -> torch.Tensor:
        """RGB target of the loss; RGBA is blended with the renderer's background as models/nerfacto.py:377-381 does."""
        model = self.pipeline.model
        image = batch["image"]
        if not image.is_cuda:
            image = image.to(next(model.parameters()).device)
        if image.shape[-1] == 4:
            image = model.renderer_rgb.blend_background(image)
        return image.reshape(-1, 3)

    def _adopt_optimizer_state(self) -> None:
        """torch.optim.Adam state <-> arena: moments a checkpoint loaded are copied in, then the state tensors become views of
        the arena's moments, and `state_dict()` refreshes the step counters (and applies any pending update) first."""
        arena = self.arena
        for name, opt in self.optimizers.optimizers.items():
            plist = opt.param_groups[0]["params"]
            steps = 0
            for p in plist:
                off = next(o for q, o in zip(arena.params, arena.offsets) if q is p)
                n = p.numel()
                st = opt.state.get(p, None)
                if st:  # resumed (engine/trainer.py:410-417 -> Optimizers.load_optimizers)
                    arena.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    arena.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps = max(steps, int(float(st["step"])))
                opt.state[p] = {"step": torch.tensor(float(steps)),
                                "exp_avg": arena.exp_avg[off:off + n].view(p.shape),
                                "exp_avg_sq": arena.exp_avg_sq[off:off + n].view(p.shape)}
            arena.step_counts[name] = steps
            original = opt.state_dict

            def state_dict(opt=opt, name=name, original=original):
                self.flush()
                for st in opt.state.values():
                    st["step"].fill_(float(self.arena.step_counts[name]))
                return original()

            opt.state_dict = state_dict

    # ---- per iteration ---------------------------------------------------------------------------------------------------
    def flush(self) -> None:
        """Apply a pending (deferred / pipelined) main-field update: call before anything reads the parameters."""
        if self.trainer is not None:
            self.trainer.finish()

    def train_iteration(self, step: int, ray_bundle, batch):
        if ray_bundle.origins.reshape(-1, 3).shape[0] != self.trainer.runner.n:
            self.flush()  # a datamanager that changes its batch size: new static buffers, new graphs
            self.build_trainer_only(ray_bundle, batch)
        t = self.trainer
        self._record_lrs(step)
        t.step = step
        # the first iterations run eagerly (they are the warm-up a capture needs, and every one of them is a real training
        # iteration on its own batch); then the schedule variants are captured once and replayed from there on
        self._eager_done = getattr(self, "_eager_done", 0)
        # More than one rank: eager segments unless NSAMD_DP_GRAPH=1 — captured segments with RCCL collectives between them
        # have only been rehearsed on a one-rank communicator (trainer.py docstring)
        may_capture = t.use_graph and (not t.dp or os.environ.get("NSAMD_DP_GRAPH", "0") == "1")
        if t.graphs is None and may_capture and self._eager_done >= self.EAGER_ITERATIONS and not getattr(t, "_capture_tried", False):
            t._capture_tried = True
            t.try_capture(warm=False)
        self._eager_done += 1
        t.set_batch(ray_bundle, {"image": self._target(batch)})
        t.train_iteration()
        r = t.runner
        model = self.pipeline.model
        outputs = r.outputs()
        if getattr(r, "_loss_vals_fresh", False) and model.config.background_color != "random":
            # nsamd_train_loss_values left the loss values and the training metrics in five floats (include/nsamd.h): one clone
            # per iteration instead of a dozen reduction launches
            rgb, inter, dist, psnr, dmetric = _LossValuesBackpropagated.apply(self._anchor, r.loss_vals)
            loss_dict = {"rgb_loss": rgb, "interlevel_loss": inter, "distortion_loss": dist}
            metrics = {"psnr": psnr.detach(), "distortion": dmetric.detach()}
            if r.cam_opt is not None:
                loss_dict["camera_opt_regularizer"] = _AlreadyBackpropagated.apply(self._anchor, r.camera_reg)[0]
        else:
            ld = r.loss_dict()
            values = _AlreadyBackpropagated.apply(self._anchor, *(ld[k] for k in _LOSS_KEYS))
            loss_dict = dict(zip(_LOSS_KEYS, values))
            if "camera_opt_regularizer" in ld:  # differentiated inside the iteration (train_step.backward_cameras)
                loss_dict["camera_opt_regularizer"] = _AlreadyBackpropagated.apply(self._anchor, ld["camera_opt_regularizer"])[0]
            mse = torch.mean((outputs["rgb"].detach() - r.target) ** 2)
            metrics = {"psnr": -10.0 * torch.log10(mse), "distortion": r.dist_per_ray.sum() / r.n}
        cam = getattr(model, "camera_optimizer", None)
        if cam is not None:
            cam.get_metrics_dict(metrics)
        return outputs, loss_dict, metrics

    def build_trainer_only(self, ray_bundle, batch) -> None:
        """Static buffers + schedule for batches of this many rays (the arena and the optimiser state stay)."""
        from .trainer import HipTrainer

        o = ray_bundle.origins.reshape(-1, 3)
        model = self.pipeline.model
        self._eager_done = 0
        runner = self.runner_factory(model, o.shape[0], o.device) if self.runner_factory is not None else None
        rb = ray_bundle.reshape(-1) if ray_bundle.origins.dim() > 2 else ray_bundle
        self.trainer = HipTrainer(model, self.arena, rb, {"image": self._target(batch)}, world=int(self.pipeline.world_size),
                                  use_graph=True, use_runner=True, pool=None, lr_source=self._lr, drive_callbacks=False,
                                  runner=runner)
        if hasattr(self.trainer.runner, "want_loss_vals"):
            self.trainer.runner.want_loss_vals = True  # the reference's trainer reads the loss dictionary every iteration
        if self.on_build is not None:
            self.on_build(self.trainer)


class NgpEngine(TrainEngine):
    """instant-ngp-hip: the arena (one optimiser group, models/instant_ngp.py:165-170) and ngp_trainer.NgpTrainer behind the
    trainer's `Optimizers`. Per iteration: the learning rate the reference's scheduler has set -> the arena's Adam; the
    parameters' `.grad` stay None — the packed kernels write into the arena's gradient views (`NgpTrainStep.grad_lookup`) — so that
    the trainer's own optimiser finds nothing to step (engine/optimizers.py:160-172).
    The occupancy refresh stays the model's BEFORE_TRAIN_ITERATION callback, run by the trainer."""

    def build(self, ray_bundle, batch) -> Optional[str]:
        from .arena import ParamArena

        model = self.pipeline.model
        reason = self._optimizer_reason()
        o = ray_bundle.origins
        if reason is None and getattr(model.config, "use_gradient_scaling", False):
            reason = "use_gradient_scaling"
        if reason is None and not o.is_cuda and self.runner_factory is None:
            reason = "rays on the CPU"
        if reason is None and batch["image"].shape[-1] == 4:
            reason = "RGBA targets"
        params = self.optimizers.parameters
        if reason is None and set(params) != {"fields"}:
            reason = f"parameter groups {sorted(params)} (the schedule trains 'fields' only)"
        if reason is not None:
            self.reason = reason
            return reason
        g = self.optimizers.optimizers["fields"].param_groups[0]
        self.arena = ParamArena({"fields": list(params["fields"])}, lr=float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]),
                                bind_grads=False)
        self._grad_views = [(p, self.arena.grad[off:off + p.numel()].view(p.shape)) for p, off in zip(self.arena.params, self.arena.offsets)]
        self._adopt_optimizer_state()
        self._anchor = torch.zeros((), device=o.device, requires_grad=True)
        try:
            self.build_trainer_only(ray_bundle, batch)
        except (RuntimeError, NotImplementedError) as e:
            # a model shape the explicit schedule does not cover (NgpTrainStep.__init__: hash-grid width, background mode ...):
            # the module path trains it, as for every other `reason`
            self.trainer = None
            self.reason = f"kernel schedule unavailable for this model ({type(e).__name__}: {e})"
            return self.reason
        return None

    def build_trainer_only(self, ray_bundle, batch) -> None:
        from .ngp_trainer import NgpTrainer

        o = ray_bundle.origins.reshape(-1, 3)
        model = self.pipeline.model
        runner = self.runner_factory(model, o.shape[0], o.device) if self.runner_factory is not None else None
        self.trainer = NgpTrainer(model, self.arena, o.shape[0], o.device, refresh=False, runner=runner)
        # the kernel schedule writes into the arena's gradient views directly; a stand-in runner (tests: autograd over the module
        # path) needs them as `param.grad` for the duration of the iteration
        self._bind_grads = not hasattr(self.trainer.runner, "grad_lookup")
        if not self._bind_grads:
            self.trainer.runner.grad_lookup = self.arena.grad_lookup()
        if self.on_build is not None:
            self.on_build(self.trainer)

    def flush(self) -> None:  # nothing is deferred on this schedule
        return None

    def train_iteration(self, step: int, ray_bundle, batch):
        t, a = self.trainer, self.arena
        a.lr = float(self.optimizers.optimizers["fields"].param_groups[0]["lr"])  # what the scheduler set for this iteration
        image = batch["image"]
        if not image.is_cuda and ray_bundle.origins.is_cuda:
            image = image.to(ray_bundle.origins.device)
        rb = ray_bundle.reshape(-1) if ray_bundle.origins.dim() > 2 else ray_bundle
        # Model.forward runs the collider in front of get_outputs (models/base_model.py:140-141): with enable_collider the
        # schedule must march the collider's near / far planes (NgpTrainer.set_batch reads the bundle's nears / fars)
        collider = getattr(self.pipeline.model, "collider", None)
        if collider is not None:
            rb = collider(rb)
        if self._bind_grads:
            for p, g in self._grad_views:
                p.grad = g
        try:
            t.set_batch(rb, {"image": image.reshape(-1, 3)})  # any number of rays: capacity-sized buffers (ngp_step.py)
            loss = t.train_iteration(step)
        finally:
            if self._bind_grads:
                for p, _ in self._grad_views:
                    p.grad = None
        r = t.runner
        outputs = r.outputs()
        loss_dict = {"rgb_loss": _AlreadyBackpropagated.apply(self._anchor, loss)[0]}
        # models/instant_ngp.py:218-224: psnr of the rendered colours against the (RGB) target; the number of kept samples —
        # already on the host (the schedule's second count read) — drives DynamicBatchPipeline's batch size
        mse = torch.mean((outputs["rgb"].detach() - r.target) ** 2)
        metrics = {"psnr": -10.0 * torch.log10(mse), "num_samples_per_batch": torch.tensor(int(r.num_kept))}
        return outputs, loss_dict, metrics


class EngineSeam:
    """What HipPipeline adds to VanillaPipeline, free of nerfstudio imports (the GPU tests compose it with a stand-in
    pipeline where the reference is absent). Expects `self.datamanager`, `self.model`, `self._model`, `self.world_size`."""

    _engine: Optional[TrainEngine] = None
    _engine_off: bool = False
    _engine_class = TrainEngine

    def attach_optimizers(self, optimizers, trainer=None, **engine_kwargs) -> None:
        """The trainer's `Optimizers` (engine/trainer.py:196-204 hands them to `get_training_callbacks`)."""
        if optimizers is not None and not self._engine_off:
            self._engine = self._engine_class(self, optimizers, trainer, **engine_kwargs)
            # readers that go through the MODEL, not the pipeline (the viewer renders with `pipeline.model.eval()` /
            # `get_outputs_for_camera`, viewer/render_state_machine.py) must see the pending main-field update too
            try:
                object.__setattr__(self.model, "_hip_flush", self.flush_engine)
            except Exception:  # noqa: BLE001 - a model that refuses attributes keeps the pipeline-level flush only
                pass

    def get_train_loss_dict(self, step: int):
        eng = self._engine
        if eng is None or eng.reason is not None:
            return self._module_path(step, *self.datamanager.next_train(step))
        ray_bundle, batch = self.datamanager.next_train(step)
        if eng.trainer is None and eng.build(ray_bundle, batch) is not None:
            return self._module_path(step, ray_bundle, batch)
        return eng.train_iteration(step, ray_bundle, batch)

    def _module_path(self, step, ray_bundle, batch):
        """The reference's own body (pipelines/base_pipeline.py:290-303) over the module path — one rank only."""
        if self.world_size > 1:
            raise NotImplementedError("nerfacto-hip with more than one rank: the captured schedule cannot run this setup "
                                      f"({getattr(self._engine, 'reason', None) or 'no optimizers were handed over'})")
        model_outputs = self._model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict

    def flush_engine(self) -> None:
        """Every reader of the parameters (evaluation, checkpoint) first applies a pending update."""
        if self._engine is not None:
            self._engine.flush()


_PIPELINE_CLASS_NAMES = ("HipPipelineConfig", "HipPipeline")
_NGP_PIPELINE_CLASS_NAMES = ("HipDynamicBatchPipelineConfig", "HipDynamicBatchPipeline")


def _publish(*classes):
    """Make classes that can only be built once nerfstudio is importable MODULE-LEVEL names: pickle (the reference's
    multi-GPU launch hands the TrainerConfig to `mp.spawn`, scripts/train.py:205) and yaml (`config.yml`, written by
    engine/trainer.py's `save_config` and read back by utils/eval_utils.py:90 for ns-eval / ns-render / ns-viewer /
    ns-export) name a class as `module.qualname` and resolve it with `getattr(module, name)`."""
    for cls in classes:
        cls.__module__ = __name__
        cls.__qualname__ = cls.__name__
        globals()[cls.__name__] = cls
    return classes


def __getattr__(name: str):  # PEP 562: a freshly spawned process / a yaml loader resolves the classes by name
    if name in _PIPELINE_CLASS_NAMES:
        pipeline_classes()
        return globals()[name]
    if name in _NGP_PIPELINE_CLASS_NAMES:
        ngp_pipeline_classes()
        return globals()[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def pipeline_classes():
    """(HipPipelineConfig, HipPipeline), built ONCE against the installed nerfstudio and published as attributes of this
    module (locally defined classes can neither be pickled nor found by yaml)."""
    if "HipPipeline" in globals():
        return globals()["HipPipelineConfig"], globals()["HipPipeline"]
    from dataclasses import dataclass, field
    from typing import Type

    import torch.distributed as dist
    from nerfstudio.pipelines.base_pipeline import Pipeline, VanillaPipeline, VanillaPipelineConfig

    class HipPipeline(EngineSeam, VanillaPipeline):
        """VanillaPipeline (pipelines/base_pipeline.py:206-460) with the training iteration on trainer.HipTrainer."""

        def __init__(self, config, device, test_mode="val", world_size=1, local_rank=0, grad_scaler=None):
            # what VanillaPipeline.__init__ builds (:231-282) — datamanager for this rank, model on the device — WITHOUT the
            # DistributedDataParallel wrapper: the arena's exchange reduces the gradients (module docstring)
            Pipeline.__init__(self)
            self.config, self.test_mode = config, test_mode
            self.datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size,
                                                        local_rank=local_rank)
            assert self.datamanager.train_dataset is not None, "Missing input dataset"
            seed_pts = None
            meta = getattr(getattr(self.datamanager, "train_dataparser_outputs", None), "metadata", None)
            if meta is not None and "points3D_xyz" in meta:
                seed_pts = (meta["points3D_xyz"], meta["points3D_rgb"])
            ds = self.datamanager.train_dataset
            self._model = config.model.setup(scene_box=ds.scene_box, num_train_data=len(ds), metadata=ds.metadata, device=device,
                                             grad_scaler=grad_scaler, seed_points=seed_pts)
            self.model.to(device)
            self.world_size = world_size
            self._engine = None
            self._engine_off = not getattr(config, "graph_train_step", True) or unsupported_model_reason(self.model) is not None
            if world_size > 1:
                if self._engine_off:
                    raise NotImplementedError(
                        "nerfacto-hip with more than one rank trains through the captured schedule and the arena's gradient "
                        f"exchange only; this configuration needs the module path ({unsupported_model_reason(self.model)})")
                dist.barrier(device_ids=[local_rank] if torch.cuda.is_available() else None)

        def get_training_callbacks(self, training_callback_attributes):
            self.attach_optimizers(getattr(training_callback_attributes, "optimizers", None),
                                   getattr(training_callback_attributes, "trainer", None))
            return super().get_training_callbacks(training_callback_attributes)

        def train(self, mode: bool = True):  # every evaluation entry point starts with `self.eval()` (:305-460)
            if not mode:
                self.flush_engine()
            return super().train(mode)

        def state_dict(self, *args, **kwargs):  # engine/trainer.py:456-478 save_checkpoint
            self.flush_engine()
            return super().state_dict(*args, **kwargs)

    @dataclass
    class HipPipelineConfig(VanillaPipelineConfig):
        _target: Type = field(default_factory=lambda: HipPipeline)
        graph_train_step: bool = True
        """Training iterations as replayed hipGraphs with the arena's fused Adam (nerfstudio_amd/trainer.py); False: the
        module path under the trainer's own optimisers."""

    _publish(HipPipelineConfig, HipPipeline)
    return HipPipelineConfig, HipPipeline


class DynamicBatchSeam(EngineSeam):
    """What HipDynamicBatchPipeline adds to DynamicBatchPipeline, free of nerfstudio imports (tests compose it with a stand-in
    where the reference is absent). Expects, besides EngineSeam's attributes, the reference class's
    `_update_dynamic_num_rays_per_batch` / `_update_pixel_samplers` and `datamanager.train_pixel_sampler`."""

    _engine_class = NgpEngine

    def get_train_loss_dict(self, step: int):
        eng = self._engine
        ray_bundle, batch = self.datamanager.next_train(step)
        if eng is None or eng.reason is not None or (eng.trainer is None and eng.build(ray_bundle, batch) is not None):
            model_outputs, loss_dict, metrics_dict = self._module_path(step, ray_bundle, batch)
        else:
            model_outputs, loss_dict, metrics_dict = eng.train_iteration(step, ray_bundle, batch)
        # pipelines/dynamic_batch.py:83-95: the next batch's number of rays from the samples this one kept
        if "num_samples_per_batch" not in metrics_dict:
            raise ValueError("'num_samples_per_batch' is not in metrics_dict."
                             "Please return 'num_samples_per_batch' in the models get_metrics_dict function to use this method.")
        self._update_dynamic_num_rays_per_batch(int(metrics_dict["num_samples_per_batch"]))
        self._update_pixel_samplers()
        assert "num_rays_per_batch" not in metrics_dict
        assert self.datamanager.train_pixel_sampler is not None
        metrics_dict["num_rays_per_batch"] = torch.tensor(self.datamanager.train_pixel_sampler.num_rays_per_batch)
        return model_outputs, loss_dict, metrics_dict

    def _module_path(self, step, ray_bundle, batch):
        """The reference's own body (pipelines/base_pipeline.py:290-303) over the module path (DDP-wrapped for N > 1)."""
        model_outputs = self._model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict


def ngp_pipeline_classes():
    """(HipDynamicBatchPipelineConfig, HipDynamicBatchPipeline), built once against the installed nerfstudio and published as
    attributes of this module (pickle / yaml resolve them by name, as `pipeline_classes`)."""
    if "HipDynamicBatchPipeline" in globals():
        return globals()["HipDynamicBatchPipelineConfig"], globals()["HipDynamicBatchPipeline"]
    from dataclasses import dataclass, field
    from typing import Type

    from nerfstudio.pipelines.dynamic_batch import DynamicBatchPipeline, DynamicBatchPipelineConfig

    class HipDynamicBatchPipeline(DynamicBatchSeam, DynamicBatchPipeline):
        """DynamicBatchPipeline (pipelines/dynamic_batch.py:40-108) with the training iteration on ngp_trainer.NgpTrainer.
        One rank: with more, VanillaPipeline wraps the model in DistributedDataParallel and training takes the module path."""

        def __init__(self, config, device, test_mode="val", world_size=1, local_rank=0, grad_scaler=None):
            DynamicBatchPipeline.__init__(self, config, device, test_mode, world_size, local_rank, grad_scaler)
            self._engine = None
            self._engine_off = not getattr(config, "kernel_schedule", True) or world_size > 1

        def get_training_callbacks(self, training_callback_attributes):
            self.attach_optimizers(getattr(training_callback_attributes, "optimizers", None),
                                   getattr(training_callback_attributes, "trainer", None))
            return super().get_training_callbacks(training_callback_attributes)

    @dataclass
    class HipDynamicBatchPipelineConfig(DynamicBatchPipelineConfig):
        _target: Type = field(default_factory=lambda: HipDynamicBatchPipeline)
        kernel_schedule: bool = True
        """Training iterations on the explicit packed-sample kernel schedule with the arena's fused Adam
        (nerfstudio_amd/ngp_step.py, ngp_trainer.py); False: the module path under the trainer's own optimiser."""

    _publish(HipDynamicBatchPipelineConfig, HipDynamicBatchPipeline)
    return HipDynamicBatchPipelineConfig, HipDynamicBatchPipeline
