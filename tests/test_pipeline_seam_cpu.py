"""`pipeline.HipPipeline` under the REFERENCE'S OWN trainer code on the CPU (needs /root/reference): the seam VERDICT r03
asked for — `Trainer.train_iteration` (engine/trainer.py:487-531) driving trainer.HipTrainer through
`get_train_loss_dict`, with the arena's fused Adam in place of the torch optimisers and, for two ranks over gloo, the
arena's pipelined gradient exchange in place of DistributedDataParallel (pipelines/base_pipeline.py:279-282).

The kernels are absent here: tests/cpu_runner.CpuRunner stands in for train_step.NerfactoTrainStep (gradients from the module
path over tests/cpu_kernels.py) and cpu_runner.cpu_adam for the Adam kernel — what is under test is the host logic: who steps
what and when, where the gradients and the optimiser state live, which learning rates are applied, what the ranks exchange."""
import collections
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdrive  # noqa: E402

needs_reference = pytest.mark.skipif(not refdrive.available(), reason="needs /root/reference")
N_RAYS, N_IMAGES, STEPS = 24, 7, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rays(rank, slot):
    from oracle import nerfacto_oracle as orc

    return orc.synthetic_rays(N_RAYS, N_IMAGES, seed=100 * rank + slot + 8)


class _Dataset:
    def __init__(self):
        from nerfstudio.data.scene_box import SceneBox

        self.scene_box, self.metadata = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), {}

    def __len__(self):
        return N_IMAGES


class _Datamanager(torch.nn.Module):
    """What VanillaPipeline needs of a datamanager: train_dataset, next_train, get_training_callbacks."""

    def __init__(self, config, device="cpu", test_mode="val", world_size=1, local_rank=0, **kw):
        super().__init__()
        self.train_dataset, self.rank, self.world_size, self.calls = _Dataset(), local_rank, world_size, []

    def next_train(self, step):
        from nerfstudio.cameras.rays import RayBundle

        self.calls.append(step)
        o, d, cam, tgt = _rays(self.rank, step % 3)
        rb = RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((N_RAYS, 1), 1e-6), camera_indices=cam[:, None])
        return rb, {"image": tgt.clone()}

    def get_training_callbacks(self, attrs):
        return []

    def get_param_groups(self):
        return {}


def _pipeline_config(graph_train_step=True):
    from dataclasses import dataclass, field
    from typing import Type

    from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio.configs.base_config import InstantiateConfig

    from nerfstudio_amd import plugin
    from nerfstudio_amd.pipeline import pipeline_classes

    @dataclass
    class DMConfig(InstantiateConfig):
        _target: Type = field(default_factory=lambda: _Datamanager)

    cfg_cls, _ = plugin._model_classes()
    pipe_cfg_cls, pipe_cls = pipeline_classes()
    args = [{"hidden_dim": 16, "log2_hashmap_size": 8, "num_levels": 5, "max_res": r, "use_linear": False} for r in (128, 256)]
    model_cfg = cfg_cls(log2_hashmap_size=10, proposal_net_args_list=args, camera_optimizer=CameraOptimizerConfig(mode="off"))
    return pipe_cfg_cls(datamanager=DMConfig(), model=model_cfg, graph_train_step=graph_train_step), pipe_cls


def _optimizers(pipeline):
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig

    groups = pipeline.get_param_groups()
    assert set(groups) == {"fields", "proposal_networks"}
    return Optimizers({k: {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                           "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=40)} for k in groups}, groups)


def _fake_trainer(pipeline, optimizers):
    return SimpleNamespace(pipeline=pipeline, optimizers=optimizers, device="cpu", mixed_precision=False,
                           gradient_accumulation_steps=collections.defaultdict(lambda: 1),
                           grad_scaler=torch.amp.GradScaler("cpu", enabled=False), config=SimpleNamespace(log_gradients=False))


def _train(pipeline, trainer, steps, world=1):
    """The reference trainer's loop body (engine/trainer.py:246-262): callbacks before / after, train_iteration between."""
    from nerfstudio.engine.callbacks import TrainingCallbackAttributes, TrainingCallbackLocation
    from nerfstudio.engine.trainer import Trainer

    attrs = TrainingCallbackAttributes(optimizers=trainer.optimizers, grad_scaler=trainer.grad_scaler, pipeline=pipeline, trainer=trainer)
    callbacks = pipeline.get_training_callbacks(attrs)
    losses = []
    for step in range(steps):
        for cb in callbacks:
            cb.run_callback_at_location(step, location=TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        loss, loss_dict, metrics = Trainer.train_iteration(trainer, step)
        for cb in callbacks:
            cb.run_callback_at_location(step, location=TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        losses.append(float(loss))
    return callbacks, losses


def _build(seed, world=1, rank=0, graph=True):
    import cpu_runner

    from nerfstudio_amd import functional as F

    F.adam_step = cpu_runner.cpu_adam
    cfg, pipe_cls = _pipeline_config(graph)
    torch.manual_seed(seed)
    pipeline = pipe_cls(cfg, device="cpu", world_size=world, local_rank=rank)
    pipeline.train()
    opts = _optimizers(pipeline)
    return pipeline, opts, _fake_trainer(pipeline, opts)


def _flat_params(model):
    return torch.cat([p.detach().reshape(-1) for p in list(model.field.parameters()) + list(model.proposal_networks.parameters())])


@needs_reference
def test_reference_trainer_drives_the_engine_through_the_pipeline_seam(monkeypatch):
    refdrive.install()
    import cpu_kernels
    import cpu_runner
    from nerfstudio.cameras.rays import RayBundle

    with cpu_kernels.installed(monkeypatch):
        # ---- (a) the engine behind the seam
        pipeline, opts, trainer = _build(seed=3)
        attach = pipeline.attach_optimizers
        monkeypatch.setattr(pipeline, "attach_optimizers", lambda o, t=None, **kw: attach(
            o, t, runner_factory=lambda m, n, dev: _runner(cpu_runner, m, n, dev, RayBundle, seed_base=50)))
        start = _flat_params(pipeline.model).clone()
        callbacks, losses = _train(pipeline, trainer, STEPS)
        eng = pipeline._engine
        assert eng is not None and eng.reason is None and eng.trainer is not None
        assert pipeline.datamanager.calls == list(range(STEPS))
        # gradients live in the arena, the torch optimisers found nothing to step, yet their state IS the arena's
        assert all(p.grad is None for g in opts.parameters.values() for p in g)
        assert np.isfinite(losses).all() and not torch.equal(start, _flat_params(pipeline.model))
        arena = eng.arena
        assert arena.step_counts["fields"] == STEPS and 0 < arena.step_counts["proposal_networks"] <= STEPS
        for name, opt in opts.optimizers.items():
            sd = opt.state_dict()
            st = sd["state"]
            assert len(st) == len(opts.parameters[name])
            assert all(float(s["step"]) == arena.step_counts[name] for s in st.values())
            p0 = opts.parameters[name][0]
            off0 = next(off for q, off in zip(arena.params, arena.offsets) if q is p0)
            assert opt.state[p0]["exp_avg"].data_ptr() == arena.exp_avg[off0:].data_ptr()
            assert float(opt.state[p0]["exp_avg_sq"].abs().sum()) > 0
        # the reference's schedulers computed the learning rates, the engine applied them
        lr_now = opts.optimizers["fields"].param_groups[0]["lr"]
        assert lr_now < 1e-2 and abs(eng._lr("fields", STEPS - 1) - 1e-2 * (1e-4 / 1e-2) ** ((STEPS - 1) / 40)) < 1e-9
        runner = eng.trainer.runner
        assert runner.calls.count("pfwd") == STEPS and runner.calls.count("bmain") == STEPS
        # ---- (b) the same training through the module path under the reference's own optimisers: same trajectory
        ref_pipe, ref_opts, ref_trainer = _build(seed=3, graph=False)
        assert ref_pipe._engine_off
        seeds = iter(range(50, 50 + STEPS))
        model_call = ref_pipe._model.forward
        monkeypatch.setattr(ref_pipe._model, "forward", lambda rb: (torch.manual_seed(next(seeds)), model_call(rb))[1])
        _, ref_losses = _train(ref_pipe, ref_trainer, STEPS)
        assert ref_pipe._engine is None
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
        a, b = _flat_params(pipeline.model), _flat_params(ref_pipe.model)
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), float((a - b).abs().max())
        # ---- (c) a checkpoint of the engine's run resumes into a fresh pipeline with the optimiser state adopted
        ckpt = {"pipeline": pipeline.state_dict(), "optimizers": {k: v.state_dict() for k, v in opts.optimizers.items()}}
        p2, o2, t2 = _build(seed=99)
        p2.load_pipeline(ckpt["pipeline"], STEPS - 1)
        o2.load_optimizers(ckpt["optimizers"])
        attach2 = p2.attach_optimizers
        monkeypatch.setattr(p2, "attach_optimizers", lambda o, t=None, **kw: attach2(
            o, t, runner_factory=lambda m, n, dev: _runner(cpu_runner, m, n, dev, RayBundle, seed_base=70)))
        assert torch.equal(_flat_params(p2.model), a)
        _train(p2, t2, 1)
        ar2 = p2._engine.arena
        assert ar2.step_counts["fields"] == STEPS + 1 and float(ar2.exp_avg.abs().sum()) > 0


def _runner(cpu_runner, model, n, dev, bundle_cls, seed_base):
    r = cpu_runner.CpuRunner(model, n, dev, seed_base=seed_base)
    r._bundle_like = bundle_cls.__new__(bundle_cls)
    return r


# ---------------------------------------------------------------------------------------------------------------------
# two ranks over gloo: the plugin pipeline, not only the arena
# ---------------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        refdrive.install()
        import cpu_kernels
        import cpu_runner
        from nerfstudio.cameras.rays import RayBundle

        with pytest.MonkeyPatch.context() as mp:
            with cpu_kernels.installed(mp):
                pipeline, opts, trainer = _build(seed=3 + rank, world=world, rank=rank)  # different init per rank: broadcast fixes it
                assert type(pipeline._model).__name__ == "HipNerfactoModel"  # not wrapped in DistributedDataParallel
                attach = pipeline.attach_optimizers
                mp.setattr(pipeline, "attach_optimizers", lambda o, t=None, **kw: attach(
                    o, t, runner_factory=lambda m, n, dev: _runner(cpu_runner, m, n, dev, RayBundle, seed_base=1000 * rank)))
                _, losses = _train(pipeline, trainer, STEPS, world)
                pipeline.eval()  # flushes the pending (pipelined) main-field update
                eng = pipeline._engine
                assert eng.trainer.exchange is not None and not eng.trainer.exchange.pending
                q.put((rank, _flat_params(pipeline.model).numpy(), losses, dict(eng.arena.step_counts),
                       eng.trainer.runner.calls))
    finally:
        dist.destroy_process_group()


@needs_reference
@pytest.mark.timeout(600)
def test_two_ranks_over_gloo_through_the_plugin_pipeline(monkeypatch):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, l0, c0, calls0), (_, p1, l1, c1, _) = res
    np.testing.assert_array_equal(p0, p1)  # replicated model after every exchange
    assert c0 == c1 and c0["fields"] == STEPS
    assert l0 != l1  # rank-local rays
    assert calls0[:3] == ["pfwd", ("main", True), "bmain"]  # the pipelined schedule's segments
    # ---- sequential data-parallel semantics, single process: mean of the two ranks' gradients, one Adam per group
    refdrive.install()
    import cpu_kernels
    import cpu_runner
    from nerfstudio.cameras.rays import RayBundle

    from nerfstudio_amd import functional as F
    from nerfstudio_amd.arena import ParamArena

    with cpu_kernels.installed(monkeypatch):
        pipeline, opts, _ = _build(seed=3)  # rank 0's initial parameters (what the broadcast distributed)
        model = pipeline.model
        arena = ParamArena({g: list(opts.parameters[g]) for g in ("fields", "proposal_networks")}, lr=1e-2, eps=1e-15, bind_grads=False)
        runners = [_runner(cpu_runner, model, N_RAYS, "cpu", RayBundle, seed_base=1000 * r) for r in range(world)]
        dms = [_Datamanager(None, local_rank=r) for r in range(world)]
        lookup = arena.grad_lookup()
        ps = model.proposal_sampler
        for step in range(STEPS):
            model.set_step(step) if hasattr(model, "set_step") else None
            frac = np.clip(step / model.config.proposal_weights_anneal_max_num_iters, 0, 1)  # models/nerfacto.py:270-280
            b = model.config.proposal_weights_anneal_slope
            ps.set_anneal(float(b * frac / ((b - 1) * frac + 1)))
            updated = ps.updated_this_step()
            arena.grad.zero_()
            for r, (runner, dm) in enumerate(zip(runners, dms)):
                rb, batch = dm.next_train(step)
                runner.grad_lookup = lookup
                runner.set_batch(rb.origins, rb.directions, rb.camera_indices, batch["image"])
                runner.forward_main_and_losses(updated)
                runner.backward_all(updated)
            lr = 1e-2 * (1e-4 / 1e-2) ** (step / 40)
            for name in (("fields", "proposal_networks") if updated else ("fields",)):
                a, e = arena.groups[name]
                arena.step_counts[name] += 1
                hyper = torch.tensor(F.adam_hyper(arena.step_counts[name], lr, arena.betas), dtype=torch.float32)
                cpu_runner.cpu_adam(arena.flat[a:e], arena.grad[a:e], arena.exp_avg[a:e], arena.exp_avg_sq[a:e],
                                    arena.step_counts[name], lr, arena.betas, arena.eps, 1.0 / world, hyper)
            if updated:
                ps.mark_updated()
            ps.step_cb(step)
        expect = _flat_params(model).numpy()
    assert float(np.abs(p0 - expect).max()) <= 1e-6 * float(np.abs(expect).max()), float(np.abs(p0 - expect).max())
