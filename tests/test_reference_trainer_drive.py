"""The `nerfacto-hip` plugin model under the REFERENCE'S OWN trainer code (VERDICT r02 item 8 / missing 3).

Authoring container (needs /root/reference; tests/refdrive answers the absent third-party imports with placeholders):
  * `plugin.HipNerfactoModel` — a real subclass of the reference's NerfactoModel, built by the reference's constructor — is
    driven by the reference's unmodified `Trainer.train_iteration` (engine/trainer.py:487-531) ->
    `VanillaPipeline.get_train_loss_dict` (pipelines/base_pipeline.py:290-303) with the reference's `Optimizers`
    (engine/optimizers.py:74-193) over the model's own `get_param_groups()`: on CPU tensors the iteration gets through the
    datamanager hand-over, the model call, the reference's CameraOptimizer and the HIP sampler mirror down to the kernel
    launch guard ("runs on an MI355X only") — no AttributeError / TypeError on the way — with and without
    config.fused_train_step.
  * tests/trainer_restatement.py (what the GPU tests use where the reference is absent) is pinned to that code: a toy
    model with the Model API trained by both for 6 iterations (two optimiser groups, ExponentialDecay schedulers, a group
    that receives no gradient on some steps) ends at the same parameter bits and learning rates.
GPU box: the package's NerfactoModel through the restated iteration (module path and fused_train_step)."""
import collections
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdrive  # noqa: E402
import trainer_restatement as R  # noqa: E402

needs_reference = pytest.mark.skipif(not refdrive.available(), reason="needs /root/reference")


def _fake_trainer(pipeline, optimizers, device):
    return SimpleNamespace(pipeline=pipeline, optimizers=optimizers, device=device, mixed_precision=False,
                           gradient_accumulation_steps=collections.defaultdict(lambda: 1),
                           grad_scaler=torch.amp.GradScaler(device.split(":")[0], enabled=False),
                           config=SimpleNamespace(log_gradients=False))


class _Datamanager:
    def __init__(self, batches):
        self.batches, self.calls = batches, []

    def next_train(self, step):
        self.calls.append(step)
        return self.batches[step % len(self.batches)]


class _ToyModel(torch.nn.Module):
    """The Model API on plain torch: two parameter groups; the second one gets gradient only on even steps (as the
    proposal networks do on non-update steps)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(6, 3, generator=g) * 0.3)
        self.b = torch.nn.Parameter(torch.randn(3, generator=g) * 0.3)
        self.step = 0

    def get_param_groups(self):
        return {"fields": [self.a], "proposal_networks": [self.b]}

    def forward(self, ray_bundle):
        x = ray_bundle["x"]
        b = self.b if self.step % 2 == 0 else self.b.detach()
        self.step += 1
        return {"rgb": torch.sigmoid(x @ self.a + b)}

    def get_metrics_dict(self, outputs, batch):
        return {"psnr": -10 * torch.log10(torch.mean((outputs["rgb"].detach() - batch["image"]) ** 2))}

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        return {"rgb_loss": torch.mean((outputs["rgb"] - batch["image"]) ** 2), "reg": 1e-3 * (self.a ** 2).sum()}


@needs_reference
def test_restated_train_iteration_equals_the_reference_trainer():
    refdrive.install()
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import Trainer
    from nerfstudio.pipelines.base_pipeline import VanillaPipeline

    g = torch.Generator().manual_seed(1)
    batches = [({"x": torch.randn(32, 6, generator=g)}, {"image": torch.rand(32, 3, generator=g)}) for _ in range(4)]
    results = []
    for which in ("reference", "restatement"):
        model = _ToyModel()
        if which == "reference":
            opts = Optimizers({k: {"optimizer": AdamOptimizerConfig(lr=lr, eps=1e-15),
                                   "scheduler": ExponentialDecaySchedulerConfig(lr_final=lr / 100, max_steps=20)}
                               for k, lr in (("fields", 1e-2), ("proposal_networks", 5e-3))}, model.get_param_groups())
            pipeline = object.__new__(VanillaPipeline)  # the reference's class, without its datamanager-building ctor
            torch.nn.Module.__init__(pipeline)
            pipeline.datamanager, pipeline._model, pipeline.world_size = _Datamanager(batches), model, 1
            trainer = _fake_trainer(pipeline, opts, "cpu")
            step_fn = lambda s: Trainer.train_iteration(trainer, s)  # noqa: E731  (the reference's own function)
        else:
            opts = R.Optimizers({k: {"optimizer": {"lr": lr, "eps": 1e-15}, "scheduler": {"lr_final": lr / 100, "max_steps": 20}}
                                 for k, lr in (("fields", 1e-2), ("proposal_networks", 5e-3))}, model.get_param_groups())
            pipeline = SimpleNamespace(datamanager=_Datamanager(batches), _model=model, model=model)
            trainer = _fake_trainer(pipeline, opts, "cpu")
            step_fn = lambda s: R.train_iteration(trainer, s)  # noqa: E731
        losses = [float(step_fn(s)[0]) for s in range(6)]
        lrs = {k: o.param_groups[0]["lr"] for k, o in opts.optimizers.items()}
        results.append((model.a.detach().clone(), model.b.detach().clone(), losses, lrs, pipeline.datamanager.calls))
    (a0, b0, l0, lr0, c0), (a1, b1, l1, lr1, c1) = results
    assert torch.equal(a0, a1) and torch.equal(b0, b1) and l0 == l1 and c0 == c1 == list(range(6))
    assert lr0 == lr1 and lr0["fields"] < 1e-2


@needs_reference
@pytest.mark.parametrize("fused", [False, True])
def test_reference_trainer_drives_the_plugin_model_down_to_the_kernel_guard(fused):
    refdrive.install()
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import Trainer
    from nerfstudio.models.nerfacto import NerfactoModel
    from nerfstudio.pipelines.base_pipeline import VanillaPipeline

    from nerfstudio_amd import plugin

    cfg_cls, model_cls = plugin._model_classes()
    assert issubclass(model_cls, NerfactoModel)
    args = [{"hidden_dim": 16, "log2_hashmap_size": 8, "num_levels": 5, "max_res": r, "use_linear": False} for r in (64, 128)]
    cfg = cfg_cls(log2_hashmap_size=10, proposal_net_args_list=args, fused_train_step=fused)
    model = model_cls(config=cfg, scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=7, metadata={})
    model.train()
    groups = model.get_param_groups()
    assert set(groups) == {"fields", "proposal_networks", "camera_opt"}
    opts = Optimizers({k: {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                           "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=200000)} for k in groups}, groups)
    n = 64
    g = torch.Generator().manual_seed(0)
    rb = RayBundle(origins=torch.randn(n, 3, generator=g) * 0.3,
                   directions=torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1),
                   pixel_area=torch.full((n, 1), 1e-6), camera_indices=torch.randint(0, 7, (n, 1), generator=g))
    dm = _Datamanager([(rb, {"image": torch.rand(n, 3, generator=g)})])
    pipeline = object.__new__(VanillaPipeline)
    torch.nn.Module.__init__(pipeline)
    pipeline.datamanager, pipeline._model, pipeline.world_size = dm, model, 1
    trainer = _fake_trainer(pipeline, opts, "cpu")
    with pytest.raises(RuntimeError, match="MI355X"):  # the whole reference call chain, stopped by the launch guard only
        Trainer.train_iteration(trainer, 0)
    assert dm.calls == [0]


# ---------------------------------------------------------------------------------------------------------------------
# GPU box: the package's own model through the restated iteration
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_restated_trainer_drives_the_hip_model_on_the_gpu(fused):
    """3+ iterations of (zero_grad(set_to_none) -> get_train_loss_dict -> backward -> Adam per group -> scheduler) over
    `nerfstudio_amd.nerfacto.NerfactoModel`, module path and fused_train_step: finite falling loss, every group stepped on
    the steps it received gradient, the fused path ends where the module path ends (same kernels) to 1e-5."""
    from test_gpu_kernels import small_cfg

    from nerfstudio_amd import _native
    from nerfstudio_amd.cameras.rays import RayBundle
    from oracle import nerfacto_oracle as orc

    _native.load()
    cfg = small_cfg(12, 10, 6)
    n, steps = 256, 8
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=2)
    rs = np.random.RandomState(0)
    jit = torch.from_numpy(rs.uniform(0, 1, (steps, 3, n, 1)).astype(np.float32)).cuda()
    finals = {}
    for mode in ((False, True) if fused else (False,)):
        from test_gpu_kernels import _hip_model

        model = _hip_model(cfg, orc.init_params(cfg, seed=5, table_std=0.3))
        model.config.fused_train_step = mode
        groups = model.get_param_groups()
        opts = R.Optimizers({k: {"optimizer": {"lr": 1e-2, "eps": 1e-15}, "scheduler": {"lr_final": 1e-4, "max_steps": 200000}}
                             for k in groups}, groups)
        state = {"k": 0}

        class DM:
            def next_train(self, step):
                rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                               camera_indices=cam.cuda()[:, None])
                state["k"] = step
                return rb, {"image": tgt.cuda()}

        class Wrapped(torch.nn.Module):  # injects the jitter draws so that both modes see the same random numbers
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, rb):
                return self.m(rb, jitters=[jit[state["k"], i] for i in range(3)])

        pipeline = SimpleNamespace(datamanager=DM(), _model=Wrapped(model), model=model)
        trainer = _fake_trainer(pipeline, opts, "cuda:0")
        losses = []
        for step in range(steps):
            model.set_step(step)
            losses.append(float(R.train_iteration(trainer, step)[0]))
            model.after_step(step)
        assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
        assert all(int(o_.state[p]["step"]) == steps for o_ in (opts.optimizers["fields"],) for p in o_.param_groups[0]["params"])
        finals[mode] = (torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone(), losses)
    if fused:
        a, b = finals[False], finals[True]
        np.testing.assert_allclose(a[1], b[1], rtol=1e-4)
        assert float((a[0] - b[0]).abs().max()) <= 1e-4 * max(1.0, float(a[0].abs().max()))


@needs_reference
def test_reference_profiler_hooks_open_roctx_ranges(monkeypatch):
    """profiler.hook_reference_profiler(): the reference's OWN `@profiler.time_function` hooks — train_iteration
    (engine/trainer.py:486) and get_train_loss_dict (pipelines/base_pipeline.py:289) — open / close roctx ranges around the
    reference's unmodified code."""
    refdrive.install()
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.trainer import Trainer
    from nerfstudio.pipelines.base_pipeline import VanillaPipeline

    from nerfstudio_amd.utils import profiler as P

    calls = []

    class FakeRoctx:
        def roctxRangePushA(self, name):
            calls.append(("push", name.decode()))
            return 0

        def roctxRangePop(self):
            calls.append(("pop",))
            return 0

    monkeypatch.setattr(P, "_ROCTX", FakeRoctx())
    assert P.hook_reference_profiler() and P.hook_reference_profiler()  # idempotent
    P.enable_ranges(True)
    try:
        model = _ToyModel()
        opts = Optimizers({k: {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15), "scheduler": None}
                           for k in ("fields", "proposal_networks")}, model.get_param_groups())
        g = torch.Generator().manual_seed(1)
        pipeline = object.__new__(VanillaPipeline)
        torch.nn.Module.__init__(pipeline)
        pipeline.datamanager = _Datamanager([({"x": torch.randn(8, 6, generator=g)}, {"image": torch.rand(8, 3, generator=g)})])
        pipeline._model, pipeline.world_size = model, 1
        Trainer.train_iteration(_fake_trainer(pipeline, opts, "cpu"), 0)
    finally:
        P.enable_ranges(False)
    names = [c[1] for c in calls if c[0] == "push"]
    assert names == ["Trainer.train_iteration", "VanillaPipeline.get_train_loss_dict"], names
    assert [c[0] for c in calls] == ["push", "push", "pop", "pop"]


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the plugin model under the reference's OWN model code, end to end, with the kernel wrappers replaced by the oracle's
# torch restatements (tests/cpu_kernels.py) — everything between the launch guard and the loss that no GPU-less run reaches
# otherwise: the reference's get_outputs indexing this package's field outputs, its renderers' call signatures, its loss
# dictionary, autograd through the mirrored modules into the reference's Optimizers.
# ---------------------------------------------------------------------------------------------------------------------
def _plugin_model(predict_normals=False, n_images=7):
    from nerfstudio.data.scene_box import SceneBox

    from nerfstudio_amd import plugin
    from oracle import nerfacto_oracle as orc

    cfg_cls, model_cls = plugin._model_classes()
    args = [{"hidden_dim": 16, "log2_hashmap_size": 8, "num_levels": 5, "max_res": r, "use_linear": False} for r in (128, 256)]
    cfg = cfg_cls(log2_hashmap_size=10, proposal_net_args_list=args, predict_normals=predict_normals)
    model = model_cls(config=cfg, scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=n_images,
                      metadata={})
    ocfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 10),
                           prop_grids=(orc.HashGridCfg(5, 16, 128, 8), orc.HashGridCfg(5, 16, 256, 8)), num_images=n_images)
    ocfg.predict_normals = predict_normals
    # (the class defaults of the reference's config, not the `nerfacto` method's overrides: average_init_density = 1)
    ocfg.average_init_density = float(cfg.average_init_density)
    assert (cfg.interlevel_loss_mult, cfg.distortion_loss_mult, cfg.background_color) == (1.0, 0.002, "last_sample")
    params = orc.init_params(ocfg, seed=31, table_std=0.5)
    sd = {k: v.clone() for k, v in params.items()}
    for i in range(2):
        sd[f"proposal_networks.{i}.mlp_base.0.hash_table"] = sd[f"proposal_networks.{i}.encoding.hash_table"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(any(s in m for s in ("aabb", "max_res", "num_levels", "log2_hashmap_size", "camera_optimizer", "lpips", "psnr",
                                    "ssim", "collider", "device_indicator")) for m in missing), missing
    return model, ocfg, params


@needs_reference
def test_reference_model_code_runs_the_plugin_end_to_end_on_cpu_stand_ins(monkeypatch):
    refdrive.install()
    import cpu_kernels
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import Trainer
    from nerfstudio.pipelines.base_pipeline import VanillaPipeline

    from oracle import nerfacto_oracle as orc

    model, ocfg, params = _plugin_model()
    model.train()
    n = 24
    o, d, cam, tgt = orc.synthetic_rays(n, ocfg.num_images, seed=8)

    def bundle():
        return RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None])

    with cpu_kernels.installed(monkeypatch):
        # ---- one forward / loss / backward through the reference's NerfactoModel code, against the oracle on the same draws
        torch.manual_seed(3)
        jit = [torch.rand((n, 1)) for _ in range(3)]  # the sampler's draws, in its order (ray_samplers.py:105, :322)
        torch.manual_seed(3)
        out = model(bundle())  # Model.forward: the reference's collider, then ITS get_outputs over this package's modules
        batch = {"image": tgt}
        metrics = model.get_metrics_dict(out, batch)
        losses = model.get_loss_dict(out, batch, metrics)
        assert {"rgb_loss", "interlevel_loss", "distortion_loss"} <= set(losses)
        sum(losses.values()).backward()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        ref = orc.nerfacto_forward(p, ocfg, o, d, cam, jit, training=True, anneal=float(model.proposal_sampler._anneal))
        lr = orc.nerfacto_losses(ref, tgt, ocfg)
        sum(lr.values()).backward()
        np.testing.assert_allclose(out["rgb"].detach().numpy(), ref["rgb"].detach().numpy(), atol=1e-6)
        np.testing.assert_allclose(out["accumulation"].detach().numpy(), ref["accumulation"].detach().numpy(), atol=1e-6)
        np.testing.assert_allclose(out["depth"].numpy(), ref["depth"].numpy(), atol=1e-5)
        for k in ("rgb_loss", "interlevel_loss", "distortion_loss"):
            np.testing.assert_allclose(float(losses[k].detach()), float(lr[k].detach()), rtol=1e-5, atol=1e-12, err_msg=k)
        named = dict(model.named_parameters())
        for k in ("field.mlp_base.model.0.hash_table", "field.mlp_head.layers.0.weight", "field.embedding_appearance.embedding.weight",
                  "proposal_networks.0.encoding.hash_table", "proposal_networks.1.mlp_base.1.layers.0.weight"):
            a, b = named[k].grad.numpy(), p[k].grad.numpy()
            assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(b), 1e-30), k
        # ---- two iterations of the reference's trainer over it: every group steps, the loss is finite
        groups = model.get_param_groups()
        opts = Optimizers({k: {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                               "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=200000)} for k in groups}, groups)
        pipeline = object.__new__(VanillaPipeline)
        torch.nn.Module.__init__(pipeline)
        dm = _Datamanager([None])
        dm.next_train = lambda step: (bundle(), batch)  # a fresh bundle per step (the camera optimiser edits it in place)
        pipeline.datamanager, pipeline._model, pipeline.world_size = dm, model, 1
        trainer = _fake_trainer(pipeline, opts, "cpu")
        before = named["field.mlp_head.layers.0.weight"].detach().clone()
        vals = [float(Trainer.train_iteration(trainer, s)[0]) for s in range(2)]
        assert np.isfinite(vals).all() and not torch.equal(before, named["field.mlp_head.layers.0.weight"].detach())
        # ---- the reference's chunked eval render over the plugin (models/base_model.py:178-205)
        model.eval()
        img = model.get_outputs_for_camera_ray_bundle(
            RayBundle(origins=o.reshape(4, 6, 3), directions=d.reshape(4, 6, 3), pixel_area=torch.full((4, 6, 1), 1e-6),
                      camera_indices=cam.reshape(4, 6, 1)))
        assert img["rgb"].shape == (4, 6, 3) and img["depth"].shape == (4, 6, 1) and float(img["rgb"].min()) >= 0


@needs_reference
def test_reference_model_code_runs_the_plugin_with_predicted_normals(monkeypatch):
    """config.predict_normals under the reference's own get_outputs / get_loss_dict (models/nerfacto.py:304, :325-344, :379-388):
    its NormalsRenderer / NormalsShader / orientation_loss / pred_normal_loss over this package's composed field."""
    refdrive.install()
    import cpu_kernels
    from nerfstudio.cameras.rays import RayBundle

    from oracle import nerfacto_oracle as orc

    model, ocfg, params = _plugin_model(predict_normals=True)
    assert "field.mlp_pred_normals.layers.2.weight" in dict(model.named_parameters())
    model.train()
    n = 12
    o, d, cam, tgt = orc.synthetic_rays(n, ocfg.num_images, seed=9)
    with cpu_kernels.installed(monkeypatch):
        torch.manual_seed(4)
        jit = [torch.rand((n, 1)) for _ in range(3)]
        torch.manual_seed(4)
        out = model(RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None]))
        batch = {"image": tgt}
        losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
        assert {"orientation_loss", "pred_normal_loss"} <= set(losses) and out["normals"].shape == (n, 3)
        sum(losses.values()).backward()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        ref = orc.nerfacto_forward(p, ocfg, o, d, cam, jit, training=True, anneal=float(model.proposal_sampler._anneal))
        lr = orc.nerfacto_losses(ref, tgt, ocfg)
        np.testing.assert_allclose(out["normals"].detach().numpy(), ref["normals"].detach().numpy(), atol=1e-5)
        np.testing.assert_allclose(out["pred_normals"].detach().numpy(), ref["pred_normals"].detach().numpy(), atol=1e-5)
        for k in ("rgb_loss", "orientation_loss", "pred_normal_loss"):
            np.testing.assert_allclose(float(losses[k].detach()), float(lr[k].detach()), rtol=1e-4, atol=1e-12, err_msg=k)
        assert dict(model.named_parameters())["field.mlp_pred_normals.layers.0.weight"].grad.abs().sum() > 0


@needs_reference
@pytest.mark.parametrize("options", [
    {"use_gradient_scaling": True}, {"use_single_jitter": False}, {"disable_scene_contraction": True},
    {"use_same_proposal_network": True}, {"background_color": "random"}, {"background_color": "white"},
    {"use_appearance_embedding": False}, {"camera_optimizer": "off"}], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_reference_model_code_runs_the_plugin_under_its_config_options(monkeypatch, options):
    """Every NerfactoModelConfig switch the plugin passes on, under the reference's own train forward / losses / backward
    and eval forward (CPU stand-ins for the kernels): no KeyError / TypeError / shape error anywhere on the way."""
    refdrive.install()
    import cpu_kernels
    from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox

    from nerfstudio_amd import plugin
    from oracle import nerfacto_oracle as orc

    cfg_cls, model_cls = plugin._model_classes()
    args = [{"hidden_dim": 16, "log2_hashmap_size": 8, "num_levels": 5, "max_res": r, "use_linear": False} for r in (128, 256)]
    kw = dict(log2_hashmap_size=10, proposal_net_args_list=args)
    kw.update(options)
    if options.get("use_same_proposal_network"):
        kw["proposal_net_args_list"] = args[:1]
    if options.get("camera_optimizer") == "off":
        kw["camera_optimizer"] = CameraOptimizerConfig(mode="off")
    model = model_cls(config=cfg_cls(**kw), scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=7,
                      metadata={})
    n = 12
    o, d, cam, tgt = orc.synthetic_rays(n, 7, seed=8)

    def bundle():
        return RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None])

    with cpu_kernels.installed(monkeypatch):
        model.train()
        out = model(bundle())
        batch = {"image": tgt}
        losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
        total = sum(losses.values())
        total.backward()
        assert bool(torch.isfinite(total)) and model.field.mlp_head.layers[0].weight.grad.abs().sum() > 0
        model.eval()
        with torch.no_grad():
            ev = model(bundle())
        assert ev["rgb"].shape == (n, 3) and "weights_list" not in ev


@needs_reference
def test_registry_discovers_both_methods_through_the_declared_entry_points(monkeypatch):
    """plugins/registry.py:34-78 run for real: the ENTRY-POINT branch keeps only objects that already are
    MethodSpecification instances (it does not call callables — only the NERFSTUDIO_METHOD_CONFIGS branch does), so
    pyproject.toml names lazily built module attributes; both branches must register `nerfacto-hip` and `instant-ngp-hip`
    with the reference's recipes around this package's model classes."""
    refdrive.install()
    from importlib.metadata import EntryPoint

    import tomli
    from nerfstudio.models.instant_ngp import NGPModel
    from nerfstudio.models.nerfacto import NerfactoModel
    from nerfstudio.pipelines.dynamic_batch import DynamicBatchPipelineConfig
    from nerfstudio.plugins import registry
    from nerfstudio.plugins.types import MethodSpecification

    declared = tomli.load(open(os.path.join(os.path.dirname(HERE), "pyproject.toml"), "rb"))["project"]["entry-points"][
        "nerfstudio.method_configs"]
    assert set(declared) == {"nerfacto-hip", "instant-ngp-hip"}

    class _EntryPoints:
        def __init__(self, eps):
            self._eps = {e.name: e for e in eps}
            self.names = set(self._eps)

        def __getitem__(self, name):
            return self._eps[name]

    eps = _EntryPoints([EntryPoint(name=k, value=v, group="nerfstudio.method_configs") for k, v in declared.items()])
    monkeypatch.setattr(registry, "entry_points", lambda group: eps)
    monkeypatch.delenv("NERFSTUDIO_METHOD_CONFIGS", raising=False)
    methods, descriptions = registry.discover_methods()
    assert set(methods) == {"nerfacto-hip", "instant-ngp-hip"} and all(descriptions[k] for k in methods)
    assert all(isinstance(eps[k].load(), MethodSpecification) for k in declared)
    nf, ngp = methods["nerfacto-hip"], methods["instant-ngp-hip"]
    assert issubclass(nf.pipeline.model._target, NerfactoModel) and nf.mixed_precision is False
    assert nf.pipeline.model.average_init_density == 0.01 and set(nf.optimizers) == {"proposal_networks", "fields", "camera_opt"}
    assert issubclass(ngp.pipeline.model._target, NGPModel) and isinstance(ngp.pipeline, DynamicBatchPipelineConfig)
    assert ngp.pipeline.model.eval_num_rays_per_chunk == 8192 and set(ngp.optimizers) == {"fields"} and ngp.mixed_precision is False
    # the environment-variable branch takes the same attributes (and would also call the builder functions)
    monkeypatch.setattr(registry, "entry_points", lambda group: _EntryPoints([]))
    monkeypatch.setenv("NERFSTUDIO_METHOD_CONFIGS", "nerfacto-hip=nerfstudio_amd.plugin:nerfacto_hip,"
                                                    "instant-ngp-hip=nerfstudio_amd.plugin:instant_ngp_hip_spec")
    methods2, _ = registry.discover_methods()
    assert set(methods2) == {"nerfacto-hip", "instant-ngp-hip"}


@needs_reference
@pytest.mark.parametrize("background", ["random", "white"])
def test_reference_ngp_model_code_runs_the_instant_ngp_plugin_on_cpu_stand_ins(monkeypatch, background):
    """`instant-ngp-hip`: plugin.HipNGPModel — a subclass built by the REFERENCE's NGPModel constructor (nerfacc answered by
    a placeholder here; the reference's populate_modules builds its estimator, install_hip_ngp_modules replaces it) — under
    the reference's own Model.forward, training callback (occupancy refresh, models/instant_ngp.py:150-163), get_metrics_dict
    (num_samples_per_batch, what DynamicBatchPipeline reads, pipelines/dynamic_batch.py:71-95), get_loss_dict, backward,
    Optimizers step and chunked eval render; kernels replaced by the oracle's restatements; outputs equal to the oracle's
    on the samples the sampler placed."""
    refdrive.install()
    import cpu_kernels
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.engine.callbacks import TrainingCallbackAttributes, TrainingCallbackLocation
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.models.instant_ngp import NGPModel
    from nerfstudio.pipelines.dynamic_batch import DynamicBatchPipeline

    from nerfstudio_amd import plugin
    from nerfstudio_amd.model_components.occupancy import OccGridEstimator
    from oracle import nerfacto_oracle as orc
    from oracle import packed_oracle as po

    cfg_cls, model_cls = plugin._ngp_model_classes()
    assert issubclass(model_cls, NGPModel)
    cfg = cfg_cls(grid_resolution=16, grid_levels=2, log2_hashmap_size=10, background_color=background, cone_angle=0.0,
                  render_step_size=0.05)
    model = model_cls(config=cfg, scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=4, metadata={})
    assert isinstance(model.occupancy_grid, OccGridEstimator) and model.sampler.occupancy_grid is model.occupancy_grid
    ocfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 10), prop_grids=(), num_images=4, average_init_density=1.0)
    params = orc.init_params(ocfg, seed=27, table_std=0.5)
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in params.items() if k.startswith("field.")}, strict=False)
    assert not unexpected, unexpected
    model.train()
    groups = model.get_param_groups()
    assert set(groups) == {"fields"}
    opts = Optimizers({"fields": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                                  "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=200000)}}, groups)
    n = 24
    o, d, cam, tgt = orc.synthetic_rays(n, 4, seed=12)
    batch = {"image": tgt}

    def bundle():
        return RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None])

    with cpu_kernels.installed(monkeypatch):
        callbacks = model.get_training_callbacks(TrainingCallbackAttributes(optimizers=opts, grad_scaler=None, pipeline=None, trainer=None))
        assert len(callbacks) == 1 and TrainingCallbackLocation.BEFORE_TRAIN_ITERATION in callbacks[0].where_to_run
        callbacks[0].run_callback_at_location(step=0, location=TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        assert 0 < int(model.occupancy_grid.binaries.sum()) <= model.occupancy_grid.binaries.numel()
        torch.manual_seed(6)
        jit = torch.rand(n)  # the sampler's stratified offsets (ray_samplers.py:489)
        torch.manual_seed(6)
        out = model(bundle())  # the reference's Model.forward (no collider for this method) -> get_outputs
        metrics = model.get_metrics_dict(out, batch)
        assert int(metrics["num_samples_per_batch"]) == int(out["num_samples_per_ray"].sum()) > n
        # DynamicBatchPipeline's feedback on it (the reference's own method, called unbound on a stand-in pipeline)
        pipe = SimpleNamespace(dynamic_num_rays_per_batch=64, config=SimpleNamespace(target_num_samples=1 << 12, max_num_samples_per_ray=1 << 10),
                               datamanager=SimpleNamespace(train_pixel_sampler=SimpleNamespace(set_num_rays_per_batch=lambda v: None),
                                                           eval_pixel_sampler=SimpleNamespace(set_num_rays_per_batch=lambda v: None)))
        DynamicBatchPipeline._update_dynamic_num_rays_per_batch(pipe, int(metrics["num_samples_per_batch"]))
        assert pipe.dynamic_num_rays_per_batch == int(64 * ((1 << 12) / int(metrics["num_samples_per_batch"])))
        torch.manual_seed(7)  # the loss's random background (renderers.py:195)
        losses = model.get_loss_dict(out, batch, metrics)
        assert set(losses) == {"rgb_loss"}
        losses["rgb_loss"].backward()
        # ---- the oracle on the samples the sampler placed
        B = model.occupancy_grid.binaries.numpy().astype(bool)
        ri, ts, te = po.occgrid_march(o.numpy(), d.numpy(), B, [-1.0, -1, -1, 1, 1, 1], cfg.render_step_size, near_plane=cfg.near_plane,
                                      far_plane=cfg.far_plane, cone_angle=0.0, jitter=jit.numpy())
        ri, ts, te = torch.from_numpy(ri).long(), torch.from_numpy(ts), torch.from_numpy(te)
        pos = o[ri] + d[ri] * ((ts + te) / 2)[:, None]
        with torch.no_grad():
            sig = orc.nerfacto_field(pos, d[ri], cam[ri], params, ocfg, training=True)[0]
            keep = po.render_visibility_from_density(ts, te, sig, ri, n, 1e-4, min(cfg.alpha_thre, model.occupancy_grid._occ_mean))
        ri, ts, te = ri[keep], ts[keep], te[keep]
        assert int(out["num_samples_per_ray"].sum()) == ri.numel()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        dens, rgb_s, _ = orc.nerfacto_field(o[ri] + d[ri] * ((ts + te) / 2)[:, None], d[ri], cam[ri], p, ocfg, training=True)
        w = po.render_weight_from_density(ts, te, dens, ri, n)[0]
        comp, acc, dep = po.composite_packed(rgb_s, w, ts, te, ri, n, background=background, training=True)
        np.testing.assert_allclose(out["rgb"].detach().numpy(), comp.detach().numpy(), atol=1e-6)
        np.testing.assert_allclose(out["accumulation"].detach().numpy(), acc.detach().numpy(), atol=1e-6)
        np.testing.assert_allclose(out["depth"].detach().numpy(), dep.detach().numpy(), rtol=1e-5, atol=1e-6)
        pred = comp
        if background == "random":
            torch.manual_seed(7)
            pred = comp + torch.rand_like(comp) * (1.0 - acc)
        ref_loss = torch.mean((tgt - pred) ** 2)
        np.testing.assert_allclose(float(losses["rgb_loss"].detach()), float(ref_loss.detach()), rtol=1e-5)
        ref_loss.backward()
        named = dict(model.named_parameters())
        for k in ("field.mlp_base.model.0.hash_table", "field.mlp_head.layers.0.weight", "field.embedding_appearance.embedding.weight"):
            a, b = named[k].grad.numpy(), p[k].grad.numpy()
            assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(b), 1e-30), k
        before = named["field.mlp_head.layers.0.weight"].detach().clone()
        opts.optimizer_step_all()  # engine/optimizers.py:158-172
        assert not torch.equal(before, named["field.mlp_head.layers.0.weight"].detach())
        # ---- the reference's chunked eval render over it
        model.eval()
        img = model.get_outputs_for_camera_ray_bundle(
            RayBundle(origins=o.reshape(4, 6, 3), directions=d.reshape(4, 6, 3), pixel_area=torch.full((4, 6, 1), 1e-6),
                      camera_indices=cam.reshape(4, 6, 1)))
        assert img["rgb"].shape == (4, 6, 3) and img["depth"].shape == (4, 6, 1)


@needs_reference
@pytest.mark.parametrize("predict_normals", [False, True])
def test_plugin_model_against_the_reference_model_itself(monkeypatch, predict_normals):
    """The REFERENCE's NerfactoModel (torch implementation, its own random initialisation) and plugin.HipNerfactoModel:
    the two state dicts have the same keys, shapes and dtypes and load into each other with strict=True (what
    Model.load_model / Trainer._load_checkpoint do, models/base_model.py:226-232); with the reference's weights loaded, one
    training forward / losses / backward of the plugin (kernel wrappers = the oracle's restatements) reproduces the
    reference model's outputs, loss terms and gradients on the same rays and draws."""
    refdrive.install()
    import cpu_kernels
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.models.nerfacto import NerfactoModel, NerfactoModelConfig

    from nerfstudio_amd import plugin
    from oracle import nerfacto_oracle as orc

    args = [{"hidden_dim": 16, "log2_hashmap_size": 8, "num_levels": 5, "max_res": r, "use_linear": False} for r in (128, 256)]
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
    torch.manual_seed(0)
    ref = NerfactoModel(config=NerfactoModelConfig(log2_hashmap_size=10, proposal_net_args_list=args, implementation="torch",
                                                   predict_normals=predict_normals, average_init_density=0.01),
                        scene_box=box, num_train_data=7, metadata={})
    cfg_cls, model_cls = plugin._model_classes()
    mine = model_cls(config=cfg_cls(log2_hashmap_size=10, proposal_net_args_list=args, predict_normals=predict_normals,
                                    average_init_density=0.01), scene_box=box, num_train_data=7, metadata={})
    a, b = ref.state_dict(), mine.state_dict()
    assert set(a) == set(b) and all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    with torch.no_grad():  # the hash tables start at 1e-4 scale: lift them so that densities and gradients are not degenerate
        for k, v in a.items():
            if k.endswith("hash_table"):
                v.mul_(3000.0)
    ref.load_state_dict(a, strict=True)
    mine.load_state_dict(a, strict=True)
    ref.load_state_dict(mine.state_dict(), strict=True)
    n = 16
    o, d, cam, tgt = orc.synthetic_rays(n, 7, seed=14)
    batch = {"image": tgt}

    def run(model):
        model.train()
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        out = model(RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None]))
        losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
        sum(losses.values()).backward()
        return out, losses, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    out_r, loss_r, grad_r = run(ref)
    with cpu_kernels.installed(monkeypatch):
        out_m, loss_m, grad_m = run(mine)
    for k in ("rgb", "accumulation", "depth", "expected_depth") + (("normals", "pred_normals") if predict_normals else ()):
        # (north_star's 1e-4: the oracle's double-accumulated scans against the reference's ATen ones move the odd resampled
        # bin edge by an ulp; depths are in scene units)
        # (the analytic normals jump at the cell boundaries of the hash grid: tests/test_normals.py)
        atol = 2e-3 if "depth" in k else (1e-2 if k == "normals" else (2e-3 if k == "pred_normals" else 1e-4))
        np.testing.assert_allclose(out_m[k].detach().numpy(), out_r[k].detach().numpy(), atol=atol, rtol=1e-4, err_msg=k)
    assert set(loss_m) == set(loss_r)
    for k in loss_r:
        np.testing.assert_allclose(float(loss_m[k].detach()), float(loss_r[k].detach()), rtol=1e-2 if "normal" in k or "orientation" in k else 2e-3,
                                   atol=1e-10, err_msg=k)
    assert set(grad_m) == set(grad_r)
    for k in grad_r:
        ga, gb = grad_m[k].numpy(), grad_r[k].numpy()
        assert np.linalg.norm(ga - gb) <= (3e-2 if predict_normals else 1e-2) * max(np.linalg.norm(gb), 1e-30) + 1e-12, \
            (k, np.linalg.norm(ga - gb), np.linalg.norm(gb))


@needs_reference
def test_method_configs_survive_pickle_and_yaml(monkeypatch):
    """the reference's multi-GPU launch pickles the TrainerConfig into `mp.spawn` (scripts/train.py:205) and
    every run writes `config.yml` with yaml.dump, which ns-eval / ns-render / ns-viewer / ns-export read back with
    yaml.load(Loader=yaml.Loader) (utils/eval_utils.py:90). Both name a class as module.qualname: the lazily built pipeline /
    model classes must be module-level names of this package, built once, and resolvable in a process that has not built
    them yet."""
    refdrive.install()
    import pickle
    import subprocess

    import yaml

    from nerfstudio_amd import pipeline, plugin

    for spec in (plugin.nerfacto_hip(), plugin.instant_ngp_hip()):
        cfg = spec.config
        again = pickle.loads(pickle.dumps(cfg))
        assert type(again.pipeline) is type(cfg.pipeline) and type(again.pipeline.model) is type(cfg.pipeline.model)
        assert again.pipeline.model._target is cfg.pipeline.model._target and again.method_name == cfg.method_name
        text = yaml.dump(cfg)
        loaded = yaml.load(text, Loader=yaml.Loader)
        assert type(loaded.pipeline) is type(cfg.pipeline) and loaded.pipeline.model._target is cfg.pipeline.model._target
        assert loaded.pipeline._target is cfg.pipeline._target
    # built once: a second call hands back the same classes (a TrainerConfig built twice compares equal by type)
    assert pipeline.pipeline_classes()[1] is pipeline.pipeline_classes()[1] is pipeline.HipPipeline
    assert plugin._model_classes()[1] is plugin.HipNerfactoModel and plugin._ngp_model_classes()[1] is plugin.HipNGPModel
    assert pipeline.HipPipelineConfig.__module__ == "nerfstudio_amd.pipeline" and pipeline.HipPipeline.__qualname__ == "HipPipeline"
    # a fresh interpreter (what a spawned rank is) unpickles the nerfacto-hip config without having built anything
    blob = pickle.dumps(plugin.nerfacto_hip().config)
    code = ("import sys, pickle; sys.path[:0] = %r; import refdrive; refdrive.install(); "
            "cfg = pickle.loads(sys.stdin.buffer.read()); "
            "print(type(cfg.pipeline).__qualname__, cfg.pipeline._target.__qualname__, cfg.pipeline.model._target.__qualname__)"
            % [HERE, os.path.dirname(HERE)])
    out = subprocess.run([sys.executable, "-c", code], input=blob, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert out.stdout.decode().split() == ["HipPipelineConfig", "HipPipeline", "HipNerfactoModel"]
