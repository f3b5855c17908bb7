"""GPU box: the training iteration a nerfstudio trainer reaches through the `nerfacto-hip` pipeline seam
(pipeline.EngineSeam / TrainEngine -> trainer.HipTrainer: eager warm-up iterations, then replayed hipGraphs with the arena's
fused Adam) against the same trainer driven directly, as bench.py drives it — at the benchmark's own size.

The reference is absent on the GPU box: tests/trainer_restatement.py (pinned to the reference's `Trainer.train_iteration` /
`Optimizers` by tests/test_reference_trainer_drive.py) is the trainer; pipeline.HipPipeline itself — EngineSeam composed with
the reference's VanillaPipeline — runs under the reference's own trainer in tests/test_pipeline_seam_cpu.py."""
import collections
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import trainer_restatement as R  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from nerfstudio_amd import _native, functional

    _native.load()
    functional.DIRECT_GRAD = True
    yield functional
    functional.DIRECT_GRAD = False


def _opt_config(groups):
    cfg = {k: {"optimizer": {"lr": 1e-2, "eps": 1e-15}, "scheduler": {"lr_final": 1e-4, "max_steps": 200000}} for k in groups}
    if "camera_opt" in cfg:  # configs/method_configs.py:117-120
        cfg["camera_opt"] = {"optimizer": {"lr": 1e-3, "eps": 1e-15}, "scheduler": {"lr_final": 1e-4, "max_steps": 5000}}
    return cfg


def _fake_trainer(pipeline, optimizers):
    return SimpleNamespace(pipeline=pipeline, optimizers=optimizers, device="cuda:0", mixed_precision=False,
                           gradient_accumulation_steps=collections.defaultdict(lambda: 1),
                           grad_scaler=torch.amp.GradScaler("cuda", enabled=False), config=SimpleNamespace(log_gradients=False))


@pytest.mark.parametrize("camera", ["off", "SO3xR3"])
def test_trainer_drives_the_captured_iteration_through_the_seam_same_bits_as_direct(F, camera):
    import bench

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.pipeline import EngineSeam
    from nerfstudio_amd.trainer import HipTrainer

    dev = torch.device("cuda")
    K = 9
    n = bench.RAYS_PER_GPU
    rs = np.random.RandomState(5)
    jitter = torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)).to(dev)
    _, _, pool = bench.synthetic_batch(dev, seed=1000)

    def batch_of(step):
        s = step % bench.BATCH_SLOTS
        rb = RayBundle(origins=pool["origins"][s], directions=pool["directions"][s], pixel_area=torch.full((n, 1), 1e-6, device=dev),
                       camera_indices=pool["cameras"][s][:, None])
        return rb, {"image": pool["target"][s]}

    # ---- route 1: the restated reference trainer -> seam -> engine (2 eager iterations, then captured graphs)
    F._SCATTER_WS.clear()
    model = bench.build_model(dev, seed=0, camera_optimizer=camera)
    groups = model.get_param_groups()
    assert ("camera_opt" in groups) == (camera != "off")
    opts = R.Optimizers(_opt_config(groups), groups)

    class SeamPipeline(EngineSeam):
        def __init__(self):
            self.datamanager = SimpleNamespace(next_train=batch_of)
            self.model = self._model = model
            self.world_size = 1

    def on_build(t):
        t.draw_jitter = False
        t.runner.jitter.copy_(jitter)
        if t.runner.bg_rays is not None:
            t.runner.bg_rays.fill_(0.25)

    pipeline = SeamPipeline()
    trainer = _fake_trainer(pipeline, opts)
    pipeline.attach_optimizers(opts, trainer, on_build=on_build)
    losses = []
    for step in range(K):
        model.set_step(step)                      # BEFORE_TRAIN_ITERATION
        loss, loss_dict, metrics = R.train_iteration(trainer, step)
        model.after_step(step)                    # AFTER_TRAIN_ITERATION
        losses.append(float(loss))
        assert set(loss_dict) >= {"rgb_loss", "interlevel_loss", "distortion_loss"} and "psnr" in metrics
    eng = pipeline._engine
    assert eng.reason is None and eng.trainer.graphs is not None and eng.trainer.defer, "the seam must reach the captured schedule"
    assert all(p.grad is None for g in groups.values() for p in g), "the torch optimisers must find nothing to step"
    eng.flush()
    torch.cuda.synchronize()
    a1 = eng.arena
    got = (a1.flat.clone(), a1.exp_avg.clone(), a1.exp_avg_sq.clone(), dict(a1.step_counts))
    sd = opts.optimizers["fields"].state_dict()  # what the reference's checkpoint code would save
    assert all(float(s["step"]) == K for s in sd["state"].values())
    lrs = {g: [eng._lr_hist[g].get(i) for i in range(K)] for g in groups}
    del eng, pipeline, trainer, opts, model
    # ---- route 2: the trainer driven directly (eager launches, Adam in order), the learning rates of the same schedulers
    F._SCATTER_WS.clear()
    model = bench.build_model(dev, seed=0, camera_optimizer=camera)
    dummy = {g: [torch.nn.Parameter(torch.zeros(1))] for g in groups}
    sched = R.Optimizers(_opt_config(groups), dummy)
    table = {g: [] for g in groups}
    for i in range(K + 1):
        for g in groups:
            table[g].append(float(sched.optimizers[g].param_groups[0]["lr"]))
        sched.scheduler_step_all(i)
    for g in groups:  # (sanity: what route 1 recorded from the torch optimisers is this table)
        assert [x for x in lrs[g][K - 3:]] == table[g][K - 3:K], (g, lrs[g], table[g])
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    rb, batch = batch_of(0)
    tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=False, use_runner=True, pool=None,
                    lr_source=lambda g, it: table[g][it])
    on_build(tr)
    tr.runner.side_stream = None
    ref_losses = []
    for step in range(K):
        tr.set_batch(*batch_of(step))
        tr.train_iteration()
        ref_losses.append(float(tr.last_loss()))
    tr.finish()
    torch.cuda.synchronize()
    assert dict(arena.step_counts) == got[3]
    # the order of the optimiser groups in the arena differs between the routes (the reference's dictionary order vs main
    # field first): compare parameter by parameter
    assert torch.equal(_by_param(got[0], a1), _by_param(arena.flat, arena)), "parameters differ between the seam and the direct route"
    assert torch.equal(_by_param(got[1], a1), _by_param(arena.exp_avg, arena))
    assert torch.equal(_by_param(got[2], a1), _by_param(arena.exp_avg_sq, arena))
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-6)
    assert losses[-1] < losses[0]


def _by_param(flat, arena):
    """The arena's tensors in (group name, position) order — independent of the order the groups were laid out in."""
    parts = []
    for name in sorted(arena.group_params):
        for p in arena.group_params[name]:
            off = next(o for q, o in zip(arena.params, arena.offsets) if q is p)
            parts.append(flat[off:off + p.numel()])
    return torch.cat(parts)


def test_camera_optimizer_on_graph_replay_equals_eager(F):
    """The benched schedule with the reference's default camera optimiser (SO3xR3, models/nerfacto.py:131): the exponential
    map, its autograd backward and the camera group's Adam captured with the rest of the iteration; six replayed iterations
    equal eager launches bit for bit, and the pose parameters move. NSAMD_CAMERAS_OUTSIDE=1 (the N > 1 arrangement: those
    parts eagerly around the replay) must give the same bits again."""
    import bench

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.trainer import HipTrainer

    dev = torch.device("cuda")
    rs = np.random.RandomState(9)
    jitter = torch.from_numpy(rs.uniform(0, 1, (3, bench.RAYS_PER_GPU)).astype(np.float32)).to(dev)
    states = []
    for use_graph, outside in ((True, False), (False, False), (True, True)):
        F._SCATTER_WS.clear()
        model = bench.build_model(dev, seed=0, camera_optimizer="SO3xR3")
        arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
        assert "camera_opt" in arena.groups
        rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
        os.environ["NSAMD_CAMERAS_OUTSIDE"] = "1" if outside else "0"
        try:
            tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=use_graph, use_runner=True, pool=pool)
        finally:
            del os.environ["NSAMD_CAMERAS_OUTSIDE"]
        assert tr.cam_inside == (not outside) and tr.runner.cameras_outside == outside and tr.cam_group == "camera_opt"
        tr.draw_jitter = False
        tr.runner.jitter.copy_(jitter)
        if use_graph:
            tr.capture()
            assert tr.defer and len(tr.graphs) == 4
        else:
            tr.runner.side_stream = None
            tr.warm_variants()
        for _ in range(6):
            tr.train_iteration()
        tr.finish()
        torch.cuda.synchronize()
        pose = model.camera_optimizer.pose_adjustment.detach().clone()
        states.append((arena.flat.clone(), arena.exp_avg.clone(), arena.exp_avg_sq.clone(), pose, dict(arena.step_counts)))
        del tr, arena, model
    g, e, o = states
    assert g[4] == e[4] == o[4] and g[4]["camera_opt"] == 8  # 2 warm-up + 6 iterations
    assert float(g[3].abs().max()) > 0, "the pose corrections must train"
    for other, what in ((e, "eager launches"), (o, "the camera parts outside the graph")):
        for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq"), g[:3], other[:3]):
            assert torch.equal(x, y), f"{name}: {int((x != y).sum())} elements differ between graph replay and {what}"


def test_ngp_schedule_through_the_dynamic_batch_seam_same_bits_as_direct(F):
    """`instant-ngp-hip` (BASELINE configs[3]) behind ITS seam: the restated reference trainer -> pipeline.DynamicBatchSeam
    (what HipDynamicBatchPipeline adds to the reference's DynamicBatchPipeline, pipelines/dynamic_batch.py:40-108) ->
    pipeline.NgpEngine -> ngp_trainer.NgpTrainer over the explicit packed-sample schedule, the ray batch RESIZED after every
    iteration by the reference's rule from the samples the iteration kept — against ngp_trainer.NgpTrainer driven directly on
    the same batches with the same learning rates: same parameter and moment bits. HipDynamicBatchPipeline itself runs under
    the reference's own trainer and pipeline code in tests/test_ngp_pipeline_seam_cpu.py."""
    import bench
    from scripts.bench_ngp import build_ngp_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.ngp_trainer import NgpTrainer
    from nerfstudio_amd.pipeline import DynamicBatchSeam

    dev = torch.device("cuda")
    K, STEP0 = 7, 513  # (past the grid's warm-up; no refresh step among them: the refresh is the model's callback, tested elsewhere)
    TARGET, MAX_PER_RAY = 1 << 16, 1 << 5  # -> 2048 rays in the first batch
    parts = [bench.synthetic_rays(1000 + i) for i in range(4)]
    pool = [torch.from_numpy(np.concatenate([p[j] for p in parts])).to(dev) for j in range(4)]
    total = pool[0].shape[0]

    def batch_of_size(step, n):
        n = min(int(n), total)
        lo = (step * 997) % (total - n + 1)
        o, d, cam, tgt = (x[lo:lo + n] for x in pool)
        return RayBundle(origins=o, directions=d, pixel_area=torch.full((n, 1), 1e-6, device=dev), camera_indices=cam), {"image": tgt}

    # ---- route 1: the restated reference trainer -> the dynamic-batch seam -> the engine
    F._SCATTER_WS.clear()
    model, _ = build_ngp_model(dev)
    groups = {"fields": list(model.field.parameters())}
    opts = R.Optimizers(_opt_config(groups), groups)
    sizes = []

    class SeamPipeline(DynamicBatchSeam):
        def __init__(self):
            self.config = SimpleNamespace(target_num_samples=TARGET, max_num_samples_per_ray=MAX_PER_RAY)
            self.dynamic_num_rays_per_batch = TARGET // MAX_PER_RAY
            self.sampler = SimpleNamespace(num_rays_per_batch=self.dynamic_num_rays_per_batch)
            self.datamanager = SimpleNamespace(next_train=self.next_train, train_pixel_sampler=self.sampler)
            self.model = self._model = model
            self.world_size = 1

        def next_train(self, step):
            sizes.append(min(self.sampler.num_rays_per_batch, total))
            return batch_of_size(step, sizes[-1])

        def _update_dynamic_num_rays_per_batch(self, num_samples_per_batch):  # pipelines/dynamic_batch.py:71-76
            self.dynamic_num_rays_per_batch = int(self.dynamic_num_rays_per_batch * (self.config.target_num_samples / num_samples_per_batch))

        def _update_pixel_samplers(self):  # :64-69
            self.sampler.num_rays_per_batch = self.dynamic_num_rays_per_batch

    pipeline = SeamPipeline()
    trainer = _fake_trainer(pipeline, opts)
    pipeline.attach_optimizers(opts, trainer)
    torch.manual_seed(77)  # (the lattice offsets and the loss's random background are torch draws on the device, in launch order)
    losses, kept, lrs = [], [], []
    for i in range(K):
        lrs.append(float(opts.optimizers["fields"].param_groups[0]["lr"]))
        loss, loss_dict, metrics = R.train_iteration(trainer, STEP0 + i)
        losses.append(float(loss))
        kept.append(int(metrics["num_samples_per_batch"]))
        assert set(loss_dict) == {"rgb_loss"} and {"psnr", "num_samples_per_batch", "num_rays_per_batch"} <= set(metrics)
        assert int(metrics["num_rays_per_batch"]) == pipeline.sampler.num_rays_per_batch
    eng = pipeline._engine
    assert eng.reason is None and isinstance(eng.trainer, NgpTrainer) and eng.trainer.runner is not None
    assert all(p.grad is None for p in groups["fields"]), "the torch optimiser must find nothing to step"
    assert kept == eng.trainer.samples and len(set(sizes)) > 1, (sizes, kept)
    n = TARGET // MAX_PER_RAY
    for i in range(K):  # the reference's rule, fed by the schedule's own count
        assert sizes[i] == min(n, total), (i, sizes, kept)
        n = int(n * (TARGET / kept[i]))
    torch.cuda.synchronize()
    a1 = eng.arena
    got = (a1.flat.clone(), a1.exp_avg.clone(), a1.exp_avg_sq.clone(), dict(a1.step_counts))
    assert got[3] == {"fields": K}
    sd = opts.optimizers["fields"].state_dict()  # what the reference's checkpoint code would save
    assert all(float(s["step"]) == K for s in sd["state"].values())
    assert np.isfinite(losses).all()
    del eng, pipeline, trainer, opts, model
    # ---- route 2: the trainer driven directly on the same batches, the same learning rates
    F._SCATTER_WS.clear()
    model, _ = build_ngp_model(dev)
    arena = ParamArena({"fields": list(model.field.parameters())}, lr=1e-2, eps=1e-15)
    tr = NgpTrainer(model, arena, sizes[0], dev, refresh=False)
    torch.manual_seed(77)
    ref_losses = []
    for i in range(K):
        arena.lr = lrs[i]
        tr.set_batch(*batch_of_size(STEP0 + i, sizes[i]))
        ref_losses.append(float(tr.train_iteration(STEP0 + i)))
    torch.cuda.synchronize()
    assert tr.samples == kept
    assert torch.equal(got[0], arena.flat), "parameters differ between the seam and the direct route"
    assert torch.equal(got[1], arena.exp_avg) and torch.equal(got[2], arena.exp_avg_sq)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-6)
