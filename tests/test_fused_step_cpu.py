"""Host logic of fused_step.FusedTrainStep and NerfactoTrainStep.prepare_grads on the CPU (no kernels: the runner is a
recording stand-in). What the GPU test (tests/test_gpu_kernels.py::test_fused_train_step_behind_the_model_api) cannot
isolate: the order of the runner calls, the update-schedule bookkeeping, the unit-weight contract of the loss terms, and
the gradient-buffer semantics a trainer's `zero_grad(set_to_none=True)` relies on (engine/optimizers.py:160-172)."""
from types import SimpleNamespace

import pytest
import torch

from nerfstudio_amd.cameras.camera_optimizers import CameraOptimizerConfig
from nerfstudio_amd.cameras.rays import RayBundle
from nerfstudio_amd.fused_step import FusedTrainStep
from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig
from nerfstudio_amd.train_step import NerfactoTrainStep


def small_model(camera_mode="off"):
    cfg = NerfactoModelConfig(log2_hashmap_size=8, camera_optimizer=CameraOptimizerConfig(mode=camera_mode),
                              proposal_net_args_list=[{"hidden_dim": 16, "log2_hashmap_size": 7, "num_levels": 5, "max_res": r,
                                                       "use_linear": False} for r in (32, 64)])
    return NerfactoModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 7).train()


class FakeRunner:
    """Records the calls FusedTrainStep makes; buffers like the real runner's, loss values fixed."""

    def __init__(self, model, n):
        self.model, self.n, self.calls = model, n, []
        self.anneal_dev = torch.ones(1)
        self.jitter = torch.zeros(3, n)
        self.target = torch.zeros(n, 3)
        self.dist_per_ray = torch.full((n,), 0.5)
        self.rgb = torch.rand(n, 3)
        self.reg_in_backward = True

    def set_batch(self, o, d, cams, target=None):
        self.calls.append(("set_batch", tuple(o.shape), target is None))

    def apply_camera_corrections(self):
        self.calls.append(("corrections",))

    def forward_proposals(self, draw_jitter=True, need_enc=True):
        self.calls.append(("proposals", draw_jitter, need_enc))

    def forward_main(self):
        self.calls.append(("main",))

    def losses(self, updated):
        self.calls.append(("losses", updated, float(self.target.sum())))

    def outputs(self):
        return {"rgb": self.rgb, "accumulation": torch.ones(self.n, 1), "expected_depth": torch.ones(self.n, 1),
                "depth": torch.ones(self.n, 1), "weights_list": []}

    def loss_dict(self):
        return {"rgb_loss": torch.tensor(0.25), "interlevel_loss": torch.tensor(0.125), "distortion_loss": torch.tensor(0.001)}

    def prepare_grads(self, updated):
        self.calls.append(("prepare_grads", updated))

    def backward_all(self, updated):
        self.calls.append(("backward_all", updated))


def bundle(n):
    return RayBundle(origins=torch.zeros(n, 3), directions=torch.ones(n, 3), pixel_area=torch.ones(n, 1),
                     camera_indices=torch.zeros(n, 1, dtype=torch.long))


def fused_with_fake(model, n):
    fs = FusedTrainStep(model)
    fs.runner = FakeRunner(model, n)
    fs._runner_for = lambda num_rays, device: fs.runner  # noqa: ARG005
    return fs


def test_call_order_losses_and_backward():
    model, n = small_model(), 32
    fs = fused_with_fake(model, n)
    assert fs.supported() is None
    model.set_step(0)  # step < 10: the proposal networks are updated (ray_samplers.py:590)
    out = fs.get_outputs(bundle(n))
    r = fs.runner
    assert [c[0] for c in r.calls] == ["set_batch", "corrections", "proposals", "main", "losses"]
    assert r.calls[0] == ("set_batch", (n, 3), True)  # no target yet: Model.get_outputs does not see the batch
    assert r.calls[2] == ("proposals", True, True) and r.calls[4][1] is True
    assert out["fused_step"] is fs and fs.updated
    assert model.proposal_sampler._steps_since_update == 0  # mark_updated, as generate_ray_samples does
    batch = {"image": torch.full((n, 3), 2.0)}
    losses = fs.get_loss_dict(out, batch)
    assert r.calls[-1] == ("losses", True, float(6.0 * n))  # redone with the real target in place
    assert set(losses) == {"rgb_loss", "interlevel_loss", "distortion_loss"}
    assert float(losses["rgb_loss"].detach()) == 0.25 and all(v.requires_grad for v in losses.values())
    metrics = fs.get_metrics_dict(out, batch)
    assert float(metrics["distortion"]) == 0.5 and "psnr" in metrics
    import functools

    functools.reduce(torch.add, losses.values()).backward()  # engine/trainer.py:514
    assert r.calls[-2:] == [("prepare_grads", True), ("backward_all", True)]
    # the anchor parameter gets nothing from autograd: the (fake) runner owns the gradient buffers
    assert model.field.mlp_base.encoding.hash_table.grad is None


def test_update_schedule_and_forced_variants():
    model, n = small_model(), 16
    fs = fused_with_fake(model, n)
    ps = model.proposal_sampler
    ps._step, ps._steps_since_update = 137, 1  # 1 > update_sched(137) = 1 is false: a non-update iteration
    fs.get_outputs(bundle(n))
    assert not fs.updated and fs.runner.calls[2] == ("proposals", True, False) and ps._steps_since_update == 1
    # a caller that replays captured schedule variants forces the decision and keeps the bookkeeping to itself (bench.py)
    ps.force_updated = True
    fs.runner.calls.clear()
    fs.get_outputs(bundle(n))
    assert fs.updated and ps._steps_since_update == 1
    ps.force_updated = None
    # injected jitter (parity tests): copied into the runner's buffer, no draw on the device
    fs.runner.calls.clear()
    jit = [torch.full((n, 1), 0.1 * (i + 1)) for i in range(3)]
    fs.get_outputs(bundle(n), jitters=jit)
    assert fs.runner.calls[2][1] is False
    assert torch.equal(fs.runner.jitter[2], torch.full((n,), 0.1 * 3))


def test_non_unit_loss_weights_are_rejected():
    model, n = small_model(), 8
    fs = fused_with_fake(model, n)
    model.set_step(0)
    out = fs.get_outputs(bundle(n))
    losses = fs.get_loss_dict(out, {"image": torch.zeros(n, 3)})
    with pytest.raises(RuntimeError, match="unit weights"):
        (2.0 * losses["rgb_loss"] + losses["interlevel_loss"] + losses["distortion_loss"]).backward()
    assert not any(c[0] == "backward_all" for c in fs.runner.calls)
    with pytest.raises(AssertionError):
        fs.get_loss_dict({"fused_step": object()}, {"image": torch.zeros(n, 3)})  # outputs of another forward


def test_camera_regulariser_goes_through_autograd():
    model, n = small_model("SO3xR3"), 8
    with torch.no_grad():
        model.camera_optimizer.pose_adjustment.add_(0.01)
    fs = fused_with_fake(model, n)
    model.set_step(0)
    out = fs.get_outputs(bundle(n))
    losses = fs.get_loss_dict(out, {"image": torch.zeros(n, 3)})
    assert "camera_opt_regularizer" in losses
    import functools

    functools.reduce(torch.add, losses.values()).backward()
    g = model.camera_optimizer.pose_adjustment.grad
    assert g is not None and float(g.abs().max()) > 0  # the regulariser's own gradient; the rays' share is the runner's
    assert fs.runner.calls[-1] == ("backward_all", True)


def test_model_routes_through_the_fused_step_only_when_asked_and_training():
    model = small_model()
    assert model._fused_step() is None  # config.fused_train_step is off by default
    model.config.fused_train_step = True
    assert isinstance(model._fused_step(), FusedTrainStep)
    with torch.no_grad():
        assert model._fused_step() is None  # evaluation inside no_grad stays on the module path
    model.eval()
    assert model._fused_step() is None
    model.train()
    # options the explicit schedule takes since round 3 (per-edge jitter, gradient scaling) do not reject any more; what is
    # still outside it says so loudly
    model.config.use_single_jitter = False
    model.config.use_gradient_scaling = True
    model._fused = None
    assert isinstance(model._fused_step(), FusedTrainStep)
    model.config.predict_normals = True
    model._fused = None
    with pytest.raises(NotImplementedError, match="predict_normals"):
        model._fused_step()


def test_prepare_grads_semantics():
    """Fresh buffers are zero-filled except the main table's (the scatter writes it); proposal networks only on update
    iterations; an existing table gradient switches the scatter to accumulation."""
    model = small_model()
    stub = SimpleNamespace(model=model, main_table_write_only=True)
    table = model.field.mlp_base.encoding.hash_table
    NerfactoTrainStep.prepare_grads(stub, updated=False)
    assert stub.main_table_write_only and table.grad is not None and table.grad.shape == table.shape
    for name, p in model.field.named_parameters():
        assert p.grad is not None, name
        if p is not table:
            assert float(p.grad.abs().max()) == 0.0, name
    assert all(p.grad is None for p in model.proposal_networks.parameters())  # no gradient this iteration: Adam skips them
    assert NerfactoTrainStep.written_params(stub) == [table]
    # second call without zero_grad: everything is accumulated into, the table included
    table.grad.fill_(1.0)
    NerfactoTrainStep.prepare_grads(stub, updated=True)
    assert not stub.main_table_write_only and float(table.grad.min()) == 1.0
    assert NerfactoTrainStep.written_params(stub) == []
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in model.proposal_networks.parameters())
    # zero_grad(set_to_none=True) brings the write-only path back
    model.zero_grad(set_to_none=True)
    NerfactoTrainStep.prepare_grads(stub, updated=False)
    assert stub.main_table_write_only
