"""CPU stand-in for train_step.NerfactoTrainStep — TEST INFRASTRUCTURE, as tests/cpu_kernels.py: it lets the HOST logic of
trainer.HipTrainer / pipeline.TrainEngine / dp_schedule.PipelinedExchange (segment order, arena bookkeeping, optimiser-state
sharing, learning-rate hand-over, the gradient exchange over gloo) run where no GPU is present. The numbers come from the
module path of the same model with the kernel wrappers replaced by the oracle's restatements (cpu_kernels.installed): one
autograd forward / backward per iteration, the gradients handed out in the runner's phases. Nothing in the product imports
this; on a GPU box the real runner launches the kernels."""
import torch


def cpu_adam(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-15, grad_scale=1.0, hyper_dev=None):
    """torch.optim.Adam's update on arena slices with the device-resident step scalars of functional.adam_hyper
    (hyper_dev = (lr / bias_correction1, sqrt(bias_correction2))) — the stand-in for csrc/misc.hip's kernel."""
    import math

    b1, b2 = betas
    g = grads * grad_scale
    exp_avg.lerp_(g, 1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    if hyper_dev is not None:
        step_size, bc2_sqrt = float(hyper_dev[0]), float(hyper_dev[1])
    else:
        step_size, bc2_sqrt = lr / (1 - b1 ** step), math.sqrt(1 - b2 ** step)
    denom = (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
    params.addcdiv_(exp_avg, denom, value=-step_size)


class CpuRunner:
    """The runner interface HipTrainer drives (train_step.NerfactoTrainStep): static batch buffers, forward phases, backward
    phases that ACCUMULATE into the arena's gradient views (`grad_lookup`), loss / output views."""

    side_stream = None
    cam_opt = None
    cameras_outside = False

    def __init__(self, model, num_rays, device, seed_base=0):
        self.model, self.n = model, int(num_rays)
        self.origins = torch.zeros(self.n, 3)
        self.directions = torch.zeros(self.n, 3)
        self.camera_indices = torch.zeros(self.n, dtype=torch.long)
        self.target = torch.zeros(self.n, 3)
        self.jitter = torch.zeros(3, self.n)
        self.dist_per_ray = torch.zeros(self.n)
        self.anneal_dev = torch.ones(1)
        self.grad_lookup = None
        self.seed_base, self.iterations = seed_base, 0
        self.calls = []
        self._losses = self._out = self._grads = None

    def set_batch(self, origins, directions, camera_indices, target=None):
        self.origins.copy_(origins.reshape(-1, 3))
        self.directions.copy_(directions.reshape(-1, 3))
        self.camera_indices.copy_(camera_indices.reshape(-1))
        if target is not None:
            self.target.copy_(target.reshape(-1, 3))

    def written_params(self):
        return []

    def apply_camera_corrections(self):
        pass

    def forward_proposals(self, draw_jitter=True, need_enc=True):
        self.calls.append("pfwd")

    def forward_main_and_losses(self, updated):
        """The whole forward + the gradient of the summed losses, through the module path (autograd, CPU stand-in kernels)."""
        from nerfstudio_amd.cameras.rays import RayBundle

        self.calls.append(("main", bool(updated)))
        m = self.model
        ps = m.proposal_sampler
        ps.force_updated = bool(updated)
        torch.manual_seed(self.seed_base + self.iterations)  # the sampler's torch.rand draws of this iteration
        self.iterations += 1
        rb_cls = type(self._bundle_like) if getattr(self, "_bundle_like", None) is not None else RayBundle  # the model's own
        rb = rb_cls(origins=self.origins.clone(), directions=self.directions.clone(), pixel_area=torch.full((self.n, 1), 1e-6),
                    camera_indices=self.camera_indices.clone()[:, None])
        try:
            out = m(rb)
            batch = {"image": self.target.clone()}
            metrics = m.get_metrics_dict(out, batch)
            losses = m.get_loss_dict(out, batch, metrics)
        finally:
            ps.force_updated = None
        keys = ("rgb_loss", "interlevel_loss", "distortion_loss")
        params = [p for p in self.grad_lookup_params() if p.requires_grad]
        grads = torch.autograd.grad(sum(losses[k] for k in keys), params, allow_unused=True)
        self._grads = {id(p): g for p, g in zip(params, grads) if g is not None}
        self._losses = {k: losses[k].detach() for k in keys}
        self._out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        self.dist_per_ray.fill_(float(metrics["distortion"]))

    def grad_lookup_params(self):
        return list(self.model.field.parameters()) + list(self.model.proposal_networks.parameters())

    def _emit(self, module):
        for p in module.parameters():
            g = self._grads.get(id(p))
            if g is not None:
                self.grad_lookup[id(p)].add_(g)

    def backward_main(self, field=True, reserve=False):  # (reserve: the GPU runner leaves compute units to the proposal chains)
        self.calls.append("bmain")
        self._emit(self.model.field)

    def backward_proposals(self, levels=None):
        self.calls.append("bprop")
        self._emit(self.model.proposal_networks)

    def backward_all(self, updated):
        self.backward_main()
        if updated:
            self.backward_proposals()

    def backward_cameras(self, updated, force=False):
        pass

    def loss_dict(self):
        return dict(self._losses)

    def outputs(self):
        return {"rgb": self._out["rgb"], "accumulation": self._out["accumulation"], "depth": self._out["depth"]}


class CpuNgpRunner:
    """The runner interface ngp_trainer.NgpTrainer drives (ngp_step.NgpTrainStep): set_batch with ANY number of rays, forward,
    loss, backward into the parameters' `.grad` (the engine binds views of the arena's gradient; the table's gradient is
    WRITTEN, the others accumulate), outputs, `num_kept`, `target`. The numbers come from the module path of the same model
    (plugin.HipNGPModel under the reference's Model.forward / get_loss_dict) over tests/cpu_kernels.py."""

    def __init__(self, model, num_rays, device, bundle_cls, seed_base=0):
        self.model, self.n, self.bundle_cls = model, int(num_rays), bundle_cls
        self.seed_base, self.iterations = seed_base, 0
        self.sizes, self.num_kept = [], 0
        self.target = None
        self._out = self._loss = None

    def set_batch(self, origins, directions, camera_indices, target=None, nears=None, fars=None):
        self.origins, self.directions = origins.reshape(-1, 3).clone(), directions.reshape(-1, 3).clone()
        self.cams = camera_indices.reshape(-1).clone()
        self.n = self.origins.shape[0]
        self.sizes.append(self.n)
        if target is not None:
            self.target = target.reshape(-1, 3).clone()

    def forward(self, jitter=None):
        torch.manual_seed(self.seed_base + self.iterations)  # the sampler's and the loss's torch.rand draws of this iteration
        self.iterations += 1
        rb = self.bundle_cls(origins=self.origins, directions=self.directions, pixel_area=torch.full((self.n, 1), 1e-6),
                             camera_indices=self.cams[:, None])
        self._out = self.model(rb)
        self.num_kept = int(self._out["num_samples_per_ray"].sum())

    def loss(self, background=None):
        self._loss = self.model.get_loss_dict(self._out, {"image": self.target})["rgb_loss"]
        return self._loss.detach()

    def backward(self):
        table = self.model.field.mlp_base.encoding.hash_table
        table.grad.zero_()  # (the schedule's scatter writes the table's gradient, ngp_step.py)
        self._loss.backward()

    def outputs(self):
        o = self._out
        return {"rgb": o["rgb"].detach(), "accumulation": o["accumulation"].detach(), "depth": o["depth"].detach(),
                "num_samples_per_ray": o["num_samples_per_ray"]}
