"""Restatement of the reference's training iteration for the tests that run where /root/reference does not exist (the GPU
box). TEST INFRASTRUCTURE ONLY — the product has no trainer (out of scope).

  train_iteration      engine/trainer.py:487-531        (zero_grad_some -> get_train_loss_dict -> sum of the loss terms ->
                                                         GradScaler.scale(loss).backward() -> optimizer_scaler_step_some ->
                                                         scaler.update() -> scheduler_step_all)
  get_train_loss_dict  pipelines/base_pipeline.py:290-303 (datamanager.next_train -> model(ray_bundle) -> get_metrics_dict ->
                                                         get_loss_dict)
  Optimizers           engine/optimizers.py:74-193 with AdamOptimizerConfig (:52-71) and ExponentialDecayScheduler
                       (engine/schedulers.py:109-142: LambdaLR)

tests/test_reference_trainer_drive.py pins this file to the reference's own code: the same toy model trained by both for
several iterations ends at the same parameter bits."""
import functools
from typing import Dict, List

import numpy as np
import torch


class Optimizers:
    def __init__(self, config: Dict[str, dict], param_groups: Dict[str, List[torch.nn.Parameter]]) -> None:
        self.config, self.optimizers, self.schedulers, self.parameters = config, {}, {}, {}
        for name, params in param_groups.items():
            if name not in config:
                raise RuntimeError(f"Optimizer config for '{name}' not found in config file.")
            oc = config[name]["optimizer"]
            self.optimizers[name] = torch.optim.Adam(params, lr=oc["lr"], eps=oc["eps"], weight_decay=oc.get("weight_decay", 0))
            self.parameters[name] = params
            sc = config[name].get("scheduler")
            if sc:
                lr_init, lr_final, max_steps = oc["lr"], sc["lr_final"], sc["max_steps"]

                def func(step, lr_init=lr_init, lr_final=lr_final, max_steps=max_steps):
                    t = np.clip(step / max_steps, 0, 1)  # no warm-up, no pre-warm-up, ramp "cosine" unused (nerfacto's recipe)
                    lr = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
                    return lr / lr_init

                self.schedulers[name] = torch.optim.lr_scheduler.LambdaLR(self.optimizers[name], lr_lambda=func)

    def zero_grad_some(self, names) -> None:
        for n in names:
            self.optimizers[n].zero_grad()

    def optimizer_scaler_step_some(self, grad_scaler, names) -> None:
        for n in names:
            opt = self.optimizers[n]
            if any(any(p.grad is not None for p in g["params"]) for g in opt.param_groups):
                grad_scaler.step(opt)

    def scheduler_step_all(self, step: int) -> None:
        for sched in self.schedulers.values():
            sched.step()


def get_train_loss_dict(pipeline, step: int):
    ray_bundle, batch = pipeline.datamanager.next_train(step)
    model_outputs = pipeline._model(ray_bundle)
    metrics_dict = pipeline.model.get_metrics_dict(model_outputs, batch)
    loss_dict = pipeline.model.get_loss_dict(model_outputs, batch, metrics_dict)
    return model_outputs, loss_dict, metrics_dict


def train_iteration(trainer, step: int):
    needs_zero = [g for g in trainer.optimizers.parameters.keys() if step % trainer.gradient_accumulation_steps[g] == 0]
    trainer.optimizers.zero_grad_some(needs_zero)
    device_type = trainer.device.split(":")[0]
    with torch.autocast(device_type=device_type, enabled=trainer.mixed_precision):
        # `self.pipeline.get_train_loss_dict(step=step)`: the pipeline's own method when it has one (a pipeline that overrides
        # it — nerfstudio_amd.pipeline.EngineSeam — must be reached), else the restated VanillaPipeline body
        own = getattr(trainer.pipeline, "get_train_loss_dict", None)
        _, loss_dict, metrics_dict = own(step=step) if own is not None else get_train_loss_dict(trainer.pipeline, step)
        loss = functools.reduce(torch.add, loss_dict.values())
    trainer.grad_scaler.scale(loss).backward()
    needs_step = [g for g in trainer.optimizers.parameters.keys()
                  if step % trainer.gradient_accumulation_steps[g] == trainer.gradient_accumulation_steps[g] - 1]
    trainer.optimizers.optimizer_scaler_step_some(trainer.grad_scaler, needs_step)
    scale = trainer.grad_scaler.get_scale()
    trainer.grad_scaler.update()
    if scale <= trainer.grad_scaler.get_scale():
        trainer.optimizers.scheduler_step_all(step)
    return loss, loss_dict, metrics_dict
