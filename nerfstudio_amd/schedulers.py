"""Learning-rate schedule of the nerfacto recipe (reference: nerfstudio/engine/schedulers.py:86-142,
configs/method_configs.py:110-121: every optimiser group uses ExponentialDecayScheduler(lr_final=1e-4, max_steps=200000)).

Host-only arithmetic: the value for iteration `step` goes into the device-resident Adam step size
(functional.adam_hyper), so a captured hipGraph of the training step follows the schedule without re-capture. The
reference steps every scheduler once per training iteration (engine/optimizers.py:183-193, trainer.py:527), whether or
not the group's optimiser stepped, so iteration i (0-based) runs with lr_init * multiplier(i)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Literal, Optional


@dataclass
class ExponentialDecaySchedulerConfig:
    """Same fields and defaults as the reference's config (schedulers.py:86-106)."""

    lr_pre_warmup: float = 1e-8
    lr_final: Optional[float] = None
    warmup_steps: int = 0
    max_steps: int = 100000
    ramp: Literal["linear", "cosine"] = "cosine"


class ExponentialDecayScheduler:
    """Linear/cosine warm-up to lr_init, then exponential decay to lr_final over max_steps (schedulers.py:109-142)."""

    def __init__(self, config: ExponentialDecaySchedulerConfig) -> None:
        self.config = config

    def get_lr(self, step: int, lr_init: float) -> float:
        cfg = self.config
        lr_final = lr_init if cfg.lr_final is None else cfg.lr_final
        if step < cfg.warmup_steps:
            if cfg.ramp == "cosine":
                frac = min(max(step / cfg.warmup_steps, 0.0), 1.0)
                return cfg.lr_pre_warmup + (lr_init - cfg.lr_pre_warmup) * math.sin(0.5 * math.pi * frac)
            return cfg.lr_pre_warmup + (lr_init - cfg.lr_pre_warmup) * step / cfg.warmup_steps
        t = min(max((step - cfg.warmup_steps) / (cfg.max_steps - cfg.warmup_steps), 0.0), 1.0)
        return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def nerfacto_schedulers():
    """The schedulers of method_configs["nerfacto"] for the two groups this package trains."""
    cfg = ExponentialDecaySchedulerConfig(lr_final=0.0001, max_steps=200000)
    return {"fields": ExponentialDecayScheduler(cfg), "proposal_networks": ExponentialDecayScheduler(cfg)}
