"""Checkpoint interchange with the reference trainer.

The reference saves `{"step", "pipeline": pipeline.state_dict(), "optimizers": {group: Adam.state_dict()}, "schedulers",
"scalers"}` (engine/trainer.py:456-478); model tensors sit under the `_model.` prefix of the pipeline
(pipelines/base_pipeline.py:226-230; `module.` in front when saved from DDP) and are loaded back with that prefix
stripped (base_pipeline.py:100-126, trainer.py:426-453). Parameter names and shapes of this package's modules are the
torch path's, so the mapping is a prefix change; the optimiser state maps the arena's flat moments onto torch.optim.Adam's
per-parameter `state` in the order of `get_param_groups()[group]` (engine/optimizers.py:82-115)."""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional

import torch
from torch import Tensor

MODEL_PREFIX = "_model."


def model_state_dict(model: torch.nn.Module) -> Dict[str, Tensor]:
    """`pipeline.state_dict()` entries of the model: this model's state dict under the reference's `_model.` prefix."""
    return {MODEL_PREFIX + k: v.detach().clone() for k, v in model.state_dict().items()}


def load_model_state(model: torch.nn.Module, pipeline_state: Mapping[str, Tensor], strict: bool = True):
    """Load the model tensors out of a reference checkpoint's `pipeline` entry (datamanager / metric buffers and a DDP
    `module.` prefix are ignored). With parameters living in a ParamArena the copy lands in the arena views."""
    own = model.state_dict()
    found = {}
    for key, value in pipeline_state.items():
        k = key[len("module."):] if key.startswith("module.") else key
        if not k.startswith(MODEL_PREFIX):
            continue
        k = k[len(MODEL_PREFIX):]
        if k in own:
            found[k] = value
    missing = [k for k in own if k not in found]
    if strict and missing:
        raise KeyError(f"checkpoint has no tensor for {missing[:5]}{' ...' if len(missing) > 5 else ''}")
    with torch.no_grad():
        for k, v in found.items():
            if own[k].shape != v.shape:
                raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(own[k].shape)}")
            own[k].copy_(v.to(own[k].device, own[k].dtype))
    return missing


def optimizer_state_dicts(model, arena, lr: Optional[Mapping[str, float]] = None) -> Dict[str, Dict[str, Any]]:
    """torch.optim.Adam.state_dict() per optimiser group, built from the arena's moments (loadable by the reference's
    `Optimizers.load_optimizers`, engine/optimizers.py:195-203). Groups that have never stepped have empty state, as a
    fresh torch optimiser does."""
    offsets = {id(p): off for p, off in zip(arena.params, arena.offsets)}
    out = {}
    for name, params in model.get_param_groups().items():
        steps = arena.step_counts.get(name, 0)
        state = {}
        if steps > 0:
            for i, p in enumerate(params):
                off, n = offsets[id(p)], p.numel()
                state[i] = {"step": torch.tensor(float(steps)),
                            "exp_avg": arena.exp_avg[off:off + n].view(p.shape).detach().clone(),
                            "exp_avg_sq": arena.exp_avg_sq[off:off + n].view(p.shape).detach().clone()}
        group = {"lr": (lr or {}).get(name, arena.lr), "betas": tuple(arena.betas), "eps": arena.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(params)))}
        out[name] = {"state": state, "param_groups": [group]}
    return out


def load_optimizer_states(model, arena, optimizers: Mapping[str, Mapping[str, Any]]) -> None:
    """Inverse of optimizer_state_dicts: Adam moments and step counts of a reference checkpoint into the arena."""
    offsets = {id(p): off for p, off in zip(arena.params, arena.offsets)}
    for name, params in model.get_param_groups().items():
        if name not in optimizers:
            continue
        state = optimizers[name]["state"]
        steps = 0
        with torch.no_grad():
            for i, p in enumerate(params):
                st = state.get(i, state.get(str(i)))
                if st is None:
                    continue
                off, n = offsets[id(p)], p.numel()
                arena.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1).to(arena.exp_avg.device))
                arena.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(arena.exp_avg_sq.device))
                steps = max(steps, int(float(st["step"])))
        arena.step_counts[name] = steps


def grad_scaler_state(scalers: Optional[Mapping[str, Any]] = None) -> Dict[str, Any]:
    """State of the reference trainer's GradScaler (trainer.py:137, :475). nerfacto trains with mixed_precision=True, so
    that scaler is ENABLED and `load_state_dict` of an empty dict raises (trainer.py:439,450) — a checkpoint must carry
    a valid state. This package computes in fp32 without loss scaling, so it passes a loaded state through unchanged
    or emits a fresh scaler's defaults (scale 2^16, growth 2, backoff 0.5, interval 2000, tracker 0)."""
    if scalers:
        return dict(scalers)
    return {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}


def make_checkpoint(model, arena, step: int, lr: Optional[Mapping[str, float]] = None,
                    scalers: Optional[Mapping[str, Any]] = None) -> Dict[str, Any]:
    """A dict in the reference trainer's layout (trainer.py:467-478) for the parts this package owns. `scalers`: the
    GradScaler state of a checkpoint this run was resumed from (kept as is); default = a fresh enabled scaler's state."""
    return {"step": step, "pipeline": model_state_dict(model), "optimizers": optimizer_state_dicts(model, arena, lr),
            "schedulers": {}, "scalers": grad_scaler_state(scalers)}
