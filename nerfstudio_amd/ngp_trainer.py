"""The instant-ngp training iteration (BASELINE configs[3]) as the reference's trainer runs it, over the explicit kernel
schedule of ngp_step.py:

    BEFORE_TRAIN_ITERATION   occupancy_grid.update_every_n_steps(step, occ_eval_fn = density x render_step_size)
                             (models/instant_ngp.py:149-163: called EVERY iteration; nerfacc refreshes the grid on every 16th —
                             all cells during the first 256 steps, afterwards a quarter of the cells of each level at random
                             plus as many occupied ones)
    get_train_loss_dict      march -> candidates' density -> visibility scan + compaction -> NerfactoField -> packed weights /
                             compositing -> MSE against the target over the random background
    backward, optimiser      packed scans -> field MLPs -> table scatter -> fused Adam over the "fields" group

`train_iteration(step)` is all of it; `grid_refresh_ms` keeps a host-clock record of the refreshes (they synchronise: the list
of occupied cells has a data-dependent length, as in nerfacc)."""
from __future__ import annotations

import time
from typing import Callable, List, Optional

import torch


class NgpTrainer:
    def __init__(self, model, arena, num_rays: int, device, module_path: bool = False,
                 after_refresh: Optional[Callable[[], None]] = None, refresh: bool = True, runner=None) -> None:
        self.model, self.arena, self.refresh = model, arena, refresh
        self.after_refresh = after_refresh  # bench.py: keeps its synthetic grid stationary
        self.table = model.field.mlp_base.encoding.hash_table
        self.runner = runner  # (tests: a CPU stand-in for ngp_step.NgpTrainStep)
        if not module_path and runner is None:
            from .ngp_step import NgpTrainStep

            self.runner = NgpTrainStep(model, num_rays, device)
        self.rb = self.batch = None
        self.samples: List = []
        self.refreshes: List[float] = []
        self.step_size = float(model.config.render_step_size)

    def set_batch(self, ray_bundle, batch) -> None:
        self.rb, self.batch = ray_bundle, batch
        if self.runner is not None:
            # (a collider's per-ray interval travels with the bundle: VolumetricSampler marches it, ray_samplers.py:470-476)
            self.runner.set_batch(ray_bundle.origins, ray_bundle.directions, ray_bundle.camera_indices, batch["image"],
                                  nears=getattr(ray_bundle, "nears", None), fars=getattr(ray_bundle, "fars", None))

    def update_occupancy_grid(self, step: int) -> None:
        """The model's BEFORE_TRAIN_ITERATION callback (models/instant_ngp.py:150-156)."""
        grid, fld = self.model.occupancy_grid, self.model.field
        if not self.refresh or not grid.refreshes_at(step):
            return
        t0 = time.perf_counter()
        grid.update_every_n_steps(step=step, occ_eval_fn=lambda x: fld.density_fn(x) * self.step_size)
        if self.after_refresh is not None:
            self.after_refresh()
        if len(self.refreshes) < 64:
            torch.cuda.synchronize()
            self.refreshes.append((time.perf_counter() - t0) * 1e3)

    def train_iteration(self, step: int):
        self.update_occupancy_grid(step)
        a, r = self.arena, self.runner
        if r is not None:
            a.zero_grad(skip=[self.table])  # the scatter writes the table's gradient
            r.forward()
            loss = r.loss()
            r.backward()
            a.step()
            self.samples.append(r.num_kept)
            return loss
        a.zero_grad()
        out = self.model(self.rb)
        loss = self.model.get_loss_dict(out, self.batch)["rgb_loss"]
        loss.backward()
        a.step()
        self.samples.append(out["num_samples_per_ray"])
        return loss

    def kept_per_step(self, last: int) -> float:
        s = self.samples[-last:]
        if self.runner is not None:
            return float(sum(s)) / max(len(s), 1)
        return float(torch.stack(s).float().sum(dim=1).mean())
