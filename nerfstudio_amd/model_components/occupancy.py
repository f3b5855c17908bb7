"""Occupancy grid of the instant-ngp path: the role nerfacc.OccGridEstimator plays for the reference
(models/instant_ngp.py:117-123 construction, :151-156 `update_every_n_steps`, ray_samplers.py:481-493 `sampling`).

nerfacc 0.5.2 is not part of /root/reference; the published behaviour is restated (oracle/packed_oracle.py; parity of the
sample placement unpinned): a multi-level grid — level l covers the region of interest scaled by 2^l about its centre,
`resolution`^3 cells — holding an exponential moving maximum of `occ_eval_fn` (density x step size) per cell and its
thresholded binary. Marching, the visibility scan with early termination and the compaction are HIP kernels
(csrc/packed.hip); the grid bookkeeping (which cells to refresh, EMA, threshold) is a handful of elementwise torch ops
every 16 training steps, off the per-step path.
"""
from typing import Callable, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import functional as F


class OccGridEstimator(nn.Module):
    def __init__(self, roi_aabb: Tensor, resolution: int = 128, levels: int = 1) -> None:
        super().__init__()
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(6)
        self.resolution, self.levels = int(resolution), int(levels)
        self.cells_per_lvl = self.resolution**3
        self.register_buffer("aabb", roi_aabb)
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros((self.levels, self.resolution, self.resolution, self.resolution),
                                                     dtype=torch.uint8))
        self._roi = [float(v) for v in roi_aabb.tolist()]  # host copy: no device sync on the sampling path

    # ---- sampling ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None, near_plane: float = 0.0,
                 far_plane: float = 1e10, t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None,
                 render_step_size: float = 1e-3, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                 stratified: bool = False, cone_angle: float = 0.0, jitter: Optional[Tensor] = None
                 ) -> Tuple[Tensor, Tensor, Tensor]:
        """-> (ray_indices int64 `[n]`, t_starts, t_ends `[n]`). With `sigma_fn` (training) the candidates from the
        occupancy march go through the packed transmittance scan: samples behind transmittance `early_stop_eps` or with
        alpha below `alpha_thre` are dropped (render_visibility_from_density) and the survivors are compacted."""
        if stratified and jitter is None:
            jitter = torch.rand(rays_o.shape[0], device=rays_o.device)
        ray_indices, t_starts, t_ends, info = F.occgrid_march(
            rays_o, rays_d, self.binaries, self._roi, render_step_size, near_plane, far_plane, t_min, t_max, cone_angle,
            jitter if stratified else None)
        if alpha_thre > 0.0:  # never skip more eagerly than the grid itself believes the scene is occupied
            alpha_thre = min(alpha_thre, float(self.occs.mean()))
        if sigma_fn is not None and t_starts.shape[0] > 0:
            sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            ray_indices, t_starts, t_ends, info, _ = F.packed_visibility_compact(ray_indices, t_starts, t_ends, sigmas, info,
                                                                                early_stop_eps, alpha_thre)
        self.last_packed_info = info
        return ray_indices, t_starts, t_ends

    # ---- grid maintenance -------------------------------------------------------------------------------------------
    def _cell_positions(self, level_idx: Tensor, cell_idx: Tensor, jitter: Tensor) -> Tensor:
        r = self.resolution
        ix = torch.stack([cell_idx // (r * r), (cell_idx // r) % r, cell_idx % r], dim=-1).float()
        u = (ix + jitter) / r  # in [0,1]^3 of the level's box
        centre = (self.aabb[:3] + self.aabb[3:]) / 2
        half = (self.aabb[3:] - self.aabb[:3]) / 2 * (2.0 ** level_idx.float())[:, None]
        return centre - half + u * 2 * half

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        """Refresh the grid every n-th training step: all cells during warm-up, afterwards a quarter of the cells at random
        plus as many occupied ones; occs = max(occs * decay, occ_eval_fn(x)); binaries = occs > min(mean(occs), occ_thre)."""
        if not self.training or step % n != 0:
            return
        dev = self.occs.device
        total = self.levels * self.cells_per_lvl
        if step < warmup_steps:
            flat = torch.arange(total, device=dev)
        else:
            k = total // 4
            uniform = torch.randint(total, (k,), device=dev)
            occupied = torch.nonzero(self.binaries.reshape(-1)).reshape(-1)
            if occupied.numel() > k:
                occupied = occupied[torch.randint(occupied.numel(), (k,), device=dev)]
            flat = torch.cat([uniform, occupied])
        level_idx, cell_idx = flat // self.cells_per_lvl, flat % self.cells_per_lvl
        x = self._cell_positions(level_idx, cell_idx, torch.rand((flat.numel(), 3), device=dev))
        occ = occ_eval_fn(x).reshape(-1).float()
        self.occs[flat] = torch.maximum(self.occs[flat] * ema_decay, occ)
        thre = torch.clamp(self.occs.mean(), max=occ_thre)
        self.binaries.copy_((self.occs > thre).view_as(self.binaries).to(torch.uint8))
