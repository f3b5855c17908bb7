"""Occupancy grid of the instant-ngp path: the role nerfacc.OccGridEstimator plays for the reference
(models/instant_ngp.py:117-123 construction, :151-156 `update_every_n_steps`, ray_samplers.py:481-493 `sampling`).

nerfacc 0.5.2 is not part of /root/reference; the published behaviour is restated (oracle/packed_oracle.py; parity of the
sample placement unpinned): a multi-level grid — level l covers the region of interest scaled by 2^l about its centre,
`resolution`^3 cells — holding an exponential moving maximum of `occ_eval_fn` (density x step size) per cell and its
thresholded binary. On the GPU everything is a HIP kernel (csrc/packed.hip): marching (one wavefront per ray, coarse
occupancy bits in LDS), the visibility scan with early termination, the compaction, and the grid refresh (cell positions,
decayed maximum, mean / threshold / binaries / coarse bitfield); only the choice of WHICH cells to refresh every 16 steps
(random draws, the list of occupied cells) is torch. CPU tensors (the CPU tests) take the same arithmetic through torch.
"""
from typing import Callable, List, Optional, Tuple

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
        # derived state, rebuilt from the buffers whenever they change (not part of the checkpoint): the coarse bitfield
        # the marcher stages in LDS, the mean of `occs` (the cap of alpha_thre in `sampling`), the host copy of the ROI
        self._coarse: Optional[Tensor] = None
        self._occ_mean: Optional[float] = None
        self._roi_host: Optional[List[float]] = None
        self._coarse_version = -1
        self._scratch = {}
        self.last_packed_info: Optional[Tensor] = None

    # ---- derived state ----------------------------------------------------------------------------------------------
    @property
    def _roi(self) -> List[float]:
        """Host copy of the region of interest (one device read, then cached; invalidated when a checkpoint is loaded —
        a stale copy would let the marcher and `_cell_positions`, which reads the buffer, disagree)."""
        if self._roi_host is None:
            self._roi_host = [float(v) for v in self.aabb.detach().cpu().tolist()]
        return self._roi_host

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._roi_host, self._coarse, self._occ_mean = None, None, None

    def _apply(self, fn, *args, **kwargs):
        """`.to()` / `.cuda()` / `.float()` replace the buffers by new tensors (whose version counters start over, possibly
        at the cached one): everything derived from them is rebuilt on the next use."""
        out = super()._apply(fn, *args, **kwargs)
        self._roi_host, self._coarse, self._occ_mean, self._coarse_version = None, None, None, -1
        self._scratch = {}
        return out

    def _buf(self, name: str, numel: int, dtype) -> Tensor:
        t = self._scratch.get(name)
        if t is None or t.numel() < numel or t.device != self.occs.device:
            t = torch.empty(numel, device=self.occs.device, dtype=dtype)
            self._scratch[name] = t
        return t

    def _refresh_derived(self, occ_thre: Optional[float] = None) -> None:
        """binaries (when `occ_thre` is given), the coarse bitfield and the cached mean from `occs`."""
        if self.occs.is_cuda:
            words = F.occgrid_coarse_words(self.levels, self.resolution)
            if self._coarse is None or self._coarse.device != self.occs.device or self._coarse.numel() != words:
                self._coarse = torch.zeros(words, device=self.occs.device, dtype=torch.int32)
            if occ_thre is None:  # binaries came from a checkpoint / were set by hand: derive the rest from them (rare)
                self._rebuild_coarse_from_binaries()
                self._occ_mean = float(self.occs.double().mean().float())
                return
            stats = self._buf("stats", 2, torch.float32)
            F.occgrid_binarise(self.occs, self.binaries, self._coarse if words else None, occ_thre,
                               self._buf("sum", 1024, torch.float64), stats)
            self._coarse_version = self.binaries._version
            self._occ_mean = float(stats[1])  # one host read per refresh (every 16 steps), none on the sampling path
        else:
            mean = self.occs.double().mean()
            if occ_thre is not None:
                thre = torch.clamp(mean, max=occ_thre).float()
                self.binaries.copy_((self.occs > thre).view_as(self.binaries).to(torch.uint8))
            self._occ_mean = float(mean.float())
            self._coarse = None

    def _rebuild_coarse_from_binaries(self) -> None:
        r, c = self.resolution, 4
        if self._coarse is None or self._coarse.numel() == 0 or r % c:
            return
        blocks = self.binaries.view(self.levels, r // c, c, r // c, c, r // c, c).amax(dim=(2, 4, 6)).reshape(-1).to(torch.int64)
        pad = (-blocks.numel()) % 32
        bits = torch.nn.functional.pad(blocks, (0, pad)).view(-1, 32)
        words = (bits << torch.arange(32, device=bits.device)).sum(dim=1)
        self._coarse.copy_(torch.where(words >= 2**31, words - 2**32, words).to(torch.int32))
        self._coarse_version = self.binaries._version

    def ensure_derived(self) -> None:
        """The derived state (coarse bitfield, cached mean) is current — called before every march."""
        stale = (self._occ_mean is None or (self._coarse is None and self.occs.is_cuda) or
                 (self._coarse is not None and self._coarse.device != self.occs.device) or
                 getattr(self, "_derived_from", None) != (self.binaries.data_ptr(), self.occs.data_ptr()))
        if stale:  # first use, or the buffers were re-homed / reassigned since the derived state was built
            self._refresh_derived()
            self._derived_from = (self.binaries.data_ptr(), self.occs.data_ptr())
        elif self.occs.is_cuda and self._coarse_version != self.binaries._version:
            self._rebuild_coarse_from_binaries()  # the binaries were written by hand since the bitfield was built

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
        self.ensure_derived()
        ray_indices, t_starts, t_ends, info = F.occgrid_march(
            rays_o, rays_d, self.binaries, self._roi, render_step_size, near_plane, far_plane, t_min, t_max, cone_angle,
            jitter if stratified else None, coarse=self._coarse)
        if alpha_thre > 0.0:  # never skip more eagerly than the grid itself believes the scene is occupied
            alpha_thre = min(alpha_thre, self._occ_mean)
        if sigma_fn is not None and t_starts.shape[0] > 0:
            sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            ray_indices, t_starts, t_ends, info, _ = F.packed_visibility_compact(ray_indices, t_starts, t_ends, sigmas, info,
                                                                                early_stop_eps, alpha_thre)
        self.last_packed_info = info
        return ray_indices, t_starts, t_ends

    # ---- grid maintenance -------------------------------------------------------------------------------------------
    def _cell_positions(self, flat: Optional[Tensor], num: int, jitter: Tensor) -> Tensor:
        """Positions inside the cells `flat` (None: all cells in order) at fractional offsets `jitter [num,3]`."""
        if self.occs.is_cuda:
            return F.occgrid_cell_positions(flat, num, self.binaries, self._roi, jitter)
        if flat is None:
            flat = torch.arange(num, device=jitter.device)
        level_idx, cell_idx = flat // self.cells_per_lvl, flat % self.cells_per_lvl
        r = self.resolution
        ix = torch.stack([cell_idx // (r * r), (cell_idx // r) % r, cell_idx % r], dim=-1).float()
        u = (ix + jitter) / r  # in [0,1]^3 of the level's box
        centre = (self.aabb[:3] + self.aabb[3:]) / 2
        half = (self.aabb[3:] - self.aabb[:3]) / 2 * (2.0 ** level_idx.float())[:, None]
        return (centre - half) + (u * 2) * half

    def refreshes_at(self, step: int, n: int = 16) -> bool:
        """`update_every_n_steps(step)` would refresh the grid (a caller that wants to time / account for the refreshes)."""
        return bool(self.training and step % n == 0)

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        """Refresh the grid every n-th training step: all cells during warm-up, afterwards a quarter of the cells at random
        plus as many occupied ones; occs = max(occs * decay, occ_eval_fn(x)) — over every estimate of a cell when it was
        drawn more than once —; binaries = occs > min(mean(occs), occ_thre)."""
        if not self.training or step % n != 0:
            return
        dev = self.occs.device
        total = self.levels * self.cells_per_lvl
        if step < warmup_steps:
            flat, num = None, total
        else:
            # per LEVEL: cells_per_lvl // 4 uniform draws + at most as many of that level's occupied cells (nerfacc's
            # `_sample_uniform_and_occupied_cells`; a pooled draw would let the level with the most occupied cells use up
            # the occupied quota and refresh the coarse levels less often — ADVICE r03)
            k = self.cells_per_lvl // 4
            parts = []
            flags = self.binaries.reshape(self.levels, -1)
            for lvl in range(self.levels):
                base = lvl * self.cells_per_lvl
                parts.append(torch.randint(self.cells_per_lvl, (k,), device=dev) + base)
                occupied = torch.nonzero(flags[lvl]).reshape(-1)
                if occupied.numel() > k:
                    occupied = occupied[torch.randint(occupied.numel(), (k,), device=dev)]
                parts.append(occupied + base)
            flat = torch.cat(parts)
            num = flat.numel()
        x = self._cell_positions(flat, num, torch.rand((num, 3), device=dev))
        occ = occ_eval_fn(x).reshape(-1).float()
        if self.occs.is_cuda:
            F.occgrid_update(self.occs, flat, occ, ema_decay, self._buf("old", total, torch.float32))
        elif flat is None:
            self.occs.copy_(torch.maximum(self.occs * ema_decay, occ))
        else:
            new = self.occs.clone()
            new[flat] = self.occs[flat] * ema_decay  # (repeats write the same value)
            self.occs.copy_(new.scatter_reduce(0, flat, occ, "amax", include_self=True))
        self._refresh_derived(occ_thre)
