"""Ray samplers (reference: nerfstudio/model_components/ray_samplers.py — Sampler :30-50,
UniformLinDispPiecewiseSampler :225-248, PDFSampler :251-372, ProposalNetworkSampler :522-617)."""
from abc import abstractmethod
from typing import Any, Callable, List, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import functional as F
from ..cameras.rays import RayBundle, RaySamples, pack_of, samples_from_bins


class Sampler(nn.Module):
    """Generate samples along a ray."""

    def __init__(self, num_samples: Optional[int] = None) -> None:
        super().__init__()
        self.num_samples = num_samples

    @abstractmethod
    def generate_ray_samples(self) -> Any:
        """Generate Ray Samples"""

    def forward(self, *args, **kwargs) -> Any:
        return self.generate_ray_samples(*args, **kwargs)


def _spacing_closure(nears: Tensor, fars: Tensor, spacing: int = 0) -> Callable:
    """The reference's `spacing_to_euclidean_fn` closure (ray_samplers.py:112-116) for downstream torch callers; the
    HIP samplers evaluate the same map in-kernel."""
    if spacing == 1:
        return lambda x: x * fars + (1 - x) * nears
    fn = lambda x: torch.where(x < 1, x / 2, 1 - 1 / (2 * x))  # noqa: E731
    inv = lambda x: torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))  # noqa: E731
    s_near, s_far = fn(nears), fn(fars)
    return lambda x: inv(x * s_far + (1 - x) * s_near)


class UniformLinDispPiecewiseSampler(Sampler):
    """First half of the samples uniform, second half linear in disparity (ray_samplers.py:225-248)."""

    spacing = 0

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples)
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None,
                             jitter: Optional[Tensor] = None) -> RaySamples:
        """`jitter` (optional, in [0,1): `[num_rays,1]` with single_jitter, else `[num_rays, num_samples+1]`) injects the
        random draw — used by the parity tests; by default it is drawn here with torch.rand as the reference does
        (ray_samplers.py:103-107)."""
        assert ray_bundle is not None
        assert ray_bundle.nears is not None
        assert ray_bundle.fars is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        num_rays = ray_bundle.origins.shape[0]
        if self.train_stratified and self.training:
            if jitter is None:
                jitter = torch.rand((num_rays, 1 if self.single_jitter else num_samples + 1), dtype=torch.float32,
                                    device=ray_bundle.origins.device)
        else:
            jitter = None
        s_bins, t_bins = F.piecewise_bins(ray_bundle.nears, ray_bundle.fars, num_samples, jitter, self.spacing)
        return samples_from_bins(ray_bundle, s_bins, t_bins,
                                 _spacing_closure(ray_bundle.nears, ray_bundle.fars, self.spacing), self.spacing)


class UniformSampler(UniformLinDispPiecewiseSampler):
    """Sample uniformly along a ray (ray_samplers.py:131-155) — the `proposal-initial-sampler uniform` of the Blender
    benchmark recipe (scripts/benchmarking/launch_train_blender.sh:28-33). Same kernel, identity spacing function."""

    spacing = 1


class PDFSampler(Sampler):
    """Inverse-CDF resampling of a weight histogram (ray_samplers.py:251-372)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = True, histogram_padding: float = 0.01) -> None:
        super().__init__(num_samples=num_samples)
        self.train_stratified = train_stratified
        self.include_original = include_original
        self.histogram_padding = histogram_padding
        self.single_jitter = single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, ray_samples: Optional[RaySamples] = None,
                             weights: Optional[Tensor] = None, num_samples: Optional[int] = None, eps: float = 1e-5,
                             jitter: Optional[Tensor] = None, anneal: float = 1.0,
                             anneal_dev: Optional[Tensor] = None) -> RaySamples:
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples and ray_bundle must be provided")
        assert weights is not None, "weights must be provided"
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        assert ray_samples.spacing_starts is not None and ray_samples.spacing_ends is not None, (
            "ray_sample spacing_starts and spacing_ends must be provided")
        if self.train_stratified and self.training:
            if jitter is None:
                jitter = torch.rand((weights.shape[0], 1 if self.single_jitter else num_samples + 1), device=weights.device)
        else:
            jitter = None
        pk = pack_of(ray_samples)
        spacing = pk.spacing if pk is not None else 0
        if pk is not None and pk.s_bins is not None:
            existing = pk.s_bins
        else:
            existing = torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)
        s_bins, t_bins = F.pdf_resample(existing, weights[..., 0], num_samples, jitter, ray_bundle.nears, ray_bundle.fars,
                                        anneal=anneal, histogram_padding=self.histogram_padding, eps=eps,
                                        anneal_dev=anneal_dev, spacing=spacing, include_original=self.include_original)
        if pk is None:
            # samples of a foreign sampler: its own s -> t closure is the only statement of the spacing function
            t_bins = ray_samples.spacing_to_euclidean_fn(s_bins)
        return samples_from_bins(ray_bundle, s_bins, t_bins, ray_samples.spacing_to_euclidean_fn, spacing)


class ProposalNetworkSampler(Sampler):
    """Proposal-network sampler (ray_samplers.py:522-617): piecewise initial samples, then for each proposal level
    density -> weights -> PDF resample, 256 -> 96 -> 48 for nerfacto."""

    def __init__(
        self,
        num_proposal_samples_per_ray: Tuple[int, ...] = (64,),
        num_nerf_samples_per_ray: int = 32,
        num_proposal_network_iterations: int = 2,
        single_jitter: bool = False,
        update_sched: Callable = lambda x: 1,
        initial_sampler: Optional[Sampler] = None,
        pdf_sampler: Optional[PDFSampler] = None,
    ) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if self.num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.initial_sampler = initial_sampler or UniformLinDispPiecewiseSampler(single_jitter=single_jitter)
        self.pdf_sampler = pdf_sampler or PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0
        # graph-replay hooks (set by a trainer that captures the step in a hipGraph): a device copy of the anneal
        # exponent, and an override of the host-side "update the proposal nets this step?" decision
        self.anneal_dev: Optional[Tensor] = None
        self.force_updated: Optional[bool] = None

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step) -> None:
        self._step = step
        self._steps_since_update += 1

    def updated_this_step(self) -> bool:
        """ray_samplers.py:590 — proposal networks get gradient on this step?"""
        return bool(self._steps_since_update > self.update_sched(self._step) or self._step < 10)

    def mark_updated(self) -> None:
        self._steps_since_update = 0

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, density_fns: Optional[List[Callable]] = None,
                             jitters: Optional[List[Tensor]] = None) -> Tuple[RaySamples, List, List]:
        """`jitters` (optional): one `[num_rays,1]` draw per level, injected by the parity tests."""
        assert ray_bundle is not None
        assert density_fns is not None
        weights_list, ray_samples_list = [], []
        n = self.num_proposal_network_iterations
        weights = None
        ray_samples = None
        updated = self.updated_this_step() if self.force_updated is None else self.force_updated
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            jit = jitters[i_level] if jitters is not None else None
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples, jitter=jit)
            else:
                assert weights is not None
                # the anneal pow(weights, anneal) (ray_samplers.py:601) happens inside the resampling kernel
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=num_samples, jitter=jit,
                                               anneal=self._anneal, anneal_dev=self.anneal_dev)
            if is_prop:
                with torch.set_grad_enabled(updated and torch.is_grad_enabled()):
                    density = self._density(density_fns[i_level], ray_samples)
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated and self.force_updated is None:
            self._steps_since_update = 0
        assert ray_samples is not None
        return ray_samples, weights_list, ray_samples_list

    @staticmethod
    def _density(density_fn: Callable, ray_samples: RaySamples) -> Tensor:
        """`density_fns[i]` is `field.density_fn` (models/nerfacto.py:188,200). When the field is one of ours the
        samples go to it as rays + bins (no `[N,S,3]` positions in HBM); any other callable gets positions."""
        owner = getattr(density_fn, "__self__", None)
        if owner is not None and hasattr(owner, "get_density") and getattr(density_fn, "__name__", "") == "density_fn" \
                and owner.__class__.__module__.startswith("nerfstudio_amd."):
            return owner.get_density(ray_samples)[0]
        return density_fn(ray_samples.frustums.get_positions())


class VolumetricSampler(Sampler):
    """Sampler of the instant-ngp path (ray_samplers.py:385-519): samples along a ray by marching through the occupancy
    grid; in training the candidates are thinned by the field's own density (transmittance-ordered early termination,
    alpha threshold). Fuses generation and density check: call forward() directly. Returns PACKED samples."""

    def __init__(self, occupancy_grid, density_fn: Optional[Callable] = None) -> None:
        super().__init__()
        assert occupancy_grid is not None
        self.density_fn = density_fn
        self.occupancy_grid = occupancy_grid

    def get_sigma_fn(self, origins: Tensor, directions: Tensor, times=None) -> Optional[Callable]:
        if self.density_fn is None or not self.training:
            return None
        density_fn = self.density_fn

        def sigma_fn(t_starts, t_ends, ray_indices):
            positions = F.packed_positions(origins, directions, ray_indices, t_starts, t_ends)  # ray_samplers.py:424-426
            return density_fn(positions).squeeze(-1)

        return sigma_fn

    def generate_ray_samples(self) -> RaySamples:
        raise RuntimeError(
            "The VolumetricSampler fuses sample generation and density check together. Please call forward() directly.")

    def forward(self, ray_bundle, render_step_size: float, near_plane: float = 0.0, far_plane: Optional[float] = None,
                alpha_thre: float = 0.01, cone_angle: float = 0.0, jitter: Optional[Tensor] = None):
        """-> (ray_samples, ray_indices): packed samples `[n]` and the ray each belongs to (ray_samplers.py:437-519)."""
        from ..cameras.rays import Frustums

        rays_o, rays_d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            t_min, t_max = ray_bundle.nears.contiguous().reshape(-1), ray_bundle.fars.contiguous().reshape(-1)
        else:
            t_min = t_max = None
        if far_plane is None:
            far_plane = 1e10
        ray_indices, starts, ends = self.occupancy_grid.sampling(
            rays_o=rays_o, rays_d=rays_d, t_min=t_min, t_max=t_max, sigma_fn=self.get_sigma_fn(rays_o, rays_d),
            render_step_size=render_step_size, near_plane=near_plane, far_plane=far_plane, stratified=self.training,
            cone_angle=cone_angle, alpha_thre=alpha_thre, jitter=jitter)
        if starts.shape[0] == 0:
            # a single fake sample (ray 0, [1, 1]) keeps every downstream shape valid (ray_samplers.py:494-500)
            ray_indices = torch.zeros((1,), dtype=torch.long, device=rays_o.device)
            starts = torch.ones((1,), dtype=torch.float32, device=rays_o.device)
            ends = torch.ones((1,), dtype=torch.float32, device=rays_o.device)
        else:
            # the (start, count) rows the marcher / compaction already hold ride along on the index tensor: the packed
            # renderers and NGPModel.get_outputs take them instead of recomputing them
            info = getattr(self.occupancy_grid, "last_packed_info", None)
            if info is not None and info.shape[0] == rays_o.shape[0]:
                ray_indices._nsamd_packed_info = info
        cams = ray_bundle.camera_indices
        ray_samples = RaySamples(
            frustums=Frustums(origins=rays_o[ray_indices], directions=rays_d[ray_indices], starts=starts[..., None],
                              ends=ends[..., None], pixel_area=ray_bundle.pixel_area[ray_indices]),
            camera_indices=None if cams is None else cams.contiguous()[ray_indices])
        return ray_samples, ray_indices
