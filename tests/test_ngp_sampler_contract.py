"""The instant-ngp sampler against the fixture the REFERENCE's own VolumetricSampler wrote (tests/golden/ngp_sampler.npz,
make_golden_ngp.py: the reference module around a stub estimator that answers nerfacc's `sampling` contract with the
oracle's samples).

CPU tier: this package's VolumetricSampler around the same oracle-backed estimator must assemble bit-identical packed
RaySamples in eval mode (no density check: nothing needs the GPU) and produce the reference's fake sample for an empty
result. GPU tier: the real thing — HIP marcher + HIP density + packed visibility scan + compaction inside
OccGridEstimator.sampling, driven by VolumetricSampler in training mode — against the reference module's training-mode
output (membership may differ only where a sample's alpha or transmittance sits on a threshold: the field's density comes
from different fp32 arithmetic)."""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc
from oracle import packed_oracle as po

ROI = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]


class _OracleEstimator(torch.nn.Module):
    def __init__(self, binaries, occs_mean, jitter):
        super().__init__()
        self.binaries, self.occs_mean, self.jitter = binaries, occs_mean, jitter
        self.last_packed_info = None

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, near_plane=0.0, far_plane=1e10, t_min=None, t_max=None,
                 render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0, stratified=False, cone_angle=0.0, jitter=None):
        idx, ts, te = po.occgrid_march(rays_o.numpy(), rays_d.numpy(), self.binaries, ROI, render_step_size, near_plane=near_plane,
                                       far_plane=far_plane, t_min=None if t_min is None else t_min.numpy(),
                                       t_max=None if t_max is None else t_max.numpy(), cone_angle=cone_angle,
                                       jitter=self.jitter if stratified else None)
        assert sigma_fn is None
        return torch.from_numpy(idx), torch.from_numpy(ts), torch.from_numpy(te)


def _bundle(g, bounds=True, device="cpu"):
    from nerfstudio_amd.cameras.rays import RayBundle

    t = lambda k: torch.from_numpy(g[k]).to(device)  # noqa: E731
    return RayBundle(origins=t("origins"), directions=t("directions"), pixel_area=t("pixel_area"), camera_indices=t("cams"),
                     nears=t("nears") if bounds else None, fars=t("fars") if bounds else None)


def test_volumetric_sampler_assembles_the_reference_samples_on_cpu(golden):
    from nerfstudio_amd.model_components.ray_samplers import VolumetricSampler

    g = golden("ngp_sampler")
    est = _OracleEstimator(g["binaries"], float(g["occs_mean"]), g["jitter"])
    sampler = VolumetricSampler(occupancy_grid=est, density_fn=None).eval()
    samples, ray_indices = sampler(ray_bundle=_bundle(g), render_step_size=float(g["step"]), near_plane=0.05, far_plane=1e3,
                                   alpha_thre=float(g["alpha_thre"]), cone_angle=float(g["cone"]))
    f = samples.frustums
    np.testing.assert_array_equal(ray_indices.numpy(), g["eval_ray_indices"])
    for name, got in (("starts", f.starts), ("ends", f.ends), ("origins", f.origins), ("directions", f.directions),
                      ("pixel_area", f.pixel_area), ("camera_indices", samples.camera_indices)):
        np.testing.assert_array_equal(got.numpy(), g[f"eval_{name}"], err_msg=name)
    # empty result -> the reference's single fake sample (ray 0, [1, 1]); far_plane None -> 1e10
    empty = _OracleEstimator(np.zeros_like(g["binaries"]), float(g["occs_mean"]), g["jitter"])
    samples, ray_indices = VolumetricSampler(occupancy_grid=empty).eval()(ray_bundle=_bundle(g, bounds=False),
                                                                            render_step_size=float(g["step"]))
    np.testing.assert_array_equal(ray_indices.numpy(), g["empty_ray_indices"])
    np.testing.assert_array_equal(samples.frustums.starts.numpy(), g["empty_starts"])
    np.testing.assert_array_equal(samples.frustums.ends.numpy(), g["empty_ends"])
    with pytest.raises(RuntimeError, match="call forward"):
        sampler.generate_ray_samples()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "train_nobounds", "eval"])
def test_volumetric_sampler_on_the_kernels_vs_the_reference_module(golden, mode):
    from nerfstudio_amd import _native
    from nerfstudio_amd.field_components.spatial_distortions import SceneContraction
    from nerfstudio_amd.fields.nerfacto_field import NerfactoField
    from nerfstudio_amd.model_components.occupancy import OccGridEstimator
    from nerfstudio_amd.model_components.ray_samplers import VolumetricSampler

    _native.load()
    g = golden("ngp_sampler")
    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 12), prop_grids=(), num_images=4, average_init_density=1.0)
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    with torch.no_grad():
        params["field.mlp_base.model.1.layers.1.bias"][0] = float(np.log(float(g["density_gain"])))
    field = NerfactoField(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=4, log2_hashmap_size=12,
                          spatial_distortion=SceneContraction(order=float("inf")))
    sd = {k[len("field."):]: v.detach().clone() for k, v in params.items() if k.startswith("field.")}
    missing, unexpected = field.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    field = field.cuda()
    B = g["binaries"]
    grid = OccGridEstimator(torch.tensor(ROI), resolution=B.shape[1], levels=B.shape[0]).cuda()
    grid.binaries.copy_(torch.from_numpy(B.astype(np.uint8)))
    grid.occs.fill_(float(g["occs_mean"]))  # the cap of alpha_thre: min(alpha_thre, mean(occs))
    sampler = VolumetricSampler(occupancy_grid=grid, density_fn=field.density_fn).cuda()
    sampler.train(mode != "eval")
    rb = _bundle(g, bounds=mode != "train_nobounds", device="cuda")
    samples, ray_indices = sampler(ray_bundle=rb, render_step_size=float(g["step"]), near_plane=0.05,
                                   far_plane=None if mode == "train_nobounds" else 1e3, alpha_thre=float(g["alpha_thre"]),
                                   cone_angle=float(g["cone"]), jitter=torch.from_numpy(g["jitter"]).cuda())
    got = {(int(r), float(s)) for r, s in zip(ray_indices.cpu().numpy(), samples.frustums.starts[:, 0].cpu().numpy())}
    ref = {(int(r), float(s)) for r, s in zip(g[f"{mode}_ray_indices"], g[f"{mode}_starts"][:, 0])}
    if mode == "eval":  # no density check: the marcher alone, bit-exact
        np.testing.assert_array_equal(ray_indices.cpu().numpy(), g["eval_ray_indices"])
        np.testing.assert_array_equal(samples.frustums.starts.cpu().numpy(), g["eval_starts"])
        np.testing.assert_array_equal(samples.frustums.ends.cpu().numpy(), g["eval_ends"])
    else:
        assert len(ref) > 400 and len(got ^ ref) <= max(2, len(ref) // 200), (len(got), len(ref), len(got ^ ref))
    # every gathered field of the packed samples, on the common samples
    idx = ray_indices.cpu().numpy()
    np.testing.assert_array_equal(samples.frustums.origins.cpu().numpy(), g["origins"][idx])
    np.testing.assert_array_equal(samples.frustums.directions.cpu().numpy(), g["directions"][idx])
    np.testing.assert_array_equal(samples.frustums.pixel_area.cpu().numpy(), g["pixel_area"][idx])
    np.testing.assert_array_equal(samples.camera_indices.cpu().numpy(), g["cams"][idx])
    info = getattr(ray_indices, "_nsamd_packed_info", None)
    assert info is not None and np.array_equal(info[:, 1].cpu().numpy(), np.bincount(idx, minlength=len(g["origins"])))
