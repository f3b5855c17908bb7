#!/usr/bin/env python3
"""Golden fixture for the instant-ngp sampler (tests/golden/ngp_sampler.npz), written by THE REFERENCE'S OWN
`VolumetricSampler` (nerfstudio/model_components/ray_samplers.py:385-519) and the reference's dense-path renderers, run
in the authoring container:

    python tests/golden/make_golden_ngp.py

nerfacc 0.5.2 (the reference's `OccGridEstimator`) is not installable here, so the sampler module is driven with a STUB
estimator whose `sampling(...)` answers nerfacc's call contract (keyword names and defaults as VolumetricSampler passes
them, ray_samplers.py:481-493) with the samples of oracle/packed_oracle.py. What the fixture therefore pins is everything
of the sampler AROUND the estimator — the reference's own code: which arguments reach the estimator (t_min / t_max from
the bundle's nears / fars, far_plane None -> 1e10, stratified == training), the `sigma_fn` it builds (positions at the
interval midpoints, the field's density, squeeze), the fake sample of an empty result, and the packed `RaySamples` it
assembles (origins, directions, starts, ends, pixel_area, camera_indices gathered by ray index). The marcher's sample
placement stays pinned to the restatement only (oracle/packed_oracle.py header).
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_refstubs"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.model_components.ray_samplers import VolumetricSampler  # noqa: E402

from oracle import nerfacto_oracle as orc  # noqa: E402
from oracle import packed_oracle as po  # noqa: E402

ROI = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]


class OracleEstimator(torch.nn.Module):
    """nerfacc.OccGridEstimator's `sampling` contract answered by the oracle (fixed stratification draw `jitter`)."""

    def __init__(self, binaries, occs_mean, jitter):
        super().__init__()
        self.binaries, self.occs_mean, self.jitter = binaries, occs_mean, jitter
        self.calls = []

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane=0.0, far_plane=1e10, t_min=None, t_max=None,
                 render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0, stratified=False, cone_angle=0.0):
        self.calls.append(dict(near_plane=near_plane, far_plane=far_plane, render_step_size=render_step_size,
                               early_stop_eps=early_stop_eps, alpha_thre=alpha_thre, stratified=stratified, cone_angle=cone_angle,
                               has_sigma_fn=sigma_fn is not None, has_t=t_min is not None))
        idx, ts, te = po.occgrid_march(rays_o.numpy(), rays_d.numpy(), self.binaries, ROI, render_step_size, near_plane=near_plane,
                                       far_plane=far_plane, t_min=None if t_min is None else t_min.numpy(),
                                       t_max=None if t_max is None else t_max.numpy(), cone_angle=cone_angle,
                                       jitter=self.jitter if stratified else None)
        idx, ts, te = torch.from_numpy(idx), torch.from_numpy(ts), torch.from_numpy(te)
        if alpha_thre > 0.0:
            alpha_thre = min(alpha_thre, self.occs_mean)
        if sigma_fn is not None and ts.shape[0] > 0:
            sig = sigma_fn(ts, te, idx)
            keep = po.render_visibility_from_density(ts, te, sig, idx, rays_o.shape[0], early_stop_eps, alpha_thre)
            idx, ts, te = idx[keep], ts[keep], te[keep]
        return idx, ts, te


def main():
    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 12), prop_grids=(), num_images=4, average_init_density=1.0)
    seed, table_std, density_gain = 31, 0.5, 40.0
    params = orc.init_params(cfg, seed=seed, table_std=table_std)
    with torch.no_grad():
        params["field.mlp_base.model.1.layers.1.bias"][0] = float(np.log(density_gain))

    def density_fn(positions):  # NerfactoField.density_fn (fields/base_field.py:48-68): [n,3] -> [n,1]
        pos, sel = orc.normalise_positions(positions.reshape(-1, 3), True)
        g = cfg.main_grid
        enc = orc.hashgrid_encode(pos, params["field.mlp_base.model.0.hash_table"], g.scalings(), g.table_size)
        pre = orc.mlp_forward(enc, params, "field.mlp_base.model.1.")[:, 0]
        return (cfg.average_init_density * torch.exp(pre) * sel)[:, None]

    rs = np.random.RandomState(5)
    n, levels, res = 72, 2, 16
    B = rs.rand(levels, res, res, res) > 0.7
    o = (rs.standard_normal((n, 3)) * 0.6).astype(np.float32)
    d = rs.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[3] = [9.0, 9.0, 9.0]
    d[3] = [1.0, 0.0, 0.0]  # misses every level: a ray without samples
    jitter = rs.uniform(0, 1, n).astype(np.float32)
    nears = rs.uniform(0.0, 0.2, (n, 1)).astype(np.float32)
    fars = rs.uniform(2.0, 6.0, (n, 1)).astype(np.float32)
    cams = rs.randint(0, 4, (n, 1)).astype(np.int64)
    pix = rs.uniform(1e-6, 2e-6, (n, 1)).astype(np.float32)
    occs_mean = 0.004
    out = dict(seed=seed, table_std=table_std, density_gain=density_gain, binaries=B, origins=o, directions=d, jitter=jitter,
               nears=nears, fars=fars, cams=cams, pixel_area=pix, occs_mean=occs_mean, step=0.03, cone=0.004, alpha_thre=0.01)
    for mode in ("train", "eval", "train_nobounds"):
        est = OracleEstimator(B, occs_mean, jitter)
        sampler = VolumetricSampler(occupancy_grid=est, density_fn=density_fn)
        sampler.train(mode != "eval")
        with_bounds = mode != "train_nobounds"
        rb = RayBundle(origins=torch.from_numpy(o), directions=torch.from_numpy(d), pixel_area=torch.from_numpy(pix),
                       camera_indices=torch.from_numpy(cams), nears=torch.from_numpy(nears) if with_bounds else None,
                       fars=torch.from_numpy(fars) if with_bounds else None)
        samples, ray_indices = sampler(ray_bundle=rb, render_step_size=0.03, near_plane=0.05,
                                       far_plane=None if mode == "train_nobounds" else 1e3, alpha_thre=0.01, cone_angle=0.004)
        call = est.calls[-1]
        assert call["stratified"] == (mode != "eval") and call["has_sigma_fn"] == (mode != "eval") and call["has_t"] == with_bounds
        assert call["far_plane"] == (1e10 if mode == "train_nobounds" else 1e3)
        f = samples.frustums
        out.update({f"{mode}_ray_indices": ray_indices.numpy(), f"{mode}_starts": f.starts.numpy(), f"{mode}_ends": f.ends.numpy(),
                    f"{mode}_origins": f.origins.numpy(), f"{mode}_directions": f.directions.numpy(),
                    f"{mode}_pixel_area": f.pixel_area.numpy(), f"{mode}_camera_indices": samples.camera_indices.numpy()})
        print(mode, "samples:", ray_indices.numel(), "rays with samples:", len(np.unique(ray_indices.numpy())))
    # the empty result: one fake sample
    est = OracleEstimator(np.zeros_like(B), occs_mean, jitter)
    sampler = VolumetricSampler(occupancy_grid=est, density_fn=density_fn).train()
    rb = RayBundle(origins=torch.from_numpy(o), directions=torch.from_numpy(d), pixel_area=torch.from_numpy(pix),
                   camera_indices=torch.from_numpy(cams))
    samples, ray_indices = sampler(ray_bundle=rb, render_step_size=0.03)
    out.update(empty_ray_indices=ray_indices.numpy(), empty_starts=samples.frustums.starts.numpy(),
               empty_ends=samples.frustums.ends.numpy())
    np.savez_compressed(os.path.join(HERE, "ngp_sampler.npz"), **out)
    print("wrote ngp_sampler.npz")


if __name__ == "__main__":
    main()
