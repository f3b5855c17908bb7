"""Host wiring of this package's own model classes (nerfacto.py, instant_ngp.py) on the CPU: the kernel wrappers replaced
by the oracle's torch restatements (tests/cpu_kernels.py), so that the module graph — sampler -> fields -> renderers -> losses,
the eval switches, the normals options, the packed instant-ngp path — runs end to end without a GPU and is compared with the
oracle's own forward on the same parameters and draws. What the GPU tier checks on the kernels, this tier checks on the
glue (argument order, shapes, dictionary keys, autograd wiring); no reference tree needed."""
import numpy as np
import pytest
import torch

import cpu_kernels
from oracle import nerfacto_oracle as orc
from oracle import packed_oracle as po

T = torch.from_numpy


def _cfg(num_images, predict_normals=False):
    c = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 10),
                        prop_grids=(orc.HashGridCfg(5, 16, 128, 8), orc.HashGridCfg(5, 16, 256, 8)), num_images=num_images)
    c.predict_normals = predict_normals
    return c


def _model(cfg, params, **kw):
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    mc = NerfactoModelConfig(
        log2_hashmap_size=cfg.main_grid.log2_hashmap_size, predict_normals=cfg.predict_normals,
        proposal_net_args_list=[{"hidden_dim": cfg.prop_hidden_dim, "log2_hashmap_size": g.log2_hashmap_size, "num_levels": g.num_levels,
                                 "max_res": g.max_res, "use_linear": False} for g in cfg.prop_grids],
        average_init_density=cfg.average_init_density, **kw)
    model = NerfactoModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
    sd = {k: v.detach().clone() for k, v in params.items()}
    for i in range(len(cfg.prop_grids)):
        sd[f"proposal_networks.{i}.mlp_base.0.hash_table"] = sd[f"proposal_networks.{i}.encoding.hash_table"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return model


@pytest.mark.parametrize("predict_normals", [False, True])
def test_nerfacto_model_module_graph_equals_the_oracle(monkeypatch, predict_normals):
    from nerfstudio_amd.cameras.rays import RayBundle

    cfg = _cfg(5, predict_normals)
    params = orc.init_params(cfg, seed=41, table_std=0.5)
    model = _model(cfg, params)
    n = 20
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=16)
    rs = np.random.RandomState(3)
    jit = [T(rs.uniform(0, 1, (n, 1)).astype(np.float32)) for _ in range(3)]

    def bundle():
        return RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None])

    with cpu_kernels.installed(monkeypatch):
        for mode in ("train", "eval"):
            training = mode == "train"
            model.train(training)
            model.zero_grad(set_to_none=True)
            model.set_step(137)
            with (torch.enable_grad() if training else torch.no_grad()):
                out = model(bundle(), jitters=jit if training else None)
            p = {k: v.clone().requires_grad_(training) for k, v in params.items()}
            with (torch.enable_grad() if training else torch.no_grad()):
                ref = orc.nerfacto_forward(p, cfg, o, d, cam, jit, training=training, anneal=model.proposal_sampler._anneal)
            keys = ("rgb", "accumulation", "expected_depth", "depth") + (("normals", "pred_normals") if predict_normals else ())
            for k in keys:
                np.testing.assert_allclose(out[k].detach().numpy().reshape(n, -1), ref[k].detach().numpy().reshape(n, -1),
                                           atol=1e-5, err_msg=f"{mode} {k}")
            assert ("weights_list" in out) == training
            if training:
                batch = {"image": tgt}
                losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
                lr = orc.nerfacto_losses(ref, tgt, cfg)
                assert set(losses) == set(lr)
                for k in lr:
                    np.testing.assert_allclose(float(losses[k].detach()), float(lr[k].detach()), rtol=1e-5, atol=1e-12, err_msg=k)
                sum(losses.values()).backward()
                sum(lr.values()).backward()
                named = dict(model.named_parameters())
                for k, v in p.items():
                    if v.grad is None:
                        continue
                    a, b = named[k].grad.numpy(), v.grad.numpy()
                    assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(b), 1e-30), k
        # the chunked full-image render of the module loop (CPU tensors never take the device-side runner)
        model.eval()
        img = model.get_outputs_for_camera_ray_bundle(
            RayBundle(origins=o.reshape(4, 5, 3), directions=d.reshape(4, 5, 3), pixel_area=torch.full((4, 5, 1), 1e-6),
                      camera_indices=cam.reshape(4, 5, 1)))
        assert img["rgb"].shape == (4, 5, 3) and img["prop_depth_1"].shape == (4, 5, 1)


def test_nerfacto_model_gradient_scaling_and_per_edge_jitter_wiring(monkeypatch):
    from nerfstudio_amd.cameras.rays import RayBundle

    cfg = _cfg(5)
    params = orc.init_params(cfg, seed=43, table_std=0.5)
    model = _model(cfg, params, use_gradient_scaling=True, use_single_jitter=False)
    model.proposal_sampler.initial_sampler.single_jitter = False
    model.proposal_sampler.pdf_sampler.single_jitter = False
    n = 12
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=17)
    rs = np.random.RandomState(4)
    jit = [T(rs.uniform(0, 1, (n, s + 1)).astype(np.float32)) for s in (256, 96, 48)]
    with cpu_kernels.installed(monkeypatch):
        model.train()
        out = model(RayBundle(origins=o, directions=d, pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None]), jitters=jit)
        batch = {"image": tgt}
        losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
        sum(losses.values()).backward()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        ref = orc.nerfacto_forward(p, cfg, o, d, cam, jit, training=True, anneal=model.proposal_sampler._anneal, use_gradient_scaling=True)
        sum(orc.nerfacto_losses(ref, tgt, cfg).values()).backward()
        np.testing.assert_allclose(out["rgb"].detach().numpy(), ref["rgb"].detach().numpy(), atol=1e-5)
        a, b = model.field.mlp_base.encoding.hash_table.grad.numpy(), p["field.mlp_base.model.0.hash_table"].grad.numpy()
        assert np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(b)


@pytest.mark.parametrize("background,scaling", [("random", False), ("white", True)])
def test_ngp_model_module_graph_equals_the_oracle(monkeypatch, background, scaling):
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.instant_ngp import DynamicBatch, InstantNGPModelConfig, NGPModel

    ocfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 10), prop_grids=(), num_images=4, average_init_density=1.0)
    params = orc.init_params(ocfg, seed=45, table_std=0.5)
    mc = InstantNGPModelConfig(grid_resolution=16, grid_levels=2, log2_hashmap_size=10, background_color=background, cone_angle=0.0,
                               render_step_size=0.05, use_gradient_scaling=scaling)
    model = NGPModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4)
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in params.items() if k.startswith("field.")}, strict=False)
    assert not unexpected, unexpected
    model.train()
    n = 24
    o, d, cam, tgt = orc.synthetic_rays(n, 4, seed=18)
    with cpu_kernels.installed(monkeypatch):
        model.update_occupancy_grid(step=0)
        jit = torch.rand(n, generator=torch.Generator().manual_seed(2))
        out = model(RayBundle(origins=o, directions=d, pixel_area=torch.full((n, 1), 1e-6), camera_indices=cam[:, None]), jitter=jit)
        batch = {"image": tgt}
        metrics = model.get_metrics_dict(out, batch)
        torch.manual_seed(7)
        loss = model.get_loss_dict(out, batch, metrics)["rgb_loss"]
        loss.backward()
        # the dynamic batch size follows the sample count (pipelines/dynamic_batch.py:71-95)
        db = DynamicBatch(target_num_samples=1 << 12, max_num_samples_per_ray=1 << 6)
        assert db.update(metrics) == int(64 * ((1 << 12) / int(metrics["num_samples_per_batch"])))
        # ---- the oracle on the samples the sampler placed
        B = model.occupancy_grid.binaries.numpy().astype(bool)
        ri, ts, te = po.occgrid_march(o.numpy(), d.numpy(), B, [-1.0, -1, -1, 1, 1, 1], mc.render_step_size, near_plane=mc.near_plane,
                                      far_plane=mc.far_plane, cone_angle=0.0, jitter=jit.numpy())
        ri, ts, te = T(ri).long(), T(ts), T(te)
        with torch.no_grad():
            sig = orc.nerfacto_field(o[ri] + d[ri] * ((ts + te) / 2)[:, None], d[ri], cam[ri], params, ocfg, training=True)[0]
            keep = po.render_visibility_from_density(ts, te, sig, ri, n, 1e-4, min(mc.alpha_thre, model.occupancy_grid._occ_mean))
        ri, ts, te = ri[keep], ts[keep], te[keep]
        assert int(out["num_samples_per_ray"].sum()) == ri.numel() > 0
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        dens, rgb_s, _ = orc.nerfacto_field(o[ri] + d[ri] * ((ts + te) / 2)[:, None], d[ri], cam[ri], p, ocfg, training=True)
        if scaling:  # model_components/losses.py:550-569 on packed samples
            sc = torch.square((ts + te) / 2).clamp(0, 1)
            dens, rgb_s = orc._GradientScalerFn.apply(dens, sc), orc._GradientScalerFn.apply(rgb_s, sc[:, None])
        w = po.render_weight_from_density(ts, te, dens, ri, n)[0]
        comp, acc, dep = po.composite_packed(rgb_s, w, ts, te, ri, n, background=background, training=True)
        np.testing.assert_allclose(out["rgb"].detach().numpy(), comp.detach().numpy(), atol=1e-6)
        np.testing.assert_allclose(out["depth"].detach().numpy(), dep.detach().numpy(), rtol=1e-5, atol=1e-6)
        pred = comp
        if background == "random":
            torch.manual_seed(7)
            pred = comp + torch.rand_like(comp) * (1.0 - acc)
        ref_loss = torch.mean((tgt - pred) ** 2)
        np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=1e-5)
        ref_loss.backward()
        named = dict(model.named_parameters())
        for k in ("field.mlp_base.model.0.hash_table", "field.mlp_head.layers.2.weight"):
            a, b = named[k].grad.numpy(), p[k].grad.numpy()
            assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(b), 1e-30), k
        model.eval()
        with torch.no_grad():
            ev = model.get_outputs_for_camera_ray_bundle(
                RayBundle(origins=o.reshape(4, 6, 3), directions=d.reshape(4, 6, 3), pixel_area=torch.full((4, 6, 1), 1e-6),
                          camera_indices=cam.reshape(4, 6, 1)))
        assert ev["rgb"].shape == (4, 6, 3)
