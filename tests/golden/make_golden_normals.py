#!/usr/bin/env python3
"""Golden fixture for the normals options of nerfacto (`predict_normals`: analytic normals from the density gradient and
the predicted-normals head) — written by THE REFERENCE ITSELF (read-only import of /root/reference, torch path, CPU).
Authoring container only:

    python tests/golden/make_golden_normals.py      ->  tests/golden/normals.npz

Field level: `NerfactoField(use_pred_normals=True).forward(ray_samples, compute_normals=True)` (fields/base_field.py:113-133,
fields/nerfacto_field.py:181-191, 203-223, 287-295) in train and eval mode — NORMALS, PRED_NORMALS, density, rgb, the raw
gradient the normals are the direction of, and the parameter gradients of a seeded scalar.
Model level: the nerfacto graph with the two extra outputs and loss terms (models/nerfacto.py:325-344, :379-388) on 16
rays — rendered / shaded normals, the five losses, gradients of every parameter group.
Parameters come from oracle.nerfacto_oracle.init_params (numpy RandomState stream): the fixture stores the seed only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up the import path of the reference and its stubs)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from nerfstudio.cameras.rays import Frustums, RayBundle, RaySamples  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.spatial_distortions import SceneContraction  # noqa: E402
from nerfstudio.fields.nerfacto_field import NerfactoField  # noqa: E402
from nerfstudio.model_components.losses import distortion_loss, interlevel_loss, orientation_loss, pred_normal_loss  # noqa: E402
from nerfstudio.model_components.ray_samplers import ProposalNetworkSampler  # noqa: E402
from nerfstudio.model_components.renderers import NormalsRenderer, RGBRenderer  # noqa: E402
from nerfstudio.model_components.scene_colliders import NearFarCollider  # noqa: E402
from nerfstudio.model_components.shaders import NormalsShader  # noqa: E402

orc = mg.orc


def normals_cfg(num_images):
    cfg = mg.small_cfg(main_log2=10, prop_log2=8, num_images=num_images)
    cfg.predict_normals = True
    return cfg


def build_field(cfg, params):
    fld = NerfactoField(
        mg.AABB, num_images=cfg.num_images, num_levels=cfg.main_grid.num_levels, base_res=cfg.main_grid.min_res,
        max_res=cfg.main_grid.max_res, log2_hashmap_size=cfg.main_grid.log2_hashmap_size,
        spatial_distortion=SceneContraction(order=float("inf")), implementation="torch",
        average_init_density=cfg.average_init_density,
        use_average_appearance_embedding=cfg.use_average_appearance_embedding, use_pred_normals=True)
    sd = {k[len("field."):]: v.clone() for k, v in params.items() if k.startswith("field.")}
    missing, unexpected = fld.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("aabb" in m or "max_res" in m or "num_levels" in m or "log2" in m) for m in missing), missing
    return fld


def field_grads(fld, out, prefix):
    out[f"{prefix}_dtable"] = fld.mlp_base.model[0].hash_table.grad
    for j in range(2):
        out[f"{prefix}_base_dW{j}"] = fld.mlp_base.model[1].layers[j].weight.grad
        out[f"{prefix}_base_db{j}"] = fld.mlp_base.model[1].layers[j].bias.grad
    for j in range(3):
        out[f"{prefix}_head_dW{j}"] = fld.mlp_head.layers[j].weight.grad
        out[f"{prefix}_head_db{j}"] = fld.mlp_head.layers[j].bias.grad
        out[f"{prefix}_pn_dW{j}"] = fld.mlp_pred_normals.layers[j].weight.grad
        out[f"{prefix}_pn_db{j}"] = fld.mlp_pred_normals.layers[j].bias.grad
    out[f"{prefix}_pnhead_dW"] = fld.field_head_pred_normals.net.weight.grad
    out[f"{prefix}_pnhead_db"] = fld.field_head_pred_normals.net.bias.grad
    out[f"{prefix}_demb"] = fld.embedding_appearance.embedding.weight.grad


def gen():
    out = {}
    # ---- field level ------------------------------------------------------------------------------------------------
    cfg = normals_cfg(7)
    seed, std = 15, 0.5
    params = orc.init_params(cfg, seed=seed, table_std=std)
    fld = build_field(cfg, params)
    rs = np.random.RandomState(33)
    M = 384
    pos, dirs, cam = mg.field_inputs(rs, M, cfg)
    out.update(f_seed=seed, f_table_std=std, f_num_images=cfg.num_images, f_positions=pos, f_directions=dirs, f_cams=cam)
    R, S = M // 4, 4
    for mode in ("train", "eval"):
        fld.train(mode == "train")
        fld.zero_grad()
        fr = Frustums(origins=pos.reshape(R, S, 3).clone(), directions=dirs.reshape(R, S, 3), starts=torch.zeros(R, S, 1),
                      ends=torch.zeros(R, S, 1), pixel_area=torch.ones(R, S, 1))
        rsamp = RaySamples(frustums=fr, camera_indices=cam.reshape(R, S, 1))
        if mode == "eval":
            with torch.no_grad():  # as Model.get_outputs_for_camera_ray_bundle calls it (base_model.py:177)
                fo = fld(rsamp, compute_normals=True)
        else:
            fo = fld(rsamp, compute_normals=True)
        with torch.enable_grad():
            raw = torch.autograd.grad(fld._density_before_activation, fld._sample_locations,
                                      grad_outputs=torch.ones_like(fld._density_before_activation), retain_graph=True)[0]
        out[f"f_{mode}_density"] = fo[FieldHeadNames.DENSITY].reshape(M)
        out[f"f_{mode}_rgb"] = fo[FieldHeadNames.RGB].reshape(M, 3)
        out[f"f_{mode}_normals"] = fo[FieldHeadNames.NORMALS].reshape(M, 3)
        out[f"f_{mode}_pred_normals"] = fo[FieldHeadNames.PRED_NORMALS].reshape(M, 3)
        out[f"f_{mode}_density_gradient"] = raw.reshape(M, 3)
        assert not fo[FieldHeadNames.NORMALS].requires_grad  # first order only: constants for the losses
        if mode == "train":
            g1 = torch.from_numpy(rs.standard_normal((M,)).astype(np.float32))
            g2 = torch.from_numpy(rs.standard_normal((M, 3)).astype(np.float32))
            g3 = torch.from_numpy(rs.standard_normal((M, 3)).astype(np.float32))
            ((fo[FieldHeadNames.DENSITY].reshape(M) * g1).sum() + (fo[FieldHeadNames.RGB].reshape(M, 3) * g2).sum()
             + (fo[FieldHeadNames.PRED_NORMALS].reshape(M, 3) * g3).sum()).backward()
            out.update(f_g_density=g1, f_g_rgb=g2, f_g_pred_normals=g3)
            field_grads(fld, out, "f")

    # ---- model level ------------------------------------------------------------------------------------------------
    cfg = normals_cfg(5)
    seed = 19
    params = orc.init_params(cfg, seed=seed, table_std=std)
    fld = build_field(cfg, params)
    _, props = mg.build_reference(cfg, {k: v for k, v in params.items() if "pred_normals" not in k})
    N = 16
    o, d, cam, tgt = orc.synthetic_rays(N, cfg.num_images, seed=4)
    o[N // 2:] *= 6.0
    rs = np.random.RandomState(71)
    jit = [torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32)) for _ in range(3)]
    out.update(m_seed=seed, m_table_std=std, m_num_images=cfg.num_images, m_origins=o, m_directions=d, m_cams=cam,
               m_target=tgt, m_j0=jit[0], m_j1=jit[1], m_j2=jit[2])
    sampler = ProposalNetworkSampler(num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96),
                                     num_proposal_network_iterations=2, single_jitter=True)
    collider = NearFarCollider(0.05, 1000.0)
    for mode in ("train", "eval"):
        training = mode == "train"
        for m in (fld, props, sampler, collider):
            m.train(training)
        fld.zero_grad()
        props.zero_grad()
        rgb_r = RGBRenderer("last_sample")
        rgb_r.train(training)
        rb = collider(RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((N, 1), 1e-6),
                                camera_indices=cam[:, None]))
        ctx = torch.enable_grad() if training else torch.no_grad()
        with ctx:
            with mg.replay_rand(jit if training else []):
                rsamp, wl, rsl = sampler(rb, density_fns=[p.density_fn for p in props])
            fo = fld(rsamp, compute_normals=True)
            w = rsamp.get_weights(fo[FieldHeadNames.DENSITY])
            wl.append(w)
            rsl.append(rsamp)
            rgb = rgb_r(rgb=fo[FieldHeadNames.RGB], weights=w)
            normals = NormalsRenderer()(normals=fo[FieldHeadNames.NORMALS], weights=w)
            pred_normals = NormalsRenderer()(fo[FieldHeadNames.PRED_NORMALS], weights=w)
            out[f"m_{mode}_rgb"] = rgb
            out[f"m_{mode}_normals"] = NormalsShader()(normals)
            out[f"m_{mode}_pred_normals"] = NormalsShader()(pred_normals)
            out[f"m_{mode}_normals_samples"] = fo[FieldHeadNames.NORMALS]
            out[f"m_{mode}_pred_normals_samples"] = fo[FieldHeadNames.PRED_NORMALS]
            out[f"m_{mode}_w"] = w[..., 0]
            raw = None
        with torch.enable_grad():
            raw = torch.autograd.grad(fld._density_before_activation, fld._sample_locations,
                                      grad_outputs=torch.ones_like(fld._density_before_activation), retain_graph=True)[0]
        out[f"m_{mode}_density_gradient"] = raw
        if training:
            r_or = orientation_loss(w.detach(), fo[FieldHeadNames.NORMALS], rb.directions)
            r_pn = pred_normal_loss(w.detach(), fo[FieldHeadNames.NORMALS].detach(), fo[FieldHeadNames.PRED_NORMALS])
            l_rgb = torch.nn.functional.mse_loss(tgt, rgb)
            l_int = interlevel_loss(wl, rsl)
            l_dist = 0.002 * distortion_loss(wl, rsl)
            l_or = 0.0001 * torch.mean(r_or)
            l_pn = 0.001 * torch.mean(r_pn)
            (l_rgb + l_int + l_dist + l_or + l_pn).backward()
            out.update(m_loss_rgb=l_rgb, m_loss_interlevel=l_int, m_loss_distortion=l_dist, m_loss_orientation=l_or,
                       m_loss_pred_normal=l_pn, m_rendered_orientation=r_or, m_rendered_pred_normal=r_pn)
            field_grads(fld, out, "m")
            for i, p in enumerate(props):
                out[f"m_prop{i}_dtable"] = p.encoding.hash_table.grad
    mg.save("normals", **out)


if __name__ == "__main__":
    gen()
