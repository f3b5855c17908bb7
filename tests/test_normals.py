"""The normals options of nerfacto (`predict_normals`): analytic normals = minus the normalised gradient of the density
pre-activation with respect to the normalised sample positions (fields/base_field.py:79-99, fields/nerfacto_field.py:215-223),
the predicted-normals head (nerfacto_field.py:181-191, 287-295, field_heads.py:190-206), their renderer / shader and the two
loss terms (models/nerfacto.py:325-344, 379-388).

tests/golden/normals.npz was written by the reference itself (tests/golden/make_golden_normals.py). Checked against it:
 * the CPU oracle (field level: to fp32 rounding; model level: within the tolerances below);
 * the package's composed field on CPU with the kernels replaced by the oracle's torch restatements — the WIRING of the
   composed path (shapes, graph, parameter names), which needs no GPU;
 * on the MI355X: the same field and the whole model on the kernels (`-m gpu`).

Tolerances. The hash encoding is piecewise trilinear: its gradient — hence the normal — jumps at every cell boundary of
every level (finest cell 1 / 2048 of the normalised cube). Wherever the sample POSITIONS are inputs (field level) two fp32
implementations agree to rounding; downstream of a sampler whose bin edges differ in the last bit (model level) the odd sample
falls on the other side of a boundary and its normal changes by a finite amount. So at model level: per-sample normals within
1e-3 on >= 97 % of the samples, rendered normals 1e-2, loss terms 1e-3 relative, gradients relative L2 <= 2e-2.
"""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc

T = torch.from_numpy


def _cfg(num_images):
    c = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 10),
                        prop_grids=(orc.HashGridCfg(5, 16, 128, 8), orc.HashGridCfg(5, 16, 256, 8)), num_images=int(num_images))
    c.predict_normals = True
    return c


def _rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


FIELD_GRADS = [("field.mlp_base.model.0.hash_table", "dtable"), ("field.embedding_appearance.embedding.weight", "demb"),
               ("field.field_head_pred_normals.net.weight", "pnhead_dW"), ("field.field_head_pred_normals.net.bias", "pnhead_db")] + \
    [(f"field.mlp_base.model.1.layers.{j}.{w}", f"base_d{c}{j}") for j in range(2) for w, c in (("weight", "W"), ("bias", "b"))] + \
    [(f"field.mlp_head.layers.{j}.{w}", f"head_d{c}{j}") for j in range(3) for w, c in (("weight", "W"), ("bias", "b"))] + \
    [(f"field.mlp_pred_normals.layers.{j}.{w}", f"pn_d{c}{j}") for j in range(3) for w, c in (("weight", "W"), ("bias", "b"))]


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the oracle against the reference's fixture
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_field_normals_match_the_reference(golden):
    g = golden("normals")
    cfg = _cfg(g["f_num_images"])
    params = orc.init_params(cfg, seed=int(g["f_seed"]), table_std=float(g["f_table_std"]))
    assert "field.mlp_pred_normals.layers.2.weight" in params and params["field.field_head_pred_normals.net.weight"].shape == (3, 64)
    # the extra tensors are drawn last: every other parameter is the one the plain configuration gets
    cfg0 = _cfg(g["f_num_images"])
    cfg0.predict_normals = False
    p0 = orc.init_params(cfg0, seed=int(g["f_seed"]), table_std=float(g["f_table_std"]))
    assert all(torch.equal(p0[k], params[k]) for k in p0)
    for mode in ("train", "eval"):
        p = {k: v.clone().requires_grad_(mode == "train") for k, v in params.items()}
        nrm = {}
        with (torch.enable_grad() if mode == "train" else torch.no_grad()):
            d, rgb, _ = orc.nerfacto_field(T(g["f_positions"]), T(g["f_directions"]), T(g["f_cams"]), p, cfg,
                                           training=(mode == "train"), normals_out=nrm)
        np.testing.assert_allclose(_np(d), g[f"f_{mode}_density"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(_np(rgb), g[f"f_{mode}_rgb"], atol=1e-6)
        np.testing.assert_allclose(_np(nrm["normals"]), g[f"f_{mode}_normals"], atol=1e-6)
        np.testing.assert_allclose(_np(nrm["pred_normals"]), g[f"f_{mode}_pred_normals"], atol=1e-6)
        assert not nrm["normals"].requires_grad  # first order only (base_field.py:92-97 has no create_graph)
        unit = np.linalg.norm(g[f"f_{mode}_normals"], axis=-1)
        zero = np.linalg.norm(g[f"f_{mode}_density_gradient"], axis=-1) == 0  # an exactly flat spot normalises to 0
        assert np.allclose(unit[~zero], 1.0, atol=1e-5) and np.all(unit[zero] == 0) and zero.sum() <= 2
        if mode == "train":
            ((d * T(g["f_g_density"])).sum() + (rgb * T(g["f_g_rgb"])).sum()
             + (nrm["pred_normals"] * T(g["f_g_pred_normals"])).sum()).backward()
            for name, key in FIELD_GRADS:
                assert _rel(p[name].grad, g["f_" + key]) < 1e-5, name


def test_oracle_model_normals_match_the_reference(golden):
    g = golden("normals")
    cfg = _cfg(g["m_num_images"])
    params = orc.init_params(cfg, seed=int(g["m_seed"]), table_std=float(g["m_table_std"]))
    jit = [T(g["m_j0"]), T(g["m_j1"]), T(g["m_j2"])]
    for mode in ("train", "eval"):
        p = {k: v.clone().requires_grad_(mode == "train") for k, v in params.items()}
        with (torch.enable_grad() if mode == "train" else torch.no_grad()):
            out = orc.nerfacto_forward(p, cfg, T(g["m_origins"]), T(g["m_directions"]), T(g["m_cams"]), jit,
                                       training=(mode == "train"))
        _check_model_outputs(out["rgb"], out["normals"], out["pred_normals"], out["normals_samples"],
                             out["pred_normals_samples"], g, mode)
        if mode == "train":
            losses = orc.nerfacto_losses(out, T(g["m_target"]), cfg)
            _check_losses(losses, g)
            sum(losses.values()).backward()
            for name, key in FIELD_GRADS:
                assert _rel(p[name].grad, g["m_" + key]) < 2e-2, name
            assert _rel(p["proposal_networks.0.encoding.hash_table"].grad, g["m_prop0_dtable"]) < 2e-2


def _check_model_outputs(rgb, normals, pred_normals, n_samples, p_samples, g, mode):
    np.testing.assert_allclose(_np(rgb), g[f"m_{mode}_rgb"], atol=1e-4, err_msg="rgb")
    np.testing.assert_allclose(_np(normals), g[f"m_{mode}_normals"], atol=1e-2, err_msg="rendered normals")
    np.testing.assert_allclose(_np(pred_normals), g[f"m_{mode}_pred_normals"], atol=2e-3, err_msg="rendered predicted normals")
    dn = np.abs(_np(n_samples) - g[f"m_{mode}_normals_samples"]).max(axis=-1)
    assert (dn < 1e-3).mean() >= 0.97, f"{mode}: {(dn >= 1e-3).sum()} of {dn.size} per-sample normals differ"
    dp = np.abs(_np(p_samples) - g[f"m_{mode}_pred_normals_samples"]).max(axis=-1)
    assert (dp < 1e-3).mean() >= 0.97


def _rendered_normals_close(a, b, what):
    """Rendered (weight-summed, renormalised) analytic normals of the kernels' own sampler against the reference's: the bin
    edges of the two samplers differ in the last bits, so on the odd ray a dominant sample sits in the neighbouring cell of
    a fine level and the ray's normal moves by a few hundredths (module docstring). Nearly all rays within 1e-2, none far."""
    dn = np.abs(_np(a) - b).max(axis=-1)
    assert (dn < 1e-2).sum() >= dn.size - max(2, dn.size // 8) and dn.max() < 0.25 and np.median(dn) < 2e-3, (what, np.sort(dn)[-4:])


def _check_losses(losses, g, normals_rtol=1e-3):
    """(the two terms built on the analytic normals inherit their cell-boundary jumps: `normals_rtol`)"""
    for k, key, rtol in (("rgb_loss", "m_loss_rgb", 1e-3), ("interlevel_loss", "m_loss_interlevel", 1e-3),
                         ("distortion_loss", "m_loss_distortion", 1e-3), ("orientation_loss", "m_loss_orientation", normals_rtol),
                         ("pred_normal_loss", "m_loss_pred_normal", normals_rtol)):
        np.testing.assert_allclose(float(losses[k].detach()), float(g[key]), rtol=rtol, atol=1e-10, err_msg=k)


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the package's composed field with the kernels replaced by the oracle's restatements (wiring only)
# ---------------------------------------------------------------------------------------------------------------------
def _field_with_params(params, cfg, device="cpu"):
    from nerfstudio_amd.field_components.spatial_distortions import SceneContraction
    from nerfstudio_amd.fields.nerfacto_field import NerfactoField

    fld = NerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=cfg.num_images,
                        log2_hashmap_size=cfg.main_grid.log2_hashmap_size, spatial_distortion=SceneContraction(order=float("inf")),
                        average_init_density=cfg.average_init_density, use_average_appearance_embedding=True,
                        use_pred_normals=True)
    sd = {k[len("field."):]: v.clone() for k, v in params.items() if k.startswith("field.")}
    missing, unexpected = fld.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(any(s in m for s in ("aabb", "max_res", "num_levels", "log2_hashmap_size")) for m in missing), missing
    return fld.to(device)


def _run_field(fld, g, mode, device):
    from nerfstudio_amd.cameras.rays import Frustums, RaySamples
    from nerfstudio_amd.field_components.field_heads import FieldHeadNames

    M = g["f_positions"].shape[0]
    R, S = M // 4, 4
    fld.train(mode == "train")
    fld.zero_grad()
    z, o = torch.zeros(R, S, 1, device=device), torch.ones(R, S, 1, device=device)
    fr = Frustums(origins=T(g["f_positions"]).to(device).reshape(R, S, 3), directions=T(g["f_directions"]).to(device).reshape(R, S, 3),
                  starts=z, ends=z.clone(), pixel_area=o)
    rs = RaySamples(frustums=fr, camera_indices=T(g["f_cams"]).to(device).reshape(R, S, 1))
    if mode == "eval":
        with torch.no_grad():
            fo = fld(rs, compute_normals=True)
    else:
        fo = fld(rs, compute_normals=True)
    return fo, FieldHeadNames, M


def _check_field(fld, fo, H, M, g, mode, device, tol, gtol):
    np.testing.assert_allclose(_np(fo[H.DENSITY]).reshape(M), g[f"f_{mode}_density"], rtol=10 * tol, atol=tol)
    np.testing.assert_allclose(_np(fo[H.RGB]).reshape(M, 3), g[f"f_{mode}_rgb"], atol=10 * tol)
    raw = np.linalg.norm(g[f"f_{mode}_density_gradient"], axis=-1)
    keep = raw > 1e-3 * np.median(raw)  # the direction of a (nearly) vanishing gradient is not a number to compare
    assert keep.sum() >= M - 4
    np.testing.assert_allclose(_np(fo[H.NORMALS]).reshape(M, 3)[keep], g[f"f_{mode}_normals"][keep], atol=100 * tol)
    np.testing.assert_allclose(_np(fo[H.PRED_NORMALS]).reshape(M, 3), g[f"f_{mode}_pred_normals"], atol=100 * tol)
    assert not fo[H.NORMALS].requires_grad
    if mode == "train":
        ((fo[H.DENSITY].reshape(M) * T(g["f_g_density"]).to(device)).sum() + (fo[H.RGB].reshape(M, 3) * T(g["f_g_rgb"]).to(device)).sum()
         + (fo[H.PRED_NORMALS].reshape(M, 3) * T(g["f_g_pred_normals"]).to(device)).sum()).backward()
        named = dict(fld.named_parameters())
        for name, key in FIELD_GRADS:
            assert _rel(named[name[len("field."):]].grad, g["f_" + key]) < gtol, name


def test_composed_field_wiring_on_cpu(golden, monkeypatch):
    from nerfstudio_amd import functional as F

    def hash_enc(x, table, grid):
        shape = x.shape[:-1]
        return orc.hashgrid_encode(x.reshape(-1, 3), table, grid.scalings(), grid.table_size).view(*shape, grid.out_dim)

    def linear(x, W, b, activation=None):
        y = x @ W.t() + (b if b is not None else 0.0)
        return {None: lambda v: v, "relu": torch.relu, "sigmoid": torch.sigmoid}[activation](y)

    monkeypatch.setattr(F, "hashgrid_encode", hash_enc)
    monkeypatch.setattr(F, "linear", linear)
    monkeypatch.setattr(F, "sh4_encode", lambda d: orc.sh_levels4(d.detach()))
    monkeypatch.setattr(F, "nerf_encode", lambda spec, nf, lo, hi, inc=False: orc.nerf_encode(spec.positions, nf, lo, hi))
    g = golden("normals")
    cfg = _cfg(g["f_num_images"])
    params = orc.init_params(cfg, seed=int(g["f_seed"]), table_std=float(g["f_table_std"]))
    fld = _field_with_params(params, cfg)
    for mode in ("train", "eval"):
        fo, H, M = _run_field(fld, g, mode, "cpu")
        _check_field(fld, fo, H, M, g, mode, "cpu", tol=1e-7, gtol=1e-5)
    # without either option the field stays the fused pipeline (which has no CPU form): the switch is per call
    fld.use_pred_normals = False
    with pytest.raises(Exception):
        _run_field_plain(fld, g)


def _run_field_plain(fld, g):
    from nerfstudio_amd.cameras.rays import Frustums, RaySamples

    z = torch.zeros(4, 4, 1)
    fr = Frustums(origins=torch.zeros(4, 4, 3), directions=torch.ones(4, 4, 3), starts=z, ends=z, pixel_area=z + 1)
    return fld(RaySamples(frustums=fr, camera_indices=torch.zeros(4, 4, 1, dtype=torch.long)))


def test_normals_renderer_shader_and_losses_match_their_definitions():
    from nerfstudio_amd.model_components.losses import orientation_loss, pred_normal_loss
    from nerfstudio_amd.model_components.renderers import NormalsRenderer
    from nerfstudio_amd.model_components.shaders import NormalsShader

    rs = np.random.RandomState(0)
    w = T(rs.uniform(0, 0.2, (5, 7, 1)).astype(np.float32))
    n = torch.nn.functional.normalize(T(rs.standard_normal((5, 7, 3)).astype(np.float32)), dim=-1)
    p = torch.nn.functional.normalize(T(rs.standard_normal((5, 7, 3)).astype(np.float32)), dim=-1)
    v = torch.nn.functional.normalize(T(rs.standard_normal((5, 3)).astype(np.float32)), dim=-1)
    r = NormalsRenderer()(normals=n, weights=w)
    s = (w * n).sum(-2)
    assert torch.allclose(r, s / (s.norm(dim=-1, keepdim=True) + 1e-10)) and torch.allclose(r.norm(dim=-1), torch.ones(5), atol=1e-6)
    assert torch.equal(NormalsRenderer()(n, w, normalize=False), s)
    assert torch.equal(NormalsShader()(r), (r + 1) / 2) and torch.equal(NormalsShader()(r, w[:, 0]), (r + 1) / 2 * w[:, 0])
    ol = orientation_loss(w, n, v)  # only normals facing AWAY from the camera (n . -v < 0) are penalised
    ndv = (n * (-v)[:, None, :]).sum(-1)
    assert torch.allclose(ol, (w[..., 0] * torch.clamp(ndv, max=0.0) ** 2).sum(-1))
    assert torch.allclose(pred_normal_loss(w, n, p), (w[..., 0] * (1 - (n * p).sum(-1))).sum(-1))
    assert float(pred_normal_loss(w, n, n).abs().max()) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the kernels
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_field_normals_golden_gpu(golden):
    g = golden("normals")
    cfg = _cfg(g["f_num_images"])
    params = orc.init_params(cfg, seed=int(g["f_seed"]), table_std=float(g["f_table_std"]))
    fld = _field_with_params(params, cfg, "cuda")
    for mode in ("train", "eval"):
        fo, H, M = _run_field(fld, g, mode, "cuda")
        # positions are inputs here: the kernels and the reference see the same cells, so the normals agree to rounding
        _check_field(fld, fo, H, M, g, mode, "cuda", tol=1e-6, gtol=2e-4)
    # the same parameters through the fused pipeline (no normals): density and rgb of the two routes agree
    from nerfstudio_amd.field_components.field_heads import FieldHeadNames

    fld.use_pred_normals = False
    fld.train()
    fo_fused = _run_field_fused(fld, g)
    np.testing.assert_allclose(_np(fo_fused[FieldHeadNames.DENSITY]).reshape(-1), g["f_train_density"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(_np(fo_fused[FieldHeadNames.RGB]).reshape(-1, 3), g["f_train_rgb"], atol=1e-5)


def _run_field_fused(fld, g):
    from nerfstudio_amd.cameras.rays import Frustums, RaySamples

    M = g["f_positions"].shape[0]
    R, S = M // 4, 4
    z = torch.zeros(R, S, 1, device="cuda")
    fr = Frustums(origins=T(g["f_positions"]).cuda().reshape(R, S, 3), directions=T(g["f_directions"]).cuda().reshape(R, S, 3),
                  starts=z, ends=z.clone(), pixel_area=z + 1)
    return fld(RaySamples(frustums=fr, camera_indices=T(g["f_cams"]).cuda().reshape(R, S, 1)))


@pytest.mark.gpu
def test_model_normals_golden_gpu(golden):
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    g = golden("normals")
    cfg = _cfg(g["m_num_images"])
    params = orc.init_params(cfg, seed=int(g["m_seed"]), table_std=float(g["m_table_std"]))
    mc = NerfactoModelConfig(
        log2_hashmap_size=cfg.main_grid.log2_hashmap_size, predict_normals=True,
        proposal_net_args_list=[{"hidden_dim": cfg.prop_hidden_dim, "log2_hashmap_size": gr.log2_hashmap_size,
                                 "num_levels": gr.num_levels, "max_res": gr.max_res, "use_linear": False} for gr in cfg.prop_grids],
        average_init_density=cfg.average_init_density, appearance_embed_dim=cfg.appearance_embed_dim)
    model = NerfactoModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
    sd = {k: v.detach().clone() for k, v in params.items()}
    for i in range(2):
        sd[f"proposal_networks.{i}.mlp_base.0.hash_table"] = sd[f"proposal_networks.{i}.encoding.hash_table"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    model = model.cuda()
    N = g["m_origins"].shape[0]
    jit = [T(g[f"m_j{i}"]).cuda() for i in range(3)]
    for mode in ("train", "eval"):
        model.train(mode == "train")
        model.zero_grad()
        rb = RayBundle(origins=T(g["m_origins"]).cuda(), directions=T(g["m_directions"]).cuda(),
                       pixel_area=torch.full((N, 1), 1e-6).cuda(), camera_indices=T(g["m_cams"]).cuda()[:, None])
        if mode == "train":
            out = model(rb, jitters=jit)
        else:
            with torch.no_grad():
                out = model(rb)
        assert out["normals"].shape == (N, 3) and out["pred_normals"].shape == (N, 3)
        np.testing.assert_allclose(_np(out["rgb"]), g[f"m_{mode}_rgb"], atol=1e-4, err_msg="rgb")
        _rendered_normals_close(out["normals"], g[f"m_{mode}_normals"], mode)
        np.testing.assert_allclose(_np(out["pred_normals"]), g[f"m_{mode}_pred_normals"], atol=2e-3)
        if mode == "train":
            batch = {"image": T(g["m_target"]).cuda()}
            losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
            assert set(losses) == {"rgb_loss", "interlevel_loss", "distortion_loss", "orientation_loss", "pred_normal_loss"}
            _check_losses(losses, g, normals_rtol=1e-2)
            sum(losses.values()).backward()
            named = dict(model.named_parameters())
            for name, key in FIELD_GRADS:
                assert _rel(named[name].grad, g["m_" + key]) < 2e-2, name
            assert _rel(model.proposal_networks[0].encoding.hash_table.grad, g["m_prop0_dtable"]) < 2e-2
    # full-image eval render with normals: the module chunk loop (the device-side loop has no normals outputs)
    model.eval()
    from nerfstudio_amd import eval_render

    assert eval_render.supported(model) == "predict_normals"
    rb = RayBundle(origins=T(g["m_origins"]).cuda().reshape(4, 4, 3), directions=T(g["m_directions"]).cuda().reshape(4, 4, 3),
                   pixel_area=torch.full((4, 4, 1), 1e-6).cuda(), camera_indices=T(g["m_cams"]).cuda().reshape(4, 4, 1))
    img = model.get_outputs_for_camera_ray_bundle(rb)
    assert img["normals"].shape == (4, 4, 3) and img["pred_normals"].shape == (4, 4, 3)
    _rendered_normals_close(img["normals"].reshape(N, 3), g["m_eval_normals"], "image")
