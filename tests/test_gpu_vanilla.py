"""Vanilla NeRF (BASELINE configs[0], the reference's own CPU-runnable case) on the MI355X kernels against the fixture the
REFERENCE's NeRFModel wrote (tests/golden/vanilla.npz: models/vanilla_nerf.py:139-217 driven with the oracle's seeded
parameters): frequency encoding, the 8 x 256 skip MLP on 128 x 128 blocks of f32 MFMA, softplus / sigmoid heads, uniform +
PDF(include_original) samplers with per-edge jitter, white-background compositing, median depth, both MSE losses, gradients
of all 48 parameter tensors."""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc
from oracle import vanilla_oracle as van

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _model(params):
    from nerfstudio_amd.vanilla_nerf import NeRFModel, VanillaModelConfig

    model = NeRFModel(VanillaModelConfig()).cuda()
    sd = model.state_dict()
    assert set(params) <= set(sd), sorted(set(params) - set(sd))[:5]   # the reference's state-dict names
    assert {k for k in sd if k.startswith("field_")} == set(params)
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=False)
    return model


def _params():
    cfg = van.VanillaCfg()
    params = {}
    params.update(van.init_field_params(cfg, 92, "field_coarse."))
    params.update(van.init_field_params(cfg, 93, "field_fine."))
    return cfg, params


def test_nerf_encoding_golden(golden):
    from nerfstudio_amd.field_components.encodings import NeRFEncoding

    g = golden("vanilla")
    x = T(g["enc_x"]).cuda()
    for name, args in (("enc_pos", (10, 0.0, 8.0, True)), ("enc_dir", (4, 0.0, 4.0, True)), ("enc_plain", (6, 0.0, 5.0, False))):
        enc = NeRFEncoding(3, *args)
        out = enc(x)
        assert out.shape == g[name].shape == (x.shape[0], enc.get_out_dim())
        # sin of arguments up to 2 pi 2^8 |x|: one ulp of the ARGUMENT is ~1e-4 there, libm-to-libm differences ~1e-7
        np.testing.assert_allclose(out.cpu().numpy(), g[name], rtol=0, atol=2e-6, err_msg=name)
    # batch shape preserved; rays + bin edges give the encoding of the sample midpoints
    assert NeRFEncoding(3, 4, 0.0, 4.0, True)(torch.zeros(5, 7, 3).cuda()).shape == (5, 7, 27)
    # points that carry gradient (pose corrections behind the predicted-normals head) take the same arithmetic through torch:
    # same values, and a gradient; the kernel entry itself still refuses them
    enc = NeRFEncoding(3, 4, 0.0, 4.0, True)
    xg = x.clone().requires_grad_(True)
    out_g = enc(xg)
    np.testing.assert_allclose(out_g.detach().cpu().numpy(), enc(x).cpu().numpy(), rtol=0, atol=2e-6)
    out_g[:, :24].sum().backward()
    freqs = 2 ** torch.linspace(0.0, 4.0, 4).cuda()
    arg = (2 * torch.pi * x)[..., None] * freqs
    want = (2 * torch.pi * freqs * (torch.cos(arg) + torch.cos(arg + torch.pi / 2))).sum(-1)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-3)
    from nerfstudio_amd import functional as Fn

    with pytest.raises(RuntimeError, match="no backward"):
        Fn.nerf_encode(Fn.PointSpec(positions=x.clone().requires_grad_(True)), 4, 0.0, 4.0)


def test_wide_mlp_with_skip_connection():
    """MLP(8 x 256, skip (4,), ReLU out) — vanilla-nerf's base network (mlp.py:143-179) — in 128 x 128 blocks; ragged widths
    200 / 319 exercise the unaligned block edges."""
    from nerfstudio_amd.field_components.mlp import MLP

    rs = np.random.RandomState(7)
    for in_dim, layers, width, out_dim, skips, out_act, M in ((63, 8, 256, 256, (4,), torch.nn.ReLU(), 300),
                                                              (10, 3, 200, 5, (1,), None, 77),
                                                              (130, 2, 129, 131, None, torch.nn.Sigmoid(), 33)):
        mlp = MLP(in_dim=in_dim, num_layers=layers, layer_width=width, out_dim=out_dim, skip_connections=skips,
                  out_activation=out_act).cuda()
        params = {f"layers.{i}.{k}": getattr(l, k).detach().cpu().clone().requires_grad_(True)
                  for i, l in enumerate(mlp.layers) for k in ("weight", "bias")}
        x = T(rs.standard_normal((M, in_dim)).astype(np.float32))
        xr = x.clone().requires_grad_(True)
        if isinstance(out_act, torch.nn.Sigmoid):
            ref = torch.sigmoid(van.mlp_skip_forward(xr, params, "", skips or ()))
        else:
            ref = van.mlp_skip_forward(xr, params, "", skips or (), out_activation="relu" if out_act is not None else None)
        xg = x.cuda().requires_grad_(True)
        out = mlp(xg)
        assert out.shape == (M, out_dim)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=5e-6)
        gy = T(rs.standard_normal((M, out_dim)).astype(np.float32))
        (ref * gy).sum().backward()
        (out * gy.cuda()).sum().backward()

        def gclose(a, b, what):
            a, b = a.detach().cpu().numpy(), b.detach().numpy()
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30), what

        gclose(xg.grad, xr.grad, "dx")
        for i, l in enumerate(mlp.layers):
            gclose(l.weight.grad, params[f"layers.{i}.weight"].grad, f"dW{i}")
            gclose(l.bias.grad, params[f"layers.{i}.bias"].grad, f"db{i}")


def test_softplus_density_head():
    from nerfstudio_amd.field_components.field_heads import DensityFieldHead, FieldHeadNames, RGBFieldHead

    head = DensityFieldHead(in_dim=40).cuda()
    assert head.field_head_name == FieldHeadNames.DENSITY and set(head.state_dict()) == {"net.weight", "net.bias"}
    x = (torch.randn(500, 40) * 8).cuda().requires_grad_(True)   # pre-activations beyond +-20: both softplus branches
    y = head(x)
    w, b = head.net.weight.detach().cpu().requires_grad_(True), head.net.bias.detach().cpu().requires_grad_(True)
    xr = x.detach().cpu().requires_grad_(True)
    ref = torch.nn.functional.softplus(xr @ w.t() + b)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)
    gy = torch.randn(500, 1)
    (ref * gy).sum().backward()
    (y * gy.cuda()).sum().backward()
    np.testing.assert_allclose(head.net.weight.grad.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4 * float(w.grad.abs().max()))
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-6)
    assert RGBFieldHead(in_dim=8).cuda()(torch.zeros(3, 5, 8).cuda()).shape == (3, 5, 3)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_vanilla_nerf_matches_reference_fixture(golden, mode):
    from nerfstudio_amd.cameras.rays import RayBundle

    g = golden("vanilla")
    cfg, params = _params()
    np.testing.assert_allclose(np.array([float(v.double().sum()) for v in params.values()]), g["param_checksum"], rtol=1e-12)
    model = _model(params)
    training = mode == "train"
    model.train(training)
    n = g["origins"].shape[0]
    rb = RayBundle(origins=T(g["origins"]).cuda(), directions=T(g["directions"]).cuda(), pixel_area=torch.ones(n, 1).cuda(),
                   nears=torch.full((n, 1), cfg.near_plane).cuda(), fars=torch.full((n, 1), cfg.far_plane).cuda())
    jit = (T(g["j0"]).cuda(), T(g["j1"]).cuda()) if training else None
    with torch.set_grad_enabled(training):
        out = model.get_outputs(rb, jitters=jit)
    pre = mode + "_"
    c = lambda t: t.detach().cpu().numpy()  # noqa: E731
    np.testing.assert_allclose(c(out["weights_coarse"][..., 0]), g[pre + "weights_coarse"], rtol=5e-4, atol=2e-7)
    np.testing.assert_allclose(c(out["weights_fine"][..., 0]), g[pre + "weights_fine"], rtol=2e-3, atol=1e-6)
    for k in ("rgb_coarse", "rgb_fine"):
        np.testing.assert_allclose(c(out[k]), g[pre + k], rtol=0, atol=1e-5, err_msg=k)   # north_star: 1e-4
    for k in ("accumulation_coarse", "accumulation_fine"):
        np.testing.assert_allclose(c(out[k]), g[pre + k], rtol=0, atol=5e-6, err_msg=k)
    for k in ("depth_coarse", "depth_fine"):
        np.testing.assert_allclose(c(out[k]), g[pre + k], rtol=2e-5, err_msg=k)
    if not training:
        return
    loss_dict = model.get_loss_dict(out, {"image": T(g["target"]).cuda()})
    assert set(loss_dict) == {"rgb_loss_coarse", "rgb_loss_fine"}
    loss = sum(loss_dict.values())
    np.testing.assert_allclose(float(loss), float(g["train_loss"]), rtol=2e-5)
    loss.backward()
    grads = dict(model.named_parameters())
    for name in params:
        gr = grads[name].grad.reshape(-1).cpu()
        idx = T(g[f"train_gidx_{name}"])
        stat = g[f"train_gstat_{name}"]
        ref_vals = g[f"train_gval_{name}"]
        atol = 3e-2 * max(float(np.abs(ref_vals).max()), float(stat[0]) / np.sqrt(gr.numel())) + 1e-12
        np.testing.assert_allclose(gr[idx].numpy(), ref_vals, rtol=0, atol=atol, err_msg=name)
        np.testing.assert_allclose(float(gr.double().norm()), float(stat[0]), rtol=2e-3, err_msg=name)


def test_vanilla_nerf_trains():
    """A few Adam steps on one batch lower both losses (the whole graph is differentiable end to end)."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.vanilla_nerf import NeRFModel, VanillaModelConfig

    torch.manual_seed(0)
    model = NeRFModel(VanillaModelConfig(num_coarse_samples=16, num_importance_samples=32)).cuda().train()
    assert set(model.get_param_groups()) == {"fields"} and len(model.get_param_groups()["fields"]) == 48
    n = 256
    o = torch.zeros(n, 3).cuda()
    o[:, 2] = 4.0
    d = torch.nn.functional.normalize(torch.randn(n, 3) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1).cuda()
    target = torch.rand(n, 3).cuda() * 0.5
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    first = None
    for _ in range(30):
        rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(n, 1).cuda())
        out = model(rb)
        loss = sum(model.get_loss_dict(out, {"image": target}).values())
        first = float(loss) if first is None else first
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert float(loss) < 0.7 * first, (first, float(loss))
