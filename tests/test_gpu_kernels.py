"""GPU parity tests, kernel by kernel: the HIP path (through the C ABI) against
 (i) the golden fixtures generated from the reference itself (tests/golden/*.npz) and
 (ii) the CPU oracle (oracle/nerfacto_oracle.py) on fresh seeded inputs.

Tolerances: integer results (sample indices, median index) and pure-IEEE stages (piecewise bins, PDF bins given
identical weights) are BIT-EXACT; hash features <= 1e-6; everything downstream of an MLP / exp within the float
tolerances written at each assert (north_star: RGB 1e-4 L-inf)."""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.cuda()


def close(a, b, atol=1e-6, rtol=1e-5, msg=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=msg)


def exact(a, b, msg=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_array_equal(a, b, err_msg=msg)


def gclose(a, b, rel=1e-4, msg="", skip_ref_nan=False):
    """gradient comparison: absolute tolerance scaled by the largest reference entry.
    skip_ref_nan: the reference's position gradient is NaN at the exact origin (0 * inf inside the unselected branch of
    SceneContraction's torch.where, spatial_distortions.py:66-69); the kernel returns the finite identity-branch
    gradient there, so rows where the REFERENCE is NaN are excluded."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    if skip_ref_nan:
        keep = np.isfinite(b).all(axis=-1)
        assert keep.sum() >= b.shape[0] - 2 and np.isfinite(a).all()
        a, b = a[keep], b[keep]
    tol = rel * max(1e-12, float(np.abs(b).max()))
    close(a, b, atol=tol, rtol=1e-3, msg=msg)


def gclose_e2e(a, b, rel=5e-4, msg=""):
    """End-to-end gradient comparison. Through the whole pipeline the forward differs from the reference at the 1e-6
    level (MFMA vs BLAS summation order, device expf), which flips the ReLU mask of the occasional hidden unit whose
    pre-activation is ~0 (expected ~0.2 flips per 768-sample batch); a flip changes the <=256 table entries / one
    weight row of ONE sample by a finite amount (observed on this fixture: one sample, 84 of 32768 table entries,
    relative L2 2.9e-3; one hidden unit = one 32-entry row of base W0). So: relative L2 error <= 1e-2 and at most
    max(0.5 % of the entries, two rows) outside the elementwise tolerance. (The per-kernel gradient tests above, fed identical inputs, stay elementwise.)"""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    l2 = float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))
    tol = rel * max(1e-12, float(np.abs(b).max()))
    bad = int((np.abs(a - b) > tol + 1e-3 * np.abs(b)).sum())
    allowed = max(int(5e-3 * a.size), 2 * (a.shape[-1] if a.ndim > 1 else 1))
    assert l2 <= 1e-2 and bad <= allowed, f"{msg}: relative L2 {l2:.2e}, {bad}/{a.size} entries outside tolerance"


@pytest.fixture(scope="module")
def F():
    from nerfstudio_amd import _native, functional

    _native.load()
    info = _native.device_info()
    assert info["wavefront_size"] == 64 and info["arch"].startswith("gfx950"), info
    return functional


def small_cfg(main_log2, prop_log2, num_images):
    return orc.NerfactoCfg(
        main_grid=orc.HashGridCfg(16, 16, 2048, int(main_log2)),
        prop_grids=(orc.HashGridCfg(5, 16, 128, int(prop_log2)), orc.HashGridCfg(5, 16, 256, int(prop_log2))),
        num_images=int(num_images),
    )


# ---------------------------------------------------------------- MFMA layout probe ---------------------------------
def test_mfma_lane_layout(F):
    """A[16,4] @ B[4,16] through one v_mfma_f32_16x16x4_f32 with the lane mapping field_mlp.hip assumes
    (asymmetric operands: a swapped row/col mapping cannot pass)."""
    from nerfstudio_amd import _native as N

    rs = np.random.RandomState(0)
    A = rs.standard_normal((16, 4)).astype(np.float32)
    B = rs.standard_normal((4, 16)).astype(np.float32)
    out = torch.empty((16, 16), device="cuda")
    a, b = dev(A), dev(B)
    N.check(N.load().nsamd_probe_mfma16(N.ptr(a), N.ptr(b), N.ptr(out), N.stream()), "probe")
    ref = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    close(out, ref, atol=1e-6)


def test_mfma_bf16_lane_layout(F):
    """A[16,32] @ B[32,16] through one v_mfma_f32_16x16x32_bf16 with the lane mapping of the three-way-split forward: lane
    (j, g) holds k = 8g .. 8g + 7 of row / column j, the result rows 4g .. 4g + 3 of column j (bf16-representable inputs:
    small integers, so the product is exact)."""
    from nerfstudio_amd import _native as N

    rs = np.random.RandomState(1)
    A = rs.randint(-8, 9, size=(16, 32)).astype(np.float32)
    B = rs.randint(-8, 9, size=(32, 16)).astype(np.float32)
    out = torch.empty((16, 16), device="cuda")
    a, b = dev(A), dev(B)
    N.check(N.load().nsamd_probe_mfma_bf16(N.ptr(a), N.ptr(b), N.ptr(out), N.stream()), "probe")
    exact(out, A @ B, "bf16 MFMA lane layout")


# ---------------------------------------------------------------- hash grid -----------------------------------------
def test_hashgrid_golden(F, golden):
    from nerfstudio_amd.field_components.encodings import HashEncoding

    g = golden("hashgrid")
    L, lo, hi, log2T, _ = [int(v) for v in g["cfg"]]
    enc = HashEncoding(num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=log2T).cuda()
    exact(enc.scalings, g["scalings"], "host-evaluated level scalings differ from the reference's")
    with torch.no_grad():
        enc.hash_table.copy_(dev(g["table"]))
    x = dev(g["x"]).requires_grad_(True)
    out = enc(x)
    assert out.shape == (g["x"].shape[0], 2 * L)
    close(out, g["out"], atol=1e-6, rtol=0, msg="hash features vs reference")
    # same IEEE op sequence as the torch path -> expect bit equality with the oracle on this machine
    o = orc.hashgrid_encode(T(g["x"]), T(g["table"]), orc.hash_level_scalings(L, lo, hi), 2**log2T)
    exact(out, o, "hash features are not bit-identical to the oracle")
    (out * dev(g["gout"])).sum().backward()
    gclose(enc.hash_table.grad, g["dtable"], 1e-5, "dtable")
    gclose(x.grad, g["dx"], 1e-4, "dx")


def test_hashgrid_main_table_bit_exact_vs_oracle(F):
    """The nerfacto MAIN grid (L = 16, 16 .. 2048, T = 2^19 — the table size that selects the one-level-per-thread forward with
    its lane-pair gathers, csrc/hashgrid.hip) on an odd number of random points, points on lattice planes and on the box's
    faces: features bit-identical to the oracle's torch expression (the kernel's blend is the reference's operation order,
    encodings.py:417-458)."""
    from nerfstudio_amd.field_components.encodings import HashEncoding

    L, lo, hi, log2T = 16, 16, 2048, 19
    enc = HashEncoding(num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=log2T).cuda()
    rs = np.random.RandomState(11)
    table = (rs.standard_normal((L * 2**log2T, 2)) * 0.1).astype(np.float32)
    with torch.no_grad():
        enc.hash_table.copy_(dev(table))
    x = rs.uniform(0, 1, (30001, 3)).astype(np.float32)
    x[:64] = np.round(x[:64] * 16) / 16          # on lattice planes of the coarsest level (lo == hi there)
    x[64:96] = rs.randint(0, 2, (32, 3))          # the box's corners and faces
    x[96:128, 0] = 1.0
    out = enc(dev(x))
    o = orc.hashgrid_encode(T(x), T(table), orc.hash_level_scalings(L, lo, hi), 2**log2T)
    exact(out, o, "main-table hash features are not bit-identical to the oracle")


def test_hashgrid_shapes_and_edges(F):
    """Reference test contract (tests/field_components/test_encodings.py:143-169): shape (10,16) for L=8,F=2; plus
    empty input and batch-shape preservation."""
    from nerfstudio_amd.field_components.encodings import HashEncoding, SHEncoding

    enc = HashEncoding(num_levels=8, features_per_level=2, log2_hashmap_size=5, min_res=2, max_res=4).cuda()
    assert enc.get_out_dim() == 16
    assert enc(torch.rand((10, 3), device="cuda")).shape == (10, 16)
    assert enc(torch.rand((4, 5, 3), device="cuda")).shape == (4, 5, 16)
    assert enc(torch.rand((0, 3), device="cuda")).shape == (0, 16)
    sh = SHEncoding(levels=4).cuda()
    assert sh(torch.rand((10, 3), device="cuda")).shape == (10, 16)
    d = torch.nn.functional.normalize(torch.randn((1000, 3)), dim=-1)
    close(sh(d.cuda()), orc.sh_levels4(d), atol=1e-6)


def test_contraction_kernel(F, golden):
    g = golden("kat")
    exact(F.contract_linf(dev(g["contract_in"])), g["contract_out"])
    x = torch.randn((5000, 3)) * 3
    exact(F.contract_linf(x.cuda()), orc.contract_linf(x))


# ---------------------------------------------------------------- fields --------------------------------------------
def _hip_model(cfg, params, training=True):
    """nerfstudio_amd NerfactoModel with the oracle's parameter dict loaded (state-dict names are the reference's)."""
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    mc = NerfactoModelConfig(
        log2_hashmap_size=cfg.main_grid.log2_hashmap_size,
        proposal_net_args_list=[
            {"hidden_dim": cfg.prop_hidden_dim, "log2_hashmap_size": g.log2_hashmap_size, "num_levels": g.num_levels,
             "max_res": g.max_res, "use_linear": False} for g in cfg.prop_grids],
        average_init_density=cfg.average_init_density,
        appearance_embed_dim=cfg.appearance_embed_dim,
        use_appearance_embedding=cfg.appearance_embed_dim > 0,
    )
    model = NerfactoModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
    sd = {k: v.detach().clone() for k, v in params.items()}
    for i in range(len(cfg.prop_grids)):
        sd[f"proposal_networks.{i}.mlp_base.0.hash_table"] = sd[f"proposal_networks.{i}.encoding.hash_table"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(any(s in m for s in ("aabb", "max_res", "num_levels", "log2_hashmap_size")) for m in missing), missing
    model = model.cuda()
    model.train(training)
    return model


def test_proposal_density_golden(F, golden):
    g = golden("fields")
    cfg = small_cfg(g["cfg_main_log2"], g["cfg_prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    model = _hip_model(cfg, params)
    for i, net in enumerate(model.proposal_networks):
        net.zero_grad()
        pos = dev(g["positions"]).requires_grad_(True)
        dens = net.density_fn(pos)
        assert dens.shape == (pos.shape[0], 1)
        close(dens[:, 0], g[f"prop{i}_density"], atol=1e-6, rtol=5e-5, msg=f"prop{i} density")
        (dens[:, 0] * dev(g[f"prop{i}_g"])).sum().backward()
        gclose(net.encoding.hash_table.grad, g[f"prop{i}_dtable"], 1e-4, "dtable")
        for j in range(2):
            gclose(net.mlp_base[1].layers[j].weight.grad, g[f"prop{i}_dW{j}"], 1e-4, f"dW{j}")
            gclose(net.mlp_base[1].layers[j].bias.grad, g[f"prop{i}_db{j}"], 1e-4, f"db{j}")
        gclose(pos.grad, g[f"prop{i}_dpos"], 2e-3, "dpos", skip_ref_nan=True)


@pytest.mark.parametrize("contract", [True, False])
def test_density_field_use_linear(F, contract):
    """HashMLPDensityField(use_linear=True) (density_fields.py:81-84, 107-109): hash features -> one dense layer ->
    trunc_exp, against the oracle's pieces under torch autograd; state-dict names as the reference's."""
    from nerfstudio_amd.field_components.spatial_distortions import SceneContraction
    from nerfstudio_amd.fields.density_fields import HashMLPDensityField
    from oracle.nerfacto_oracle import hash_level_scalings, hashgrid_encode, normalise_positions, trunc_exp

    torch.manual_seed(3)
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    fld = HashMLPDensityField(aabb, use_linear=True, spatial_distortion=SceneContraction(order=float("inf")) if contract else None,
                              num_levels=5, max_res=128, base_res=16, log2_hashmap_size=12, average_init_density=0.7).cuda()
    assert set(fld.state_dict()) >= {"linear.weight", "linear.bias", "encoding.hash_table"} and not hasattr(fld, "mlp_base")
    with torch.no_grad():
        fld.encoding.hash_table.uniform_(-0.5, 0.5)
        fld.linear.weight.normal_(0, 0.5)
    pos = ((torch.rand(300, 3) * 2 - 1) * (3.0 if contract else 1.2)).cuda().requires_grad_(True)
    gout = torch.randn(300).cuda()
    dens = fld.density_fn(pos)
    assert dens.shape == (300, 1)
    (dens[:, 0] * gout).sum().backward()
    # oracle
    table = fld.encoding.hash_table.detach().cpu().requires_grad_(True)
    W, b = fld.linear.weight.detach().cpu().requires_grad_(True), fld.linear.bias.detach().cpu().requires_grad_(True)
    p, sel = normalise_positions(pos.detach().cpu(), contract, aabb)
    enc = hashgrid_encode(p, table, hash_level_scalings(5, 16, 128), 1 << 12)
    ref = 0.7 * trunc_exp(enc @ W.t() + b)[:, 0] * sel
    (ref * gout.cpu()).sum().backward()
    assert 0.2 < float(sel.float().mean()) < 1.0 or contract
    close(dens[:, 0], ref, atol=1e-6, rtol=2e-5, msg="density")
    gclose(fld.encoding.hash_table.grad, table.grad, 1e-4, "dtable")
    gclose(fld.linear.weight.grad, W.grad, 1e-4, "dW")
    gclose(fld.linear.bias.grad, b.grad, 1e-4, "db")


def test_nerfacto_field_golden(F, golden):
    from nerfstudio_amd.cameras.rays import Frustums, RaySamples
    from nerfstudio_amd.field_components.field_heads import FieldHeadNames

    g = golden("fields")
    cfg = small_cfg(g["cfg_main_log2"], g["cfg_prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    M = g["positions"].shape[0]
    R, S = M // 4, 4
    for mode in ("train", "eval"):
        model = _hip_model(cfg, params, training=(mode == "train"))
        fld = model.field
        o = dev(g["positions"]).reshape(R, S, 3).requires_grad_(mode == "train")
        fr = Frustums(origins=o, directions=dev(g["directions"]).reshape(R, S, 3), starts=torch.zeros(R, S, 1).cuda(),
                      ends=torch.zeros(R, S, 1).cuda(), pixel_area=torch.ones(R, S, 1).cuda())
        fo = fld(RaySamples(frustums=fr, camera_indices=dev(g["cams"]).reshape(R, S, 1)))
        dens, rgb = fo[FieldHeadNames.DENSITY], fo[FieldHeadNames.RGB]
        assert dens.shape == (R, S, 1) and rgb.shape == (R, S, 3)
        close(dens.reshape(M), g[f"main_{mode}_density"], atol=1e-6, rtol=1e-4, msg=f"{mode} density")
        close(rgb.reshape(M, 3), g[f"main_{mode}_rgb"], atol=1e-5, rtol=0, msg=f"{mode} rgb (north_star: 1e-4)")
        if mode == "train":
            ((dens.reshape(M) * dev(g["main_g_density"])).sum() + (rgb.reshape(M, 3) * dev(g["main_g_rgb"])).sum()).backward()
            gclose(fld.mlp_base.encoding.hash_table.grad, g["main_dtable"], 1e-4, "dtable")
            gclose(fld.embedding_appearance.embedding.weight.grad, g["main_demb"], 1e-4, "demb")
            for j in range(2):
                gclose(fld.mlp_base.mlp.layers[j].weight.grad, g[f"main_base_dW{j}"], 1e-4, f"base dW{j}")
                gclose(fld.mlp_base.mlp.layers[j].bias.grad, g[f"main_base_db{j}"], 1e-4, f"base db{j}")
            for j in range(3):
                gclose(fld.mlp_head.layers[j].weight.grad, g[f"main_head_dW{j}"], 1e-4, f"head dW{j}")
                gclose(fld.mlp_head.layers[j].bias.grad, g[f"main_head_db{j}"], 1e-4, f"head db{j}")
            gclose(o.grad.reshape(M, 3), g["main_dpos"], 5e-3, "dpos", skip_ref_nan=True)


def test_field_mlp_ragged_sizes(F):
    """M not a multiple of the 16-point MFMA tile, M = 1, and M = 0 (edge cases)."""
    cfg = small_cfg(8, 6, 3)
    params = orc.init_params(cfg, seed=1, table_std=0.3)
    model = _hip_model(cfg, params)
    rs = np.random.RandomState(3)
    for M in (1, 15, 17, 33, 100):
        pos = torch.from_numpy((rs.standard_normal((M, 3)) * 0.7).astype(np.float32))
        d = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((M, 3)).astype(np.float32)), dim=-1)
        cam = torch.from_numpy(rs.randint(0, 3, (M,)).astype(np.int64))
        for v in params.values():
            v.requires_grad_(True)
            v.grad = None
        od, orgb, _ = orc.nerfacto_field(pos, d, cam, params, cfg, training=True)
        gd = torch.from_numpy(rs.standard_normal((M,)).astype(np.float32))
        gr = torch.from_numpy(rs.standard_normal((M, 3)).astype(np.float32))
        ((od * gd).sum() + (orgb * gr).sum()).backward()
        from nerfstudio_amd.cameras.rays import Frustums, RaySamples
        from nerfstudio_amd.field_components.field_heads import FieldHeadNames

        model.zero_grad()
        fr = Frustums(origins=pos.cuda(), directions=d.cuda(), starts=torch.zeros(M, 1).cuda(),
                      ends=torch.zeros(M, 1).cuda(), pixel_area=torch.ones(M, 1).cuda())
        fo = model.field(RaySamples(frustums=fr, camera_indices=cam.cuda()[:, None]))
        close(fo[FieldHeadNames.DENSITY][:, 0], od, atol=1e-6, rtol=1e-4, msg=f"M={M}")
        close(fo[FieldHeadNames.RGB], orgb, atol=1e-5, rtol=0, msg=f"M={M}")
        ((fo[FieldHeadNames.DENSITY][:, 0] * gd.cuda()).sum() + (fo[FieldHeadNames.RGB] * gr.cuda()).sum()).backward()
        gclose(model.field.mlp_head.layers[0].weight.grad, params["field.mlp_head.layers.0.weight"].grad, 2e-4, f"M={M}")
        gclose(model.field.mlp_base.mlp.layers[0].weight.grad, params["field.mlp_base.model.1.layers.0.weight"].grad, 2e-4)
        gclose(model.field.mlp_base.encoding.hash_table.grad, params["field.mlp_base.model.0.hash_table"].grad, 2e-4)
        gclose(model.field.embedding_appearance.embedding.weight.grad,
               params["field.embedding_appearance.embedding.weight"].grad, 2e-4)
    empty = model.proposal_networks[0].density_fn(torch.zeros((0, 3), device="cuda"))
    assert empty.shape == (0, 1)


@pytest.mark.parametrize("S,appearance", [(48, "cameras"), (16, "cameras"), (32, "const"), (48, "none"), (80, "cameras")])
def test_field_ray_terms_equal_the_plain_kernels(F, S, appearance):
    """nsamd_field_ray_terms + the kernels that start head layer 0 from the per-ray terms (include/nsamd.h, nsamd_field_mlp.
    ray_terms) against the plain kernels on the same inputs: the density is bit-equal (the base MLP is untouched), rgb and
    every gradient — the 48 per-ray columns of head layer 0's weight gradient and the appearance rows come from the per-tile
    sums of dL/d(pre-activation) — agree to fp32 rounding (another summation order of the same products). 37 rays: the
    ray-terms kernel's last 16-ray tile is ragged; S = 48 / 16 / 32 / 80: three, one, two, five tiles per ray."""
    import ctypes as C

    from nerfstudio_amd import _native as N

    lib = N.load()
    rs = np.random.RandomState(S)
    R, ncam = 37, 5
    M = R * S
    g = lambda *shape, s=1.0: torch.from_numpy((rs.standard_normal(shape) * s).astype(np.float32)).cuda()  # noqa: E731
    k0 = {"cameras": 63, "const": 63, "none": 31}[appearance]
    prm = [g(64, 32, s=0.3), g(64, s=0.1), g(16, 64, s=0.2), g(16, s=0.1), g(64, k0, s=0.2), g(64, s=0.1), g(64, 64, s=0.2),
           g(64, s=0.1), g(3, 64, s=0.2), g(3, s=0.1)]
    emb = g(ncam, 32) if appearance == "cameras" else None
    app_const = g(32) if appearance == "const" else None
    cams = torch.from_numpy(rs.randint(0, ncam, (R,)).astype(np.int64)).cuda() if appearance == "cameras" else None
    enc, sel = g(32, M, s=0.5), (torch.from_numpy(rs.uniform(0, 1, (M,)).astype(np.float32)) > 0.1).float().cuda()
    dirs = torch.nn.functional.normalize(g(R, 3), dim=-1).contiguous()
    gd, gr = g(M, s=0.1), g(M, 3)

    def run(with_terms):
        fm = N.FieldMlp(*(N.ptr(p) for p in prm), N.ptr(emb), ncam if emb is not None else 0, 1.0)
        keep = None
        if with_terms:
            terms = torch.full((R, 64), float("nan"), device="cuda")
            xin = torch.full((R, 16 + (0 if appearance == "none" else 32)), float("nan"), device="cuda")
            N.check(lib.nsamd_field_ray_terms(N.ptr(dirs), N.ptr(cams), N.ptr(app_const), R, fm, N.ptr(terms), N.ptr(xin),
                                              N.stream()), "field_ray_terms")
            fm.ray_terms, fm.ray_inputs = N.ptr(terms), N.ptr(xin)
            keep = (terms, xin)
        dens, rgb = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
        N.check(lib.nsamd_field_mlp_fwd(N.ptr(enc), N.ptr(sel), N.ptr(dirs), N.ptr(cams), N.ptr(app_const), S, M, fm, N.ptr(dens),
                                        N.ptr(rgb), N.stream()), "field_mlp_fwd")
        grads = [torch.zeros_like(p) for p in prm]
        gemb = torch.zeros_like(emb) if emb is not None else None
        denc = torch.empty_like(enc)
        ws, ws_n = F.field_bwd_workspace(torch.device("cuda"))
        ws.fill_(float("nan"))  # whatever the kernels read of the scratch they must have written
        N.check(lib.nsamd_field_mlp_bwd(N.ptr(enc), N.ptr(sel), N.ptr(dirs), N.ptr(cams), N.ptr(app_const), S, M, fm, N.ptr(gd),
                                        N.ptr(gr), N.ptr(denc), N.FieldMlpGrads(*(N.ptr(x) for x in grads), N.ptr(gemb)),
                                        N.ptr(ws), ws_n, N.stream()), "field_mlp_bwd")
        torch.cuda.synchronize()
        return dens, rgb, denc, grads, gemb, keep

    d0, rgb0, denc0, g0, e0, _ = run(False)
    d1, rgb1, denc1, g1, e1, (terms, xin) = run(True)
    assert torch.isfinite(terms).all() and torch.isfinite(xin).all()
    exact(d1, d0, "density")
    close(rgb1, rgb0, atol=2e-6, rtol=0, msg="rgb")
    names = ["base_W0", "base_b0", "base_W1", "base_b1", "head_W0", "head_b0", "head_W1", "head_b1", "head_W2", "head_b2"]
    for name, a, b in zip(names + ["denc"], g1 + [denc1], g0 + [denc0]):
        assert torch.isfinite(a).all(), name
        gclose(a, b, 2e-5, name)
    if e0 is not None:
        gclose(e1, e0, 2e-5, "appearance embedding")


# ---------------------------------------------------------------- samplers ------------------------------------------
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_samplers_golden_bit_exact(F, golden, mode):
    g = golden("samplers")
    nears, fars = dev(g["nears"]), dev(g["fars"])
    tr = mode == "train"
    s0, t0 = F.piecewise_bins(nears, fars, 256, dev(g["j0"]) if tr else None)
    exact(s0, g[f"{mode}_l0_s_bins"], "piecewise s_bins")
    exact(t0, g[f"{mode}_l0_t_bins"], "piecewise t_bins")
    w0 = F.weights_from_density(t0, dev(g[f"{mode}_l0_density"]))
    close(w0, g[f"{mode}_l0_weights"], atol=2e-7, rtol=2e-6, msg="weights")
    # identical inputs (the reference's weights) -> bit-exact indices and bins against the ORACLE
    w_ref = T(g[f"{mode}_l0_weights"])
    dbg = {}
    so, to, io = orc.pdf_resample(T(g[f"{mode}_l0_s_bins"]), w_ref, 96, T(g["j1"]) if tr else None, T(g["nears"]), T(g["fars"]),
                                  debug=dbg)
    s1, t1, i1 = F.pdf_resample(s0, dev(w_ref), 96, dev(g["j1"]) if tr else None, nears, fars, return_indices=True)
    exact(i1, io, "PDF sample indices vs oracle")
    exact(s1, so, "PDF s_bins vs oracle")
    exact(t1, to, "PDF t_bins vs oracle")
    # against the REFERENCE's recorded indices: equal, except where the draw u ties with every cdf entry between the two
    # answers to within 2 ulp — the tie the last bit of torch.sum(weights) decides (the reference's own CPU and CUDA kernels
    # differ there; include/nsamd.h, nsamd_pdf_resample). Every flipped index must BE such a tie (north_star: bit-exact
    # sample indices), not merely be one of "at most 4".
    mine, ref = i1.cpu().numpy(), g[f"{mode}_l1_inds"]
    cdf, u = dbg["cdf"].numpy(), dbg["u"].numpy()
    flips = np.argwhere(mine != ref)
    assert len(flips) <= 4, f"{len(flips)} indices differ from the reference"
    for r, c in flips:
        lo, hi = sorted((int(mine[r, c]), int(ref[r, c])))
        gap = np.abs(cdf[r, lo:hi] - u[r, c])
        assert np.all(gap <= 2 * np.spacing(np.float32(u[r, c]))), \
            f"index [{r},{c}]: kernel {mine[r, c]} vs reference {ref[r, c]} is not a cdf tie (|cdf - u| = {gap}, u = {u[r, c]})"
    close(s1, g[f"{mode}_l1_s_bins"], atol=2e-6, rtol=0)
    # annealed resample: pow() is not bit-reproducible across libms, so compare on values
    w1 = T(g[f"{mode}_l1_weights"])
    s2, t2 = F.pdf_resample(dev(g[f"{mode}_l1_s_bins"]), dev(w1), 48, dev(g["j2"]) if tr else None, nears, fars,
                            anneal=float(g["anneal"]))
    close(s2, g[f"{mode}_l2_s_bins"], atol=3e-6, rtol=0)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_sampler_variants_golden_vanilla(F, golden, mode):
    """single_jitter=False (one draw per bin edge) and PDFSampler(include_original=True) against the REFERENCE's own
    vanilla-nerf samplers (tests/golden/vanilla.npz: UniformSampler(64) -> PDFSampler(128, include_original=True) with
    per-edge jitter, ray_samplers.py:104-107, 318-322, 356-357) and bit-exactly against the oracle."""
    from oracle import vanilla_oracle as vo

    g = golden("vanilla")
    tr = mode == "train"
    n = g["origins"].shape[0]
    nears, fars = torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)
    j0, j1 = (T(g["j0"]), T(g["j1"])) if tr else (None, None)
    s0, t0 = F.piecewise_bins(nears.cuda(), fars.cuda(), 64, dev(g["j0"]) if tr else None, spacing=1)
    so, to = vo.uniform_bins(nears, fars, 64, j0)
    exact(s0, so, "uniform s_bins vs oracle")
    exact(t0, to, "uniform t_bins vs oracle")
    close(t0, g[f"{mode}_t_bins_coarse"], atol=1e-6, rtol=0, msg="uniform t_bins vs reference")
    w = T(g[f"{mode}_weights_coarse"])
    s1, t1, i1 = F.pdf_resample(s0, w.cuda(), 128, dev(g["j1"]) if tr else None, nears.cuda(), fars.cuda(), spacing=1,
                                include_original=True, return_indices=True)
    assert s1.shape == (n, 64 + 128 + 2) and i1.shape == (n, 129)
    sm, tm = vo.pdf_resample_with_original(so, w, nears, fars, 128, j1)
    exact(s1, sm, "merged s_bins vs oracle")
    exact(t1, tm, "merged t_bins vs oracle")
    assert bool((s1[:, 1:] >= s1[:, :-1]).all()), "merged edges are sorted"
    # against the REFERENCE's recorded searchsorted indices: every flip must BE a cdf tie (|cdf - u| <= 2 ulp over the entries
    # between the two answers), as for the nerfacto sampler above — not merely be one of "at most 4" 
    dbg = {}
    io = orc.pdf_resample(so, w, 128, j1, nears, fars, uniform=True, debug=dbg)[2]
    exact(i1, io, "PDF sample indices vs oracle")
    mine, ref = i1.cpu().numpy(), g[f"{mode}_pdf_inds"]
    cdf, u = dbg["cdf"].numpy(), dbg["u"].numpy()
    flips = np.argwhere(mine != ref)
    assert len(flips) <= 4, f"{len(flips)} searchsorted indices differ from the reference"
    for r, c in flips:
        lo, hi = sorted((int(mine[r, c]), int(ref[r, c])))
        gap = np.abs(cdf[r, lo:hi] - u[r, c])
        assert np.all(gap <= 2 * np.spacing(np.float32(u[r, c]))), \
            f"index [{r},{c}]: kernel {mine[r, c]} vs reference {ref[r, c]} is not a cdf tie (|cdf - u| = {gap}, u = {u[r, c]})"
    close(s1, g[f"{mode}_s_bins_fine"], atol=2e-6, rtol=0, msg="merged s_bins vs reference")
    close(t1, g[f"{mode}_t_bins_fine"], atol=1e-5, rtol=0, msg="merged t_bins vs reference")


def test_sampler_variant_modules():
    """The mirror classes with the reference's defaults (single_jitter=False, include_original=True): shapes, sortedness,
    and the draw layout (one per edge)."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.model_components.ray_samplers import PDFSampler, UniformSampler

    n = 37
    rb = RayBundle(origins=torch.zeros(n, 3).cuda(), directions=torch.nn.functional.normalize(torch.randn(n, 3), dim=-1).cuda(),
                   pixel_area=torch.ones(n, 1).cuda(), nears=torch.full((n, 1), 2.0).cuda(), fars=torch.full((n, 1), 6.0).cuda())
    us, ps = UniformSampler(num_samples=64), PDFSampler(num_samples=128)
    assert us.single_jitter is False and ps.single_jitter is False and ps.include_original is True
    us.train(), ps.train()
    rs0 = us(rb)
    w = torch.rand(n, 64, 1).cuda()
    rs1 = ps(rb, rs0, w)
    assert rs0.frustums.starts.shape == (n, 64, 1) and rs1.frustums.starts.shape == (n, 193, 1)
    e = torch.cat([rs1.frustums.starts[..., 0], rs1.frustums.ends[:, -1:, 0]], dim=-1)
    assert bool((e[:, 1:] >= e[:, :-1]).all()) and float(e.min()) >= 2.0 - 1e-6 and float(e.max()) <= 6.0 + 1e-6
    # per-edge draws: the stratified bins of two rays with equal near/far differ in more than a common shift
    d = rs0.spacing_starts[0, :, 0] - rs0.spacing_starts[1, :, 0]
    assert float(d.std()) > 1e-4
    with pytest.raises(ValueError, match="one draw per ray"):
        us(rb, jitter=torch.rand(n, 7).cuda())


def test_weights_backward_golden(F, golden):
    g = golden("samplers")
    dens = dev(g["train_l0_density"]).requires_grad_(True)
    w = F.weights_from_density(dev(g["train_l0_t_bins"]), dens)
    (w * dev(g["weights_g"])).sum().backward()
    gclose(dens.grad, g["weights_ddensity"], 1e-5, "d weights / d density")


@pytest.mark.parametrize("anneal", [0.0, 0.01, 0.5, 0.999])
def test_annealed_resample_zero_and_denormal_weights(F, anneal):
    """pow(weights, anneal) (ray_samplers.py:601) inside the resampling kernel runs on the hardware log2 / exp2: the cases
    those instructions do not cover must still be torch.pow's — anneal = 0 (the first training step: pow(0, 0) = 1), exact
    zeros, and DENORMAL weights (1e-38 .. 1e-45: transmittance underflow behind dense matter; under a small exponent
    pow(1e-43, 0.01) = 0.37 shapes the early, nearly uniform proposal sampling)."""
    n, s_prev, s_new = 64, 96, 48
    rs = np.random.RandomState(3)
    w = np.exp(rs.uniform(-110.0, 0.0, (n, s_prev))).astype(np.float32)   # down to 1e-48: zeros and denormals included
    w[rs.uniform(size=w.shape) < 0.2] = 0.0
    assert (w == 0).any() and ((w > 0) & (w < 1.17e-38)).any()
    nears, fars = torch.full((n, 1), 0.05), torch.full((n, 1), 1000.0)
    s0, _ = orc.piecewise_bins(nears, fars, s_prev, None)
    jit = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32))
    wa = torch.pow(torch.from_numpy(w), anneal)                            # the reference's anneal, then its PDF sampler
    so, to, _ = orc.pdf_resample(s0, wa, s_new, jit, nears, fars)
    s1, t1 = F.pdf_resample(s0.cuda(), torch.from_numpy(w).cuda(), s_new, jit.cuda(), nears.cuda(), fars.cuda(), anneal=anneal)
    assert bool(torch.isfinite(s1).all())
    close(s1, so, atol=3e-6, rtol=0, msg=f"annealed resample, anneal = {anneal}")
    close(t1, to, atol=0, rtol=2e-5)


def test_sampler_ragged(F):
    """num_rays not a multiple of the 16-ray workgroup, S not a multiple of anything, 1 ray, 0 rays."""
    rs = np.random.RandomState(5)
    for n, s_prev, s_new in ((1, 7, 5), (17, 33, 9), (50, 256, 96), (3, 96, 48)):
        nears = torch.full((n, 1), 0.05)
        fars = torch.full((n, 1), 1000.0)
        jit = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32))
        so, to = orc.piecewise_bins(nears, fars, s_prev, jit)
        s0, t0 = F.piecewise_bins(nears.cuda(), fars.cuda(), s_prev, jit.cuda())
        exact(s0, so)
        exact(t0, to)
        dens = torch.from_numpy(np.exp(rs.standard_normal((n, s_prev)) * 2).astype(np.float32))
        wo = orc.weights_from_density(to, dens)
        w = F.weights_from_density(t0, dens.cuda())
        close(w, wo, atol=2e-7, rtol=2e-6)
        j2 = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32))
        s1o, t1o, i1o = orc.pdf_resample(so, wo, s_new, j2, nears, fars)
        s1, t1, i1 = F.pdf_resample(s0, wo.cuda(), s_new, j2.cuda(), nears.cuda(), fars.cuda(), return_indices=True)
        exact(i1, i1o)
        exact(s1, s1o)
        exact(t1, t1o)
    s, t = F.piecewise_bins(torch.zeros((0, 1)).cuda(), torch.zeros((0, 1)).cuda(), 8, None)
    assert s.shape == (0, 9)


# ---------------------------------------------------------------- compositing / losses / raygen --------------------
def test_render_golden(F, golden):
    g = golden("render")
    t_bins = dev(g["t_bins"])
    dens = dev(g["density"]).requires_grad_(True)
    rgb = dev(g["rgb"]).requires_grad_(True)
    w = F.weights_from_density(t_bins, dens)
    close(w, g["weights"], atol=2e-7, rtol=2e-6)
    for bg in ("last_sample", "white", "black", "random"):
        close(F.composite(rgb, w, None, bg, expected_depth=False)[0], g[f"rgb_train_{bg}"], atol=1e-6, msg=bg)
    out, acc, dexp, dmed = F.composite_eval(dev(g["rgb_eval_in"]), w.detach(), t_bins, "last_sample")
    close(out, g["rgb_eval_last_sample"], atol=1e-6)
    close(acc[:, None], g["accumulation"], atol=1e-6)
    dm, idx = F.depth_median(dev(g["weights"]), t_bins, return_index=True)
    exact(idx[:, None], g["depth_median_idx"], "median sample index (given the reference's weights)")
    close(dm[:, None], g["depth_median"], atol=0, rtol=1e-7)
    close(dexp[:, None], g["depth_expected"], rtol=2e-5)
    comp, acc2, de = F.composite(rgb, w, t_bins, "last_sample", expected_depth=True)
    ((comp * dev(g["g_rgb"])).sum() + (acc2[:, None] * dev(g["g_acc"])).sum() + (de[:, None] * dev(g["g_dep"])).sum()).backward()
    gclose(dens.grad, g["d_density"], 1e-4, "d density")
    gclose(rgb.grad, g["d_rgb"], 1e-5, "d rgb")


def test_renderer_modules_reference_contract(F):
    """tests/model_components/test_renderers.py:12-83 bounds, through the module API (shapes with trailing 1)."""
    from nerfstudio_amd.cameras.rays import Frustums, RaySamples
    from nerfstudio_amd.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer

    n, s = 10, 20
    weights = torch.ones((n, s, 1), device="cuda") / s
    rgb = torch.ones((n, s, 3), device="cuda")
    r = RGBRenderer("black")
    out = r(rgb=rgb, weights=weights)
    assert out.shape == (n, 3) and float(out.max()) > 0.9
    r0 = RGBRenderer("black")(rgb=torch.zeros_like(rgb), weights=weights)
    assert float(r0.abs().max()) == pytest.approx(0)
    acc = AccumulationRenderer()(weights)
    assert acc.shape == (n, 1) and float(acc.max()) > 0.9
    t = torch.linspace(0.0, 1.0, s + 1, device="cuda")[None].expand(n, s + 1).contiguous()
    fr = Frustums(origins=torch.zeros(n, s, 3).cuda(), directions=torch.ones(n, s, 3).cuda(), starts=t[:, :-1, None],
                  ends=t[:, 1:, None], pixel_area=torch.ones(n, s, 1).cuda())
    rsamp = RaySamples(frustums=fr, deltas=(t[:, 1:] - t[:, :-1])[..., None])
    for method in ("median", "expected"):
        d = DepthRenderer(method)(weights=weights, ray_samples=rsamp)
        assert d.shape == (n, 1) and float(d.min()) > 0


def test_modules_take_reference_style_samples_without_pack(F):
    """The reference's RaySamples carries no `pack` (this package's dense per-ray layout): the mirror modules must give
    the same results from the reference's fields alone — materialised frustum centres, contiguous frustum bins — as from
    the pack. Sampler output with its pack vs the same samples rebuilt from their public fields;
    `combine_rgb` against the plain sums of renderers.py:72-119."""
    from nerfstudio_amd.cameras.rays import RayBundle, RaySamples, pack_of
    from nerfstudio_amd.model_components.losses import distortion_loss, interlevel_loss
    from nerfstudio_amd.model_components.ray_samplers import PDFSampler, UniformLinDispPiecewiseSampler
    from nerfstudio_amd.model_components.renderers import DepthRenderer, RGBRenderer
    from nerfstudio_amd.model_components.scene_colliders import NearFarCollider

    cfg = small_cfg(10, 8, 5)
    model = _hip_model(cfg, orc.init_params(cfg, seed=3, table_std=0.5))
    n = 37
    o, d, cam, _ = orc.synthetic_rays(n, cfg.num_images, seed=8)
    rb = NearFarCollider(0.05, 1000.0)(RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                                                 camera_indices=cam.cuda()[:, None]))
    sampler = UniformLinDispPiecewiseSampler(num_samples=48, single_jitter=True).cuda().train()
    jit = torch.rand(n, 1, device="cuda")
    rs = sampler(ray_bundle=rb, jitter=jit)
    assert pack_of(rs) is not None
    bare = RaySamples(frustums=rs.frustums, camera_indices=rs.camera_indices, deltas=rs.deltas, spacing_starts=rs.spacing_starts,
                      spacing_ends=rs.spacing_ends, spacing_to_euclidean_fn=rs.spacing_to_euclidean_fn)
    assert pack_of(bare) is None and pack_of(rs[:5]) is None and tuple(rs[:5].shape) == (5, 48)
    with torch.no_grad():
        f0, f1 = model.field(rs), model.field(bare)
        p0, p1 = model.proposal_networks[0].get_density(rs)[0], model.proposal_networks[0].get_density(bare)[0]
    for k in f0:
        close(f0[k], f1[k], atol=1e-6, rtol=1e-6, msg=str(k))
    close(p0, p1, atol=1e-7, rtol=1e-6)
    w = rs.get_weights(p0)
    close(w, bare.get_weights(p1), atol=1e-7, rtol=1e-6)
    for method in ("median", "expected"):
        exact(DepthRenderer(method)(weights=w, ray_samples=rs), DepthRenderer(method)(weights=w, ray_samples=bare))
    pdf = PDFSampler(num_samples=16, include_original=False, single_jitter=True).cuda().train()
    j2 = torch.rand(n, 1, device="cuda")
    r0, r1 = pdf(ray_bundle=rb, ray_samples=rs, weights=w, jitter=j2), pdf(ray_bundle=rb, ray_samples=bare, weights=w, jitter=j2)
    exact(pack_of(r0).t_bins, pack_of(r1).t_bins)
    w2 = r0.get_weights(torch.rand(n, 16, 1, device="cuda"))
    exact(interlevel_loss([w, w2], [rs, r0]), interlevel_loss([w, w2], [bare, r0]))
    exact(distortion_loss([w], [rs]), distortion_loss([w], [bare]))
    # combine_rgb (classmethod, training-mode semantics in any module mode)
    rgb = torch.rand(n, 48, 3, device="cuda")
    acc = w.sum(dim=-2)
    plain = (w * rgb).sum(dim=-2)
    close(RGBRenderer.combine_rgb(rgb, w, background_color="random"), plain, atol=2e-6)
    close(RGBRenderer.combine_rgb(rgb, w, background_color="white"), plain + (1 - acc), atol=2e-6)
    close(RGBRenderer.combine_rgb(rgb, w, background_color="last_sample"), plain + rgb[:, -1] * (1 - acc), atol=2e-6)


@pytest.mark.parametrize("mode", ["SO3xR3", "SE3"])
def test_camera_optimizer_gradients_golden(F, golden, mode):
    """SURVEY.md §8 a3 / VERDICT r01 item 7: nerfacto's default camera optimiser makes origins / directions functions of
    `pose_adjustment`; the loss gradient reaches it through the sample positions of all three levels (hash encoding,
    selector, contraction). Fixture from the reference's own CameraOptimizer (tests/golden/camera_opt.npz). Both drivers:
    the module path (autograd through dL/dposition) and the explicit runner (nsamd_hashgrid_encode_bwd_rays: per-ray
    reduction on the device) must give the reference's dL/d(origins, directions) and dL/dpose_adjustment. Tolerances
    are relative L2 (far rays inherit the conditioning of t ~ 1000, see the CPU oracle test of the same fixture)."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig
    from nerfstudio_amd.train_step import NerfactoTrainStep

    g = golden("camera_opt")
    cfg = small_cfg(g["main_log2"], g["prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    mc = NerfactoModelConfig(
        log2_hashmap_size=cfg.main_grid.log2_hashmap_size,
        proposal_net_args_list=[{"hidden_dim": cfg.prop_hidden_dim, "log2_hashmap_size": pg.log2_hashmap_size,
                                 "num_levels": pg.num_levels, "max_res": pg.max_res, "use_linear": False} for pg in cfg.prop_grids],
        average_init_density=cfg.average_init_density, camera_optimizer=CameraOptimizerConfig(mode=mode))

    def build():
        model = NerfactoModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
        sd = {k: v.detach().clone() for k, v in params.items()}
        for i in range(len(cfg.prop_grids)):
            sd[f"proposal_networks.{i}.mlp_base.0.hash_table"] = sd[f"proposal_networks.{i}.encoding.hash_table"]
        sd["camera_optimizer.pose_adjustment"] = T(g["pose_adjustment"]).clone()
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        return model.cuda().train()

    def rel(a, b):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
        return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))

    n = g["origins"].shape[0]
    jit = [dev(g[f"j{i}"]) for i in range(3)]
    ref_losses = g[f"{mode}_losses"]
    # ---- module path -------------------------------------------------------------------------------------------------
    model = build()
    assert list(model.get_param_groups_ordered()) == ["fields", "proposal_networks", "camera_opt"]
    rb = RayBundle(origins=dev(g["origins"]), directions=dev(g["directions"]), pixel_area=torch.full((n, 1), 1e-6, device="cuda"),
                   camera_indices=dev(g["cams"])[:, None])
    out = model(rb, jitters=jit)
    close(rb.origins, g[f"{mode}_origins"], atol=1e-6, rtol=1e-6)  # the bundle now carries the corrected rays
    close(rb.directions, g[f"{mode}_directions"], atol=1e-6, rtol=1e-6)
    close(out["rgb"], g[f"{mode}_rgb"], atol=2e-5, rtol=0)
    batch = {"image": dev(g["target"])}
    losses = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
    for k, name in enumerate(("rgb_loss", "interlevel_loss", "distortion_loss", "camera_opt_regularizer")):
        close(losses[name], ref_losses[k], rtol=1e-3)
    sum(losses.values()).backward()
    g_pose_module = model.camera_optimizer.pose_adjustment.grad.clone()
    assert rel(g_pose_module, g[f"{mode}_g_pose"]) <= 1e-2, rel(g_pose_module, g[f"{mode}_g_pose"])
    # ---- explicit runner ---------------------------------------------------------------------------------------------
    model = build()
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    runner = NerfactoTrainStep(model, n, torch.device("cuda"))
    runner.set_batch(dev(g["origins"]), dev(g["directions"]), dev(g["cams"]), dev(g["target"]))
    runner.jitter.copy_(torch.cat([j.reshape(1, n) for j in jit], dim=0))
    arena.zero_grad(skip=runner.written_params())
    runner.forward_backward(True, draw_jitter=False)
    torch.cuda.synchronize()
    # the pose arithmetic ran as kernels (nsamd_camera_apply / nsamd_camera_backward), not as torch ops
    assert runner.cam_kernels and runner._corrected is None
    close(runner.origins, g[f"{mode}_origins"], atol=1e-6, rtol=1e-6)
    close(runner.directions, g[f"{mode}_directions"], atol=1e-6, rtol=1e-6)
    L = runner.n_prop
    d_o = sum(runner.d_origins[: L + 1])
    d_d = sum(runner.d_directions[: L + 1])
    assert rel(d_o, g[f"{mode}_d_origins"]) <= 2e-2 and rel(d_d, g[f"{mode}_d_directions"]) <= 1e-2, (
        rel(d_o, g[f"{mode}_d_origins"]), rel(d_d, g[f"{mode}_d_directions"]))
    g_pose = model.camera_optimizer.pose_adjustment.grad
    assert g_pose.data_ptr() >= arena.grad.data_ptr(), "the pose gradient lives in the arena (group camera_opt)"
    assert rel(g_pose, g[f"{mode}_g_pose"]) <= 1e-2 and rel(g_pose, g_pose_module.cpu().numpy()) <= 1e-4
    ld = runner.loss_dict()
    for k, name in enumerate(("rgb_loss", "interlevel_loss", "distortion_loss", "camera_opt_regularizer")):
        close(ld[name], ref_losses[k], rtol=1e-3)
    # ... bit-reproducibly (per-camera sums in a fixed order) ...
    first = g_pose.detach().clone()
    arena.zero_grad(skip=runner.written_params())
    runner.forward_backward(True, draw_jitter=False)
    assert torch.equal(g_pose, first)
    # ... equal to torch autograd over the mirror's exponential map on the SAME per-ray upstream gradients (the backward
    # kernel alone: identical inputs, so this one is tight) ...
    from nerfstudio_amd.cameras.lie_groups import exp_map_SE3, exp_map_SO3xR3

    co = model.camera_optimizer
    p2 = co.pose_adjustment.detach().clone().requires_grad_(True)
    c = (exp_map_SO3xR3 if mode == "SO3xR3" else exp_map_SE3)(p2[runner.camera_indices])
    o2 = runner.raw_origins + c[:, :3, 3]
    d2 = torch.bmm(c[:, :3, :3], runner.raw_directions[..., None]).squeeze(-1)
    reg2 = (p2[:, :3].norm(dim=-1).mean() * co.config.trans_l2_penalty + p2[:, 3:].norm(dim=-1).mean() * co.config.rot_l2_penalty)
    ((o2 * d_o).sum() + (d2 * d_d).sum() + reg2).backward()
    assert rel(first, p2.grad.cpu().numpy()) <= 1e-5, rel(first, p2.grad.cpu().numpy())
    # ... and to the all-torch route (NSAMD_CAMERA_KERNELS=0: index / exp map / bmm and autograd through them) within what an
    # ulp on a ray does to this gradient: the two routes round origins / directions differently in the last bit, samples at
    # t ~ 1000 move, cells flip (the conditioning the golden bound above carries as well)
    import os

    os.environ["NSAMD_CAMERA_KERNELS"] = "0"
    try:
        runner.cam_kernels = False
        arena.zero_grad(skip=runner.written_params())
        runner.forward_backward(True, draw_jitter=False)
        assert runner._corrected is None and rel(g_pose, first.cpu().numpy()) <= 5e-3, rel(g_pose, first.cpu().numpy())
        close(runner.loss_dict()["camera_opt_regularizer"], ref_losses[3], rtol=1e-3)
    finally:
        del os.environ["NSAMD_CAMERA_KERNELS"]
    runner.cam_kernels = True
    arena.zero_grad(skip=runner.written_params())
    runner.forward_backward(True, draw_jitter=False)
    assert torch.equal(g_pose, first)
    # one optimiser step per group: the camera group has its own learning rate (method_configs.py:117-120)
    before = model.camera_optimizer.pose_adjustment.detach().clone()
    arena.step(groups=["fields", "proposal_networks"])
    arena.step(groups=["camera_opt"], lr=1e-3)
    moved = (model.camera_optimizer.pose_adjustment.detach() - before).abs()
    assert float(moved.max()) <= 1e-3 * 1.001 and float(moved.max()) > 5e-4  # Adam's first step: lr * sign(g)


def test_fused_proposal_forward_equals_the_two_kernel_pair(F):
    """nsamd_density_field_fwd (hash grid + MLP + trunc_exp in one launch, features in registers) is bit-identical to
    nsamd_hashgrid_encode_fwd + nsamd_density_mlp_fwd — densities, and the optional enc / selector / pre outputs — in ray
    mode and on explicit positions, with and without the optional outputs; unsupported shapes report NSAMD_ERR_UNSUPPORTED."""
    from nerfstudio_amd import _native as N

    lib = N.load()
    torch.manual_seed(3)
    n, S = 211, 96
    M = n * S
    spec = F.HashGridSpec(5, 16, 256, 12)
    table = (torch.randn(5 << 12, 2) * 0.5).cuda()
    W0, b0, W1, b1 = (torch.randn(16, 10) * 0.4).cuda(), (torch.randn(16) * 0.1).cuda(), (torch.randn(1, 16) * 0.4).cuda(), torch.randn(1).cuda()
    dm = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), 10, 16, 0.01)
    o = (torch.randn(n, 3) * 0.7).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1).cuda()
    _, t_bins = F.piecewise_bins(torch.full((n,), 0.05).cuda(), torch.full((n,), 1000.0).cuda(), S, torch.rand(n).cuda())
    pos = (o[:, None] + d[:, None] * ((t_bins[:, :-1] + t_bins[:, 1:]) / 2)[..., None]).reshape(-1, 3).contiguous()
    e = lambda *s_: torch.empty(*s_, device="cuda")  # noqa: E731
    for P in (N.make_points(None, o, d, t_bins, S), N.make_points(positions=pos)):
        enc_a, sel_a, den_a, pre_a = e(10, M), e(M), e(M), e(M)
        N.check(lib.nsamd_hashgrid_encode_fwd(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), N.ptr(enc_a), 1, M,
                                              N.ptr(sel_a), N.stream()), "hash")
        N.check(lib.nsamd_density_mlp_fwd(N.ptr(enc_a), N.ptr(sel_a), M, dm, N.ptr(den_a), N.ptr(pre_a), N.stream()), "mlp")
        enc_b, sel_b, den_b, pre_b = e(10, M), e(M), e(M), e(M)
        N.check(lib.nsamd_density_field_fwd(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), dm, N.ptr(enc_b),
                                            N.ptr(sel_b), N.ptr(den_b), N.ptr(pre_b), N.stream()), "fused")
        for a, b in ((enc_a, enc_b), (sel_a, sel_b), (den_a, den_b), (pre_a, pre_b)):
            assert torch.equal(a, b)
        den_c = e(M)
        N.check(lib.nsamd_density_field_fwd(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), dm, None, None,
                                            N.ptr(den_c), None, N.stream()), "fused, densities only")
        assert torch.equal(den_a, den_c) and float(den_a.max()) > 0
    spec7 = F.HashGridSpec(7, 16, 256, 12)
    dm7 = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), 14, 16, 0.01)
    assert lib.nsamd_density_field_fwd(N.make_points(positions=pos), M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec7.native(),
                                       dm7, None, None, N.ptr(e(M)), None, N.stream()) == N.ERR_UNSUPPORTED


def test_losses_golden(F, golden):
    g = golden("losses")
    ws = [dev(g[f"w{i}"]).requires_grad_(True) for i in range(3)]
    bins = [dev(g[f"s_bins{i}"]) for i in range(3)]
    li = F.interlevel_loss(ws, bins)
    ld = F.distortion_loss(ws[-1], bins[-1])
    close(li, g["interlevel"], rtol=2e-5)
    close(ld, g["distortion"], rtol=2e-5)
    (li + 0.5 * ld).backward()
    for i in range(3):
        gclose(ws[i].grad if ws[i].grad is not None else torch.zeros_like(ws[i]), g[f"dw{i}"], 1e-4, f"dw{i}")


def test_raygen_golden(F, golden):
    g = golden("raygen")
    o, d, pa, dn = F.raygen_pinhole(dev(g["ray_indices"]), dev(g["c2w"]), dev(g["fx"]), dev(g["fy"]), dev(g["cx"]), dev(g["cy"]))
    exact(o, g["origins"])
    close(d, g["directions"], atol=2e-7)
    close(pa, g["pixel_area"], rtol=2e-4)
    close(dn, g["directions_norm"], rtol=1e-6)


def test_adam_matches_torch(F):
    """nsamd_adam_step against torch.optim.Adam(eps=1e-15) on the CPU: the kernel follows
    torch/optim/adam.py operation by operation, so both moments are the SAME BITS as torch's after every step; the
    parameters are the same bits as that operation order evaluated with IEEE arithmetic (numpy) and within one ulp of
    torch's — whose vectorised CPU sqrt is not correctly rounded (~0.6 % of its results are one ulp off; the rest of its
    chain reproduces exactly from its own sqrt values)."""
    torch.manual_seed(0)
    f = np.float32
    n = 10007  # not a multiple of 4: exercises the tail
    p0 = torch.randn(n)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-15)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pn = p0.numpy().copy()
    b1, b2 = 0.9, 0.999
    for step in range(1, 6):
        g = torch.randn(n)
        ref.grad = g.clone()
        opt.step()
        F.adam_step(p, g.cuda(), m, v, step, lr=1e-2, eps=1e-15)
        st = opt.state[ref]
        exact(m, st["exp_avg"], f"exp_avg after step {step}: the same bits as torch.optim.Adam")
        exact(v, st["exp_avg_sq"], f"exp_avg_sq after step {step}: the same bits as torch.optim.Adam")
        step_size, bc2_sqrt = f(1e-2 / (1 - b1**step)), f((1 - b2**step) ** 0.5)
        mn, vn = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy()
        denom = ((np.sqrt(vn).astype(f) / bc2_sqrt).astype(f) + f(1e-15)).astype(f)
        pn = (pn + ((f(-step_size) * mn).astype(f) / denom).astype(f)).astype(f)
        exact(p, pn, f"parameters after step {step}: adam.py's operation order in IEEE fp32")
        close(p, ref.detach(), atol=2e-9 * step, rtol=3e-7, msg=f"parameters after step {step} vs torch (its CPU sqrt: <= 1 ulp of a 1e-2 update per step)")
    # the device-resident step scalars of a replayed graph give the same bits as the host-derived ones
    p2, m2, v2 = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p3, m3, v3 = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = torch.randn(n).cuda()
        hyper = torch.tensor(F.adam_hyper(step, 1e-2), dtype=torch.float32).cuda()
        F.adam_step(p2, g, m2, v2, step, lr=1e-2, eps=1e-15)
        F.adam_step(p3, g, m3, v3, 1, lr=123.0, eps=1e-15, hyper_dev=hyper)  # (step / lr arguments are overridden)
        exact(p3, p2, "hyper_dev path")


# ---------------------------------------------------------------- whole pipeline ------------------------------------
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pipeline_golden(F, golden, mode):
    """Full nerfacto forward (+ backward in training) on the reference's own numbers: 16 rays, (256, 96, 48) samples."""
    from nerfstudio_amd.cameras.rays import RayBundle

    g = golden("pipeline")
    cfg = small_cfg(g["main_log2"], g["prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    model = _hip_model(cfg, params, training=(mode == "train"))
    n = g["origins"].shape[0]
    rb = RayBundle(origins=dev(g["origins"]), directions=dev(g["directions"]), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=dev(g["cams"])[:, None])
    jit = [dev(g[f"j{i}"]) for i in range(3)] if mode == "train" else None
    out = model(rb, jitters=jit)
    close(out["rgb"], g[f"{mode}_rgb"], atol=1e-4, rtol=0, msg="RGB L-inf (north_star bound 1e-4)")
    close(out["accumulation"], g[f"{mode}_acc"], atol=1e-4, rtol=0)
    close(out["expected_depth"], g[f"{mode}_expected_depth"], rtol=2e-3)
    close(out["depth"], g[f"{mode}_depth"], rtol=2e-3)
    for i in range(2):
        close(out[f"prop_depth_{i}"], g[f"{mode}_prop_depth_{i}"], rtol=2e-3)
    if mode == "train":
        for i in range(3):
            rs_i = out["ray_samples_list"][i]
            close(rs_i.pack.s_bins, g[f"{mode}_s_bins{i}"], atol=1e-5, rtol=0, msg=f"s_bins level {i}")
            close(out["weights_list"][i][..., 0], g[f"{mode}_w{i}"], atol=5e-5, rtol=2e-3, msg=f"weights level {i}")
        batch = {"image": dev(g["target"])}
        metrics = model.get_metrics_dict(out, batch)
        losses = model.get_loss_dict(out, batch, metrics)
        close(losses["rgb_loss"], g["loss_rgb"], rtol=1e-4)
        close(losses["interlevel_loss"], g["loss_interlevel"], rtol=2e-3)
        close(losses["distortion_loss"], g["loss_distortion"], rtol=2e-3)
        sum(losses.values()).backward()
        fld = model.field
        gclose_e2e(fld.mlp_base.encoding.hash_table.grad, g["g_main_table"], 5e-4, "main table")
        gclose_e2e(fld.embedding_appearance.embedding.weight.grad, g["g_emb"], 5e-4, "embedding")
        for j in range(2):
            gclose_e2e(fld.mlp_base.mlp.layers[j].weight.grad, g[f"g_base_W{j}"], 5e-4, f"base W{j}")
            gclose_e2e(fld.mlp_base.mlp.layers[j].bias.grad, g[f"g_base_b{j}"], 5e-4, f"base b{j}")
        for j in range(3):
            gclose_e2e(fld.mlp_head.layers[j].weight.grad, g[f"g_head_W{j}"], 5e-4, f"head W{j}")
            gclose_e2e(fld.mlp_head.layers[j].bias.grad, g[f"g_head_b{j}"], 5e-4, f"head b{j}")
        for i, net in enumerate(model.proposal_networks):
            gclose_e2e(net.encoding.hash_table.grad, g[f"g_prop{i}_table"], 1e-3, f"prop{i} table")
            for j in range(2):
                gclose_e2e(net.mlp_base[1].layers[j].weight.grad, g[f"g_prop{i}_W{j}"], 1e-3, f"prop{i} W{j}")
                gclose_e2e(net.mlp_base[1].layers[j].bias.grad, g[f"g_prop{i}_b{j}"], 1e-3, f"prop{i} b{j}")


def test_pipeline_vs_oracle_full_tables(F):
    """The real nerfacto configuration (T = 2^19 / 2^17, L = 16 / 5) on 256 rays against the oracle run here."""
    from nerfstudio_amd.cameras.rays import RayBundle

    cfg = orc.NerfactoCfg(num_images=20)
    params = orc.init_params(cfg, seed=2, table_std=0.3)
    model = _hip_model(cfg, params, training=True)
    n = 256
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=4)
    o[n // 2:] *= 5.0
    rs = np.random.RandomState(9)
    jit = [torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)) for _ in range(3)]
    with torch.no_grad():
        ref = orc.nerfacto_forward(params, cfg, o, d, cam, jit, training=True)
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=cam.cuda()[:, None])
    with torch.no_grad():
        out = model(rb, jitters=[j.cuda() for j in jit])
    close(out["rgb"], ref["rgb"], atol=1e-4, rtol=0, msg="RGB L-inf (north_star bound 1e-4)")
    close(out["accumulation"], ref["accumulation"], atol=1e-4, rtol=0)


def test_full_size_properties(F):
    """BASELINE size (4096 rays x 256/96/48 samples, full tables): size-independent invariants of the path."""
    from nerfstudio_amd.cameras.rays import RayBundle

    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0, table_std=0.3)
    model = _hip_model(cfg, params, training=True)
    n = 4096
    o, d, cam, _ = orc.synthetic_rays(n, cfg.num_images, seed=0)
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=cam.cuda()[:, None])
    out = model(rb)
    for i, s in enumerate((256, 96, 48)):
        rs_i = out["ray_samples_list"][i]
        sb, tb = rs_i.pack.s_bins, rs_i.pack.t_bins
        assert sb.shape == (n, s + 1)
        assert bool((sb[:, 1:] >= sb[:, :-1]).all()), "spacing bins must be sorted"
        assert bool((tb[:, 1:] >= tb[:, :-1]).all()), "euclidean bins must be sorted"
        assert float(sb.min()) >= 0.0 and float(sb.max()) <= 1.0
        w = out["weights_list"][i][..., 0]
        assert bool((w >= 0).all()) and float(w.sum(-1).max()) <= 1.0 + 1e-5, "weights are a sub-probability"
    rgb = out["rgb"]
    assert rgb.shape == (n, 3) and bool(torch.isfinite(rgb).all())
    assert float(rgb.min()) >= -1e-6 and float(rgb.max()) <= 1.0 + 1e-6, "convex combination of sigmoid colours"
    w = out["weights_list"][-1][..., 0]
    close(out["accumulation"][:, 0], w.sum(-1), atol=1e-5)
    # linearity of compositing in the colours
    c1 = torch.rand((n, 48, 3), device="cuda")
    c2 = torch.rand((n, 48, 3), device="cuda")
    a = F.composite(c1, w, None, "black", expected_depth=False)[0]
    b = F.composite(c2, w, None, "black", expected_depth=False)[0]
    ab = F.composite(c1 + c2, w, None, "black", expected_depth=False)[0]
    close(ab, a + b, atol=2e-6)
    # determinism of the forward (no atomics on the forward path)
    model.proposal_sampler._step = 0
    out2 = model(rb, jitters=None)
    assert out2["rgb"].shape == (n, 3)


def test_train_step_runner_matches_autograd_path(F):
    """nerfstudio_amd/train_step.py (explicit kernel schedule, gradients straight into the arena) against the
    nn.Module / autograd path on the same rays, jitter and parameters: outputs, losses and every gradient."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = small_cfg(12, 10, 6)
    params = orc.init_params(cfg, seed=7, table_std=0.4)
    n = 300  # not a multiple of 16 rays / 4 rays per workgroup
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=8)
    o[n // 2:] *= 4.0
    rs = np.random.RandomState(2)
    jit = torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)).cuda()

    # autograd path
    model_a = _hip_model(cfg, params)
    model_a.set_step(137)
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=cam.cuda()[:, None])
    out = model_a(rb, jitters=[jit[i][:, None] for i in range(3)])
    batch = {"image": tgt.cuda()}
    ld = model_a.get_loss_dict(out, batch, model_a.get_metrics_dict(out, batch))
    sum(ld.values()).backward()

    # runner
    model_b = _hip_model(cfg, params)
    model_b.set_step(137)
    arena = ParamArena(model_b.parameters())
    step = NerfactoTrainStep(model_b, n, torch.device("cuda"))
    step.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
    step.jitter.copy_(jit)
    step.anneal_dev.fill_(model_b.proposal_sampler._anneal)
    arena.zero_grad()
    step.forward_backward(updated=True, draw_jitter=False)
    ob = step.outputs()
    exact(step.s_bins[2], out["ray_samples_list"][2].pack.s_bins, "final sample bins")
    close(ob["rgb"], out["rgb"], atol=1e-6, rtol=0)
    close(ob["accumulation"], out["accumulation"], atol=1e-6, rtol=0)
    close(ob["expected_depth"], out["expected_depth"], rtol=1e-6)
    exact(ob["depth"], out["depth"])
    lb = step.loss_dict()
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss"):
        close(lb[k], ld[k], rtol=1e-5, atol=1e-9, msg=k)
    pa, pb = dict(model_a.named_parameters()), dict(model_b.named_parameters())
    for k in pa:
        gclose(pb[k].grad, pa[k].grad, 2e-5, k)
    # a non-update step leaves the proposal networks without gradient
    arena.zero_grad()
    step.forward_backward(updated=False, draw_jitter=False)
    for k, p in pb.items():
        if k.startswith("proposal_networks"):
            assert float(p.grad.abs().max()) == 0.0, k


def test_gradient_scaling_and_per_edge_jitter_module_path_runner_and_oracle(F):
    """NerfactoModelConfig.use_gradient_scaling (models/nerfacto.py:321-322 -> losses.py:534-569) and use_single_jitter=False
    (one draw per bin edge, ray_samplers.py:104-107, 318-322): the module / autograd path against the oracle (outputs,
    losses, gradients), and the explicit kernel schedule against the module path."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = small_cfg(12, 10, 6)
    params = orc.init_params(cfg, seed=17, table_std=0.4)
    n = 192
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=6)
    o[: n // 3] *= 0.2  # near samples: squared distance < 1, the scaling is not the identity there
    rs = np.random.RandomState(1)
    jit = [torch.from_numpy(rs.uniform(0, 1, (n, s + 1)).astype(np.float32)) for s in (256, 96, 48)]  # one per bin EDGE

    def build():
        m = _hip_model(cfg, params)
        m.config.use_gradient_scaling = True
        m.config.use_single_jitter = False
        m.proposal_sampler.initial_sampler.single_jitter = False
        m.proposal_sampler.pdf_sampler.single_jitter = False
        m.set_step(137)
        return m

    model_a = build()
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=cam.cuda()[:, None])
    out = model_a(rb, jitters=[j.cuda() for j in jit])
    batch = {"image": tgt.cuda()}
    ld = model_a.get_loss_dict(out, batch, model_a.get_metrics_dict(out, batch))
    sum(ld.values()).backward()
    # ---- oracle
    oparams = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    ref = orc.nerfacto_forward(oparams, cfg, o, d, cam, jit, training=True, anneal=model_a.proposal_sampler._anneal,
                               use_gradient_scaling=True)
    lr = orc.nerfacto_losses(ref, tgt, cfg)
    sum(lr.values()).backward()
    close(out["rgb"], ref["rgb"], atol=1e-4, rtol=0, msg="RGB (per-edge jitter)")
    for i in range(3):
        # (per-edge jitter puts 97 / 49 independent draws on every ray's CDF: a few of 18 624 edges sit where a 1e-6
        # difference of the proposal weights moves them by 1.2e-5; 1e-5 holds for the single-jitter fixtures)
        close(out["ray_samples_list"][i].pack.s_bins, ref["s_bins_list"][i], atol=3e-5, rtol=0, msg=f"s_bins level {i}")
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss"):
        close(ld[k], lr[k], rtol=2e-3, atol=1e-9, msg=k)
    fld = model_a.field
    gclose_e2e(fld.mlp_base.encoding.hash_table.grad, oparams["field.mlp_base.model.0.hash_table"].grad, 5e-4, "main table")
    gclose_e2e(fld.mlp_head.layers[0].weight.grad, oparams["field.mlp_head.layers.0.weight"].grad, 5e-4, "head W0")
    gclose_e2e(fld.mlp_base.mlp.layers[1].weight.grad, oparams["field.mlp_base.model.1.layers.1.weight"].grad, 5e-4, "base W1")
    # the scaling matters on this batch: without it the table gradient is a different one
    oparams2 = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    ref2 = orc.nerfacto_forward(oparams2, cfg, o, d, cam, jit, training=True, anneal=model_a.proposal_sampler._anneal)
    sum(orc.nerfacto_losses(ref2, tgt, cfg).values()).backward()
    g1, g2 = oparams["field.mlp_base.model.0.hash_table"].grad, oparams2["field.mlp_base.model.0.hash_table"].grad
    # (most of this batch's samples lie beyond distance 1, where the scale is clamped to 1: the difference is 0.8 % here —
    # 16x the tolerance of the comparison above, so a path that dropped the scaling would fail it)
    assert float((g1 - g2).norm() / g2.norm()) > 4 * 5e-4
    # ---- the explicit kernel schedule with the same options
    model_b = build()
    arena = ParamArena(model_b.parameters())
    step = NerfactoTrainStep(model_b, n, torch.device("cuda"))
    assert not step.single_jitter and step.gradient_scaling
    step.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
    for lvl in range(3):
        step.jitter_edges[lvl].copy_(jit[lvl])
    step.anneal_dev.fill_(model_b.proposal_sampler._anneal)
    arena.zero_grad()
    step.forward_backward(updated=True, draw_jitter=False)
    exact(step.s_bins[2], out["ray_samples_list"][2].pack.s_bins, "final sample bins")
    close(step.outputs()["rgb"], out["rgb"], atol=1e-6, rtol=0)
    lb = step.loss_dict()
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss"):
        close(lb[k], ld[k], rtol=1e-5, atol=1e-9, msg=k)
    pa, pb = dict(model_a.named_parameters()), dict(model_b.named_parameters())
    for k in pa:
        gclose(pb[k].grad, pa[k].grad, 2e-5, k)
    # device-drawn per-edge jitter: every level's buffer is refreshed
    before = [j.clone() for j in step.jitter_edges]
    arena.zero_grad()
    step.forward_backward(updated=False, draw_jitter=True)
    assert all(not torch.equal(a, b) for a, b in zip(before, step.jitter_edges))
    assert bool(torch.isfinite(step.outputs()["rgb"]).all())


@pytest.mark.parametrize("camera_mode", ["off", "SO3xR3"])
def test_fused_train_step_behind_the_model_api(F, camera_mode):
    """config.fused_train_step: get_outputs -> get_metrics_dict -> get_loss_dict -> sum(losses).backward() driven exactly
    as nerfstudio's pipeline / trainer drive a model (pipelines/base_pipeline.py:290-303, engine/trainer.py:514-516), with
    the explicit kernel schedule underneath (fused_step.FusedTrainStep), against the nn.Module / autograd path: outputs,
    metrics, losses and every parameter gradient (also the camera optimiser's); `zero_grad(set_to_none=True)` semantics
    (a non-update iteration leaves the proposal networks without gradient), and accumulation into existing gradients."""
    from nerfstudio_amd.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio_amd.cameras.rays import RayBundle

    cfg = small_cfg(12, 10, 6)
    params = orc.init_params(cfg, seed=11, table_std=0.4)
    n = 300
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=12)
    o[n // 2:] *= 4.0
    jit = torch.from_numpy(np.random.RandomState(5).uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    batch = {"image": tgt.cuda()}

    def build(fused):
        m = _hip_model(cfg, params)
        if camera_mode != "off":
            m.camera_optimizer = CameraOptimizerConfig(mode=camera_mode).setup(num_cameras=cfg.num_images, device="cuda")
            with torch.no_grad():  # non-trivial pose corrections, identical in both models
                g = torch.Generator().manual_seed(3)
                m.camera_optimizer.pose_adjustment.copy_((torch.randn(cfg.num_images, 6, generator=g) * 0.01).cuda())
        m.config.fused_train_step = fused
        m.set_step(137)
        return m

    def iteration(m, step_jitter=True):
        rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                       camera_indices=cam.cuda()[:, None])
        out = m(rb, jitters=[jit[i][:, None] for i in range(3)] if step_jitter else None)
        metrics = m.get_metrics_dict(out, batch)
        losses = m.get_loss_dict(out, batch, metrics)
        import functools

        functools.reduce(torch.add, losses.values()).backward()  # engine/trainer.py:514
        return out, metrics, losses

    ma, mb = build(False), build(True)
    out_a, met_a, ld_a = iteration(ma)
    out_b, met_b, ld_b = iteration(mb)
    assert "fused_step" in out_b and "fused_step" not in out_a
    close(out_b["rgb"], out_a["rgb"], atol=1e-6, rtol=0)
    close(out_b["accumulation"], out_a["accumulation"], atol=1e-6, rtol=0)
    close(out_b["expected_depth"], out_a["expected_depth"], rtol=1e-6)
    exact(out_b["depth"], out_a["depth"])
    for i in range(2):
        exact(out_b[f"prop_depth_{i}"], out_a[f"prop_depth_{i}"])
    assert set(ld_b) == set(ld_a) and set(met_a) <= set(met_b) | {"distortion"}
    for k in ld_a:
        close(ld_b[k], ld_a[k], rtol=1e-5, atol=1e-9, msg=k)
    close(met_b["psnr"], met_a["psnr"], rtol=1e-5)
    close(met_b["distortion"], met_a["distortion"], rtol=1e-5, atol=1e-9)
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    assert set(pa) == set(pb)

    def same_gradient(k):
        if k.startswith("camera_optimizer"):  # far rays (t ~ 1000) condition this one: relative L2, as the golden test does
            a, b = pb[k].grad.double(), pa[k].grad.double()
            assert float((a - b).norm() / b.norm()) <= 1e-3 and float(b.abs().max()) > 0, k
        else:
            gclose(pb[k].grad, pa[k].grad, 2e-5, k)

    for k in pa:
        assert pb[k].grad is not None, k
        same_gradient(k)

    # the trainer's zero_grad(set_to_none=True) (engine/optimizers.py:160-172), then a NON-update iteration: the
    # proposal networks are run without gradient (ray_samplers.py:590-599), their .grad stays None as on the module path
    first = {k: p.grad.clone() for k, p in pb.items()}
    for m in (ma, mb):
        m.zero_grad(set_to_none=True)
        m.after_step(137)  # AFTER_TRAIN_ITERATION: steps_since_update 0 -> 1, not > update_sched(137) = 1
        m.set_step(138)
    assert not mb.proposal_sampler.updated_this_step()
    iteration(ma)
    iteration(mb)
    for k in pa:
        if k.startswith("proposal_networks"):
            assert pb[k].grad is None and pa[k].grad is None, k
        else:
            same_gradient(k)
    # a gradient that is already there is accumulated into — the main table's too (its scatter normally overwrites)
    table = next(k for k, prm in pb.items() if prm is mb.field.mlp_base.encoding.hash_table)
    once = pb[table].grad.clone()
    mb.after_step(138), mb.set_step(138)
    mb.proposal_sampler._steps_since_update = 0  # keep it a non-update iteration
    iteration(mb)
    gclose(pb[table].grad, 2.0 * once, 1e-5, "accumulated table gradient")
    assert first[table].shape == once.shape


def test_train_step_runner_random_background(F, monkeypatch):
    """background_color="random" on the fused runner (renderers.py:112-115, 194-196; models/nerfacto.py:377-381) against the
    module path with the same rand_like draw: rendered colour (no background), losses, every gradient."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.model_components import renderers as R
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = small_cfg(12, 10, 6)
    params = orc.init_params(cfg, seed=17, table_std=0.4)
    n = 130
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=18)
    jit = torch.rand(3, n).cuda()
    bg = torch.rand(n, 3).cuda()

    def model():
        m = _hip_model(cfg, params)
        m.config.background_color = "random"
        m.renderer_rgb.background_color = "random"
        m.set_step(50)
        return m

    model_a = model()
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=cam.cuda()[:, None])
    out = model_a(rb, jitters=[jit[i][:, None] for i in range(3)])
    monkeypatch.setattr(R.torch, "rand_like", lambda t: bg.to(t))
    batch = {"image": tgt.cuda()}
    ld = model_a.get_loss_dict(out, batch, model_a.get_metrics_dict(out, batch))
    monkeypatch.undo()
    sum(ld.values()).backward()

    model_b = model()
    arena = ParamArena(model_b.parameters())
    step = NerfactoTrainStep(model_b, n, torch.device("cuda"))
    assert step.bg_mode == 3
    step.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
    step.jitter.copy_(jit)
    step.bg_rays.copy_(bg)
    step.anneal_dev.fill_(model_b.proposal_sampler._anneal)
    arena.zero_grad()
    step.forward_backward(updated=True, draw_jitter=False)
    close(step.outputs()["rgb"], out["rgb"], atol=1e-6, rtol=0)
    lb = step.loss_dict()
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss"):
        close(lb[k], ld[k], rtol=1e-5, atol=1e-9, msg=k)
    pa, pb = dict(model_a.named_parameters()), dict(model_b.named_parameters())
    for k in pa:
        gclose(pb[k].grad, pa[k].grad, 2e-5, k)
    # a drawn step fills the background itself
    step.forward_backward(updated=False, draw_jitter=True)
    assert not torch.equal(step.bg_rays, bg) and float(step.bg_rays.min()) >= 0.0 and float(step.bg_rays.max()) < 1.0


def test_eval_render_path_full_image(F):
    """§8 f3: RayGenerator over a full (small) image -> chunked eval render
    (Model.get_outputs_for_camera_ray_bundle, models/base_model.py:178-205) against the oracle in eval mode:
    no jitter, near plane reset to 0, mean appearance embedding, nan_to_num + clamp in the RGB renderer."""
    from nerfstudio_amd.model_components.ray_generators import RayGenerator

    class Cams:  # the tensors nerfstudio's `Cameras` exposes
        pass

    H, W = 18, 26
    rs = np.random.RandomState(12)
    q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
    cams = Cams()
    cams.camera_to_worlds = torch.from_numpy(np.concatenate([q, [[0.3], [-0.2], [0.5]]], axis=1).astype(np.float32))[None]
    cams.fx = torch.tensor([[30.0]])
    cams.fy = torch.tensor([[31.0]])
    cams.cx = torch.tensor([[W / 2.0]])
    cams.cy = torch.tensor([[H / 2.0]])
    cfg = small_cfg(11, 9, 4)
    params = orc.init_params(cfg, seed=21, table_std=0.4)
    model = _hip_model(cfg, params, training=False)
    model.config.eval_num_rays_per_chunk = 200  # 468 rays -> 3 chunks, the last one ragged
    gen = RayGenerator(cams).cuda()
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    idx = torch.stack([torch.zeros_like(yy), yy, xx], dim=-1).reshape(-1, 3).cuda()
    rb = gen(idx)
    ref_rays = orc.raygen_pinhole(idx.cpu(), cams.camera_to_worlds, cams.fx[:, 0], cams.fy[:, 0], cams.cx[:, 0], cams.cy[:, 0])
    close(rb.directions, ref_rays["directions"], atol=2e-7)
    image_rb = rb._map(lambda t: t.view(H, W, -1))
    out = model.get_outputs_for_camera_ray_bundle(image_rb)
    assert out["rgb"].shape == (H, W, 3) and out["depth"].shape == (H, W, 1) and out["accumulation"].shape == (H, W, 1)
    with torch.no_grad():
        ref = orc.nerfacto_forward(params, cfg, ref_rays["origins"], ref_rays["directions"],
                                   torch.zeros(H * W, dtype=torch.int64), None, training=False)
    close(out["rgb"].reshape(-1, 3), ref["rgb"], atol=1e-4, rtol=0, msg="eval RGB L-inf (north_star bound 1e-4)")
    close(out["accumulation"].reshape(-1, 1), ref["accumulation"], atol=1e-4, rtol=0)
    assert float(out["rgb"].min()) >= 0.0 and float(out["rgb"].max()) <= 1.0
    # the expected-depth clip uses the batch-global min/max, which the chunk loop evaluates per chunk, exactly like
    # the reference's loop does; compare the un-clipped quantity through the median depth instead
    close(out["depth"].reshape(-1, 1), ref["depth"], rtol=2e-3)


def test_eval_renderer_equals_the_module_chunk_loop(F, monkeypatch):
    """eval_render.EvalRenderer (device-side chunk loop: one captured kernel schedule per chunk, outputs copied into
    preallocated image buffers) against the reference-shaped Python loop over `forward` + torch.cat: every output of a
    frame whose ray count is not a multiple of the chunk — bit for bit (same kernels, same order), eager and replayed,
    and again after the parameters changed (the captured graph reads them from the same arena)."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.eval_render import EvalRenderer

    cfg = small_cfg(12, 10, 5)
    model = _hip_model(cfg, orc.init_params(cfg, seed=3, table_std=0.4), training=False)
    model.config.eval_num_rays_per_chunk = 512
    model.proposal_sampler.set_anneal(0.37)
    H, W = 37, 41  # 1517 rays: 2 full chunks + 493
    o, d, cam, _ = orc.synthetic_rays(H * W, cfg.num_images, seed=9)
    o[::3] *= 3.0
    rb = RayBundle(origins=o.cuda().view(H, W, 3), directions=d.cuda().view(H, W, 3),
                   pixel_area=torch.full((H, W, 1), 1e-6).cuda(), camera_indices=torch.zeros((H, W, 1), dtype=torch.int64).cuda())
    monkeypatch.setenv("NSAMD_EVAL_RUNNER", "0")
    ref = model.get_outputs_for_camera_ray_bundle(rb)
    monkeypatch.setenv("NSAMD_EVAL_RUNNER", "1")
    keys = {"rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1"}
    assert keys <= set(ref)
    for use_graph in (False, True):
        out = EvalRenderer(model, use_graph=use_graph).render(rb)
        for k in keys:
            assert out[k].shape == ref[k].shape, k
            assert torch.equal(out[k], ref[k]), f"{k} ({'graph' if use_graph else 'eager'}): max |d| = {float((out[k] - ref[k]).abs().max()):.3e}"
    # through the model entry point, twice (the second frame replays the captured chunk), then with changed parameters
    a = model.get_outputs_for_camera_ray_bundle(rb)
    b = model.get_outputs_for_camera_ray_bundle(rb)
    assert model._eval_runner.graph is not None and all(torch.equal(a[k], ref[k]) and torch.equal(b[k], ref[k]) for k in keys)
    with torch.no_grad():
        model.field.mlp_base.encoding.hash_table.mul_(1.3)
        model.field.embedding_appearance.embedding.weight.add_(0.05)
    monkeypatch.setenv("NSAMD_EVAL_RUNNER", "0")
    ref2 = model.get_outputs_for_camera_ray_bundle(rb)
    monkeypatch.setenv("NSAMD_EVAL_RUNNER", "1")
    c = model.get_outputs_for_camera_ray_bundle(rb)
    assert not torch.equal(ref2["rgb"], ref["rgb"]) and all(torch.equal(c[k], ref2[k]) for k in keys)


def test_eval_render_of_a_camera_generates_its_rays_inside_the_chunk_loop(F):
    """Model.get_outputs_for_camera (models/base_model.py:166-175) for one undistorted pinhole camera: the device-side chunk loop
    generates each chunk's rays itself (nsamd_raygen_pinhole_grid — no [H, W] bundle, no index list) and must give, bit for bit,
    what the bundle of `generate_rays(camera_indices=0, keep_shape=True)` gives through get_outputs_for_camera_ray_bundle — the
    bundle here comes from nsamd_raygen_pinhole, which tests/golden pins to the reference's own generate_rays. A frame that is
    not a multiple of the chunk; a camera the grid generator does not cover (fisheye) must take the bundle route."""
    from nerfstudio_amd.model_components.ray_generators import RayGenerator

    cfg = small_cfg(12, 10, 5)
    model = _hip_model(cfg, orc.init_params(cfg, seed=3, table_std=0.4), training=False)
    model.config.eval_num_rays_per_chunk = 512
    H, W = 37, 41  # 1517 rays: 2 full chunks + 493
    c2w = torch.tensor([[0.96, -0.10, 0.26, 0.3], [0.05, 0.98, 0.19, -0.2], [-0.27, -0.17, 0.95, 0.9]])

    class Cam:
        camera_to_worlds = c2w[None].cuda()
        fx, fy = torch.tensor([[0.9 * W]]).cuda(), torch.tensor([[0.8 * W]]).cuda()
        cx, cy = torch.tensor([[W / 2.0 + 0.25]]).cuda(), torch.tensor([[H / 2.0 - 0.5]]).cuda()
        height, width = torch.tensor([[H]]), torch.tensor([[W]])
        camera_type = torch.tensor([[1]])  # CameraType.PERSPECTIVE
        distortion_params = None
        calls = 0

        def generate_rays(self, camera_indices=0, keep_shape=True, obb_box=None):
            type(self).calls += 1
            yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            idx = torch.stack([torch.zeros_like(yy), yy, xx], dim=-1).reshape(-1, 3).cuda()
            import types

            pin = types.SimpleNamespace(camera_to_worlds=self.camera_to_worlds, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy)
            return RayGenerator(pin).cuda()(idx).reshape((H, W))

    cam = Cam()
    ref = model.get_outputs_for_camera_ray_bundle(cam.generate_rays())
    Cam.calls = 0
    for _ in range(2):  # (the second frame replays the captured chunk)
        out = model.get_outputs_for_camera(cam)
        assert Cam.calls == 0, "the pinhole camera must not build a ray bundle"
        for k in ("rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1"):
            assert out[k].shape == ref[k].shape == (H, W, ref[k].shape[-1]), k
            assert torch.equal(out[k], ref[k]), f"{k}: max |d| = {float((out[k] - ref[k]).abs().max()):.3e}"
    cam.camera_type = torch.tensor([[2]])  # fisheye: the general generator's job
    out = model.get_outputs_for_camera(cam)
    assert Cam.calls == 1 and torch.equal(out["rgb"], ref["rgb"])


def test_training_trajectory_matches_oracle(F):
    """30 full training steps (forward, 3 losses, backward, Adam) on the GPU runner and on the CPU oracle from the same
    initialisation, rays and jitter draws: the loss curves must track each other (tight at first, ulp-level
    differences grow slowly through the optimisation)."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = small_cfg(12, 10, 5)
    n, steps = 64, 30
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=31)
    rs = np.random.RandomState(32)
    jit = rs.uniform(0, 1, (steps, 3, n)).astype(np.float32)

    params = orc.init_params(cfg, seed=33, table_std=0.3)
    model = _hip_model(cfg, params)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)  # the reference's two optimiser groups
    assert list(arena.groups) == ["fields", "proposal_networks"] and arena.groups["fields"][0] == 0
    runner = NerfactoTrainStep(model, n, torch.device("cuda"))
    runner.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
    hip_losses, schedule, anneals = [], [], []
    for step in range(steps):
        model.set_step(step)
        ps = model.proposal_sampler
        updated = ps.updated_this_step()
        schedule.append(updated)
        anneals.append(ps._anneal)
        runner.anneal_dev.fill_(ps._anneal)
        runner.jitter.copy_(torch.from_numpy(jit[step]))
        arena.zero_grad()
        runner.forward_backward(updated, draw_jitter=False)
        # a group is stepped only when it received gradients (engine/optimizers.py:160-172)
        arena.step(groups=["fields", "proposal_networks"] if updated else ["fields"])
        hip_losses.append(float(sum(runner.loss_dict().values())))
        if updated:
            ps.mark_updated()
        model.after_step(step)

    oparams = orc.init_params(cfg, seed=33, table_std=0.3)
    plist = list(oparams.values())
    for p in plist:
        p.requires_grad_(True)
    opt = torch.optim.Adam(plist, lr=1e-2, eps=1e-15)
    ref_losses = []
    for step in range(steps):
        opt.zero_grad(set_to_none=True)
        j = [torch.from_numpy(jit[step, i])[:, None] for i in range(3)]
        out = orc.nerfacto_forward(oparams, cfg, o, d, cam, j, training=True, anneal=anneals[step],
                                   proposal_requires_grad=schedule[step])
        loss = sum(orc.nerfacto_losses(out, tgt, cfg).values())
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    hip_losses, ref_losses = np.array(hip_losses), np.array(ref_losses)
    assert arena.step_counts["fields"] == steps and arena.step_counts["proposal_networks"] == sum(schedule) < steps
    assert ref_losses[-1] < 0.7 * ref_losses[0], "the oracle itself should be learning"
    np.testing.assert_allclose(hip_losses[:5], ref_losses[:5], rtol=2e-4)
    np.testing.assert_allclose(hip_losses, ref_losses, rtol=3e-2)


def test_standalone_mlp_any_shape(F):
    """§8 a9: `MLP.forward` of arbitrary widths (csrc/linear.hip) against the oracle's mlp_forward (= the reference's
    MLP.pytorch_fwd): reference test contract tests/field_components/test_mlp.py:11-28 (shape) plus values and grads."""
    from nerfstudio_amd.field_components.mlp import MLP

    rs = np.random.RandomState(41)
    for in_dim, layers, width, out_dim, out_act, M in ((6, 2, 8, 10, None, 10), (63, 3, 64, 3, torch.nn.Sigmoid(), 1000),
                                                      (32, 4, 128, 16, None, 257), (5, 1, 16, 7, None, 17), (10, 2, 16, 1, None, 1)):
        mlp = MLP(in_dim=in_dim, num_layers=layers, layer_width=width, out_dim=out_dim, out_activation=out_act).cuda()
        x = torch.from_numpy(rs.standard_normal((M, in_dim)).astype(np.float32))
        params = {f"layers.{i}.{k}": getattr(l, k).detach().cpu().clone().requires_grad_(True)
                  for i, l in enumerate(mlp.layers) for k in ("weight", "bias")}
        xr = x.clone().requires_grad_(True)
        ref = orc.mlp_forward(xr, params, "", out_activation="sigmoid" if out_act is not None else None)
        xg = x.cuda().requires_grad_(True)
        out = mlp(xg)
        assert out.shape == (M, out_dim)
        close(out, ref, atol=2e-6, rtol=1e-5, msg=f"mlp {in_dim}->{width}x{layers}->{out_dim}")
        g = torch.from_numpy(rs.standard_normal((M, out_dim)).astype(np.float32))
        (ref * g).sum().backward()
        (out * g.cuda()).sum().backward()
        gclose(xg.grad, xr.grad, 1e-5, "dx")
        for i, l in enumerate(mlp.layers):
            gclose(l.weight.grad, params[f"layers.{i}.weight"].grad, 1e-5, f"dW{i}")
            gclose(l.bias.grad, params[f"layers.{i}.bias"].grad, 1e-5, f"db{i}")
    assert mlp(torch.zeros((4, 5, 10), device="cuda")).shape == (4, 5, 1)  # batch shape preserved


def test_uniform_initial_sampler(F):
    """UniformSampler (ray_samplers.py:131-155; the Blender recipe's proposal-initial-sampler) + PDF resampling on top of
    it: bit-exact bins against the oracle."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.model_components.ray_samplers import PDFSampler, UniformSampler

    n = 37
    rs = np.random.RandomState(51)
    nears, fars = torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)  # Blender near / far
    jit = [torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)) for _ in range(2)]
    rb = RayBundle(origins=torch.zeros(n, 3).cuda(), directions=torch.ones(n, 3).cuda(), pixel_area=torch.ones(n, 1).cuda(),
                   nears=nears.cuda(), fars=fars.cuda())
    smp = UniformSampler(single_jitter=True).cuda().train()
    rs0 = smp(rb, num_samples=64, jitter=jit[0].cuda())
    so, to = orc.piecewise_bins(nears, fars, 64, jit[0], uniform=True)
    exact(rs0.pack.s_bins, so)
    exact(rs0.pack.t_bins, to)
    assert float(rs0.pack.t_bins.min()) >= 2.0 and float(rs0.pack.t_bins.max()) <= 6.0
    w = torch.from_numpy(rs.uniform(0, 1, (n, 64, 1)).astype(np.float32) ** 4)
    pdf = PDFSampler(include_original=False, single_jitter=True).cuda().train()
    rs1 = pdf(rb, rs0, w.cuda(), num_samples=32, jitter=jit[1].cuda())
    s1o, t1o, _ = orc.pdf_resample(so, w[..., 0], 32, jit[1], nears, fars, uniform=True)
    exact(rs1.pack.s_bins, s1o)
    exact(rs1.pack.t_bins, t1o)


def test_fused_training_entry_points_equal_the_separate_ones(F):
    """nsamd_proposal_resample / nsamd_render_train(_bwd) / nsamd_proposal_losses are launch fusions: same kernels'
    arithmetic, so the outputs must be bit-identical to the separate entry points on the same inputs."""
    import ctypes as C

    from nerfstudio_amd import _native as N

    lib = N.load()
    torch.manual_seed(3)
    n, S0, S1 = 1027, 96, 48  # ragged ray count: tail workgroup
    st = N.stream()
    e = lambda *s: torch.empty(*s, device="cuda")
    nears, fars = torch.full((n,), 0.05, device="cuda"), torch.full((n,), 1000.0, device="cuda")
    s0, t0 = F.piecewise_bins(nears, fars, S0, torch.rand(n, device="cuda"))
    dens0 = torch.rand(n, S0, device="cuda") * 3
    dens0[5] = 0.0  # an all-zero ray
    jit = torch.rand(n, device="cuda")
    u = F._linspace("u", S1, torch.device("cuda"))
    anneal = torch.tensor([0.7], device="cuda")
    # separate
    w_a = e(n, S0); s_a, t_a, med_a = e(n, S1 + 1), e(n, S1 + 1), e(n)
    N.check(lib.nsamd_weights_fwd(N.ptr(t0), N.ptr(dens0), n, S0, N.ptr(w_a), st), "w")
    N.check(lib.nsamd_composite_fwd(None, N.ptr(w_a), N.ptr(t0), n, S0, N.BG_NONE, None, 0, None, None, None, N.ptr(med_a),
                                    None, None, st), "med")
    N.check(lib.nsamd_pdf_resample(N.ptr(s0), N.ptr(w_a), S0, N.ptr(u), N.ptr(jit), N.ptr(nears), N.ptr(fars), 1.0,
                                   N.ptr(anneal), 0.01, 1e-5, 1.0 / (2 * (S1 + 1)), 0, 0, 0, n, S1, N.ptr(s_a), N.ptr(t_a), None, st),
            "pdf")
    # fused
    w_b = e(n, S0); s_b, t_b, med_b = e(n, S1 + 1), e(n, S1 + 1), e(n)
    N.check(lib.nsamd_proposal_resample(N.ptr(t0), N.ptr(s0), N.ptr(dens0), S0, N.ptr(u), N.ptr(jit), N.ptr(nears),
                                        N.ptr(fars), 1.0, N.ptr(anneal), 0.01, 1e-5, 1.0 / (2 * (S1 + 1)), 0, n, S1,
                                        N.ptr(w_b), N.ptr(med_b), N.ptr(s_b), N.ptr(t_b), st), "fused")
    for a, b, name in ((w_a, w_b, "weights"), (med_a, med_b, "median"), (s_a, s_b, "s_bins"), (t_a, t_b, "t_bins")):
        assert torch.equal(a, b), name

    # ---- main render: weights + composite + mse, and its backward ----
    rgb = torch.rand(n, S1, 3, device="cuda"); dens1 = torch.rand(n, S1, device="cuda") * 5
    target = torch.rand(n, 3, device="cuda")
    bgv = (C.c_float * 3)(0.1, 0.2, 0.3)
    for bg in (N.BG_NONE, 1, 2):
        w_a = e(n, S1); rgb_a, acc_a, dexp_a, dmed_a = e(n, 3), e(n), e(n), e(n)
        ws_a = e(2 + 2 * ((n + 3) // 4)); loss_a = torch.zeros(1, device="cuda"); dro_a = e(n, 3)
        N.check(lib.nsamd_weights_fwd(N.ptr(t_a), N.ptr(dens1), n, S1, N.ptr(w_a), st), "w")
        N.check(lib.nsamd_composite_fwd(N.ptr(rgb), N.ptr(w_a), N.ptr(t_a), n, S1, bg, bgv, 0, N.ptr(rgb_a), N.ptr(acc_a),
                                        N.ptr(dexp_a), N.ptr(dmed_a), None, N.ptr(ws_a), st), "c")
        N.check(lib.nsamd_mse_loss(N.ptr(rgb_a), N.ptr(target), 3 * n, 1.0 / (3 * n), N.ptr(loss_a), N.ptr(dro_a), st), "m")
        w_b = e(n, S1); rgb_b, acc_b, dexp_b, dmed_b = e(n, 3), e(n), e(n), e(n)
        ws_b = e(2 + 2 * ((n + 3) // 4)); sq_b = e(n); dro_b = e(n, 3)
        N.check(lib.nsamd_render_train(N.ptr(rgb), N.ptr(dens1), N.ptr(t_a), n, S1, bg, bgv, N.ptr(target), 1.0 / (3 * n),
                                       N.ptr(w_b), N.ptr(rgb_b), N.ptr(acc_b), N.ptr(dexp_b), N.ptr(dmed_b), N.ptr(ws_b),
                                       N.ptr(sq_b), N.ptr(dro_b), None, st), "rt")
        for a, b, name in ((w_a, w_b, "w"), (rgb_a, rgb_b, "rgb"), (acc_a, acc_b, "acc"), (dexp_a, dexp_b, "dexp"),
                           (dmed_a, dmed_b, "dmed"), (dro_a, dro_b, "d_rgb_out")):
            assert torch.equal(a, b), (bg, name)
        close(sq_b.sum(), loss_a[0], rtol=1e-5)
        dw_add = torch.randn(n, S1, device="cuda") * 1e-3
        drgb_a, dw_a, dd_a = e(n, S1, 3), e(n, S1), e(n, S1)
        N.check(lib.nsamd_composite_bwd(N.ptr(rgb), N.ptr(w_a), None, n, S1, bg, bgv, N.ptr(dro_a), None, None, None,
                                        N.ptr(dw_add), N.ptr(drgb_a), N.ptr(dw_a), st), "cb")
        N.check(lib.nsamd_weights_bwd(N.ptr(t_a), N.ptr(dens1), N.ptr(dw_a), n, S1, N.ptr(dd_a), st), "wb")
        drgb_b, dd_b = e(n, S1, 3), e(n, S1)
        N.check(lib.nsamd_render_train_bwd(N.ptr(rgb), N.ptr(w_a), N.ptr(dens1), N.ptr(t_a), n, S1, bg, bgv, N.ptr(dro_a),
                                           N.ptr(dw_add), N.ptr(drgb_b), N.ptr(dd_b), None, st), "rtb")
        assert torch.equal(drgb_a, drgb_b) and torch.equal(dd_a, dd_b), bg
    # background_color="random" (mode 3): rgb_out carries no background, the loss is on rgb_out + bg (1 - acc); against the
    # same formulas in torch autograd (renderers.py:112-115, 194-196)
    bg_rays = torch.rand(n, 3, device="cuda")
    w_r = e(n, S1); rgb_r, acc_r, sq_r, dro_r = e(n, 3), e(n), e(n), e(n, 3)
    N.check(lib.nsamd_render_train(N.ptr(rgb), N.ptr(dens1), N.ptr(t_a), n, S1, 3, None, N.ptr(target), 1.0 / (3 * n),
                                   N.ptr(w_r), N.ptr(rgb_r), N.ptr(acc_r), None, None, None, N.ptr(sq_r), N.ptr(dro_r),
                                   N.ptr(bg_rays), st), "rt3")
    drgb_r, dd_r = e(n, S1, 3), e(n, S1)
    N.check(lib.nsamd_render_train_bwd(N.ptr(rgb), N.ptr(w_r), N.ptr(dens1), N.ptr(t_a), n, S1, 3, None, N.ptr(dro_r), None,
                                       N.ptr(drgb_r), N.ptr(dd_r), N.ptr(bg_rays), st), "rtb3")
    rgb_t, dens_t = rgb.detach().cpu().requires_grad_(True), dens1.detach().cpu().requires_grad_(True)
    w_t = orc.weights_from_density(t_a.cpu(), dens_t)
    comp = (w_t[..., None] * rgb_t).sum(-2)
    acc_t = w_t.sum(-1, keepdim=True)
    loss_t = ((comp + bg_rays.cpu() * (1.0 - acc_t) - target.cpu()) ** 2).mean()
    loss_t.backward()
    close(rgb_r, comp, atol=1e-6, rtol=0, msg="random background: rgb_out is the plain composite")
    close(sq_r.sum() / (3 * n), loss_t, rtol=1e-5)
    gclose(drgb_r, rgb_t.grad, 1e-5, "d rgb (random background)")
    gclose(dd_r, dens_t.grad, 1e-4, "d density (random background)")
    assert lib.nsamd_render_train(N.ptr(rgb), N.ptr(dens1), N.ptr(t_a), n, S1, 3, None, N.ptr(target), 1.0, N.ptr(w_r),
                                  N.ptr(rgb_r), None, None, None, None, None, None, None, st) == -1  # mode 3 needs bg_rays

    # ---- proposal losses ----
    wp = [torch.rand(n, S0, device="cuda") / S0, torch.rand(n, 64, device="cuda") / 64]
    sp = [s0, torch.sort(torch.rand(n, 65, device="cuda"), dim=-1).values]
    wf = torch.rand(n, S1, device="cuda") / S1
    per_a = [e(n), e(n)]; dwp_a = [e(n, S0), e(n, 64)]; dist_a, dwd_a = e(n), e(n, S1)
    for i in range(2):
        N.check(lib.nsamd_interlevel_loss(N.ptr(s_a), N.ptr(wf), S1, N.ptr(sp[i]), N.ptr(wp[i]), wp[i].shape[1], n, 0.37,
                                          N.ptr(per_a[i]), N.ptr(dwp_a[i]), st), "il")
    N.check(lib.nsamd_distortion_loss(N.ptr(s_a), N.ptr(wf), S1, n, 0.011, N.ptr(dist_a), N.ptr(dwd_a), st), "dl")
    per_b = [e(n), e(n)]; dwp_b = [e(n, S0), e(n, 64)]; dist_b, dwd_b = e(n), e(n, S1)
    parr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    N.check(lib.nsamd_proposal_losses(N.ptr(s_a), N.ptr(wf), S1, 2, parr(sp), parr(wp), (C.c_int32 * 2)(S0, 64), n, 0.37,
                                      0.011, parr(per_b), parr(dwp_b), N.ptr(dist_b), N.ptr(dwd_b), st), "pl")
    for a, b in zip(per_a + dwp_a + [dist_a, dwd_a], per_b + dwp_b + [dist_b, dwd_b]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("S,levels,log2_T,max_res", [(256, 5, 17, 128), (96, 5, 17, 256), (48, 16, 19, 2048)])
def test_table_scatter_binned_equals_scan_path(F, S, levels, log2_T, max_res):
    """The binned two-pass scatter (fine x-pair records, coarse run merging + workgroup combining, pass 2) against the
    scratch-free tile-scan kernel of the same entry point: two independent implementations of dL/dtable on ray-mode
    samples — uniform and strongly concentrated bins, dense and 60 %-zero gradients."""
    from nerfstudio_amd import _native as N

    lib = N.load()
    torch.manual_seed(11)
    n = 257  # ragged
    M = n * S
    spec = F.HashGridSpec(levels, 16, max_res, log2_T)
    o = (torch.randn(n, 3) * 0.5).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1).cuda()
    nears, fars = torch.full((n,), 0.05, device="cuda"), torch.full((n,), 1000.0, device="cuda")
    _, t_uniform = F.piecewise_bins(nears, fars, S, torch.rand(n, device="cuda"))
    centre = torch.rand(n, 1, device="cuda") * 2 + 0.3  # samples bunched within +-1 % of one depth per ray
    t_conc = centre * (1 + 0.02 * (torch.linspace(0, 1, S + 1, device="cuda")[None] - 0.5))
    table = torch.randn(levels << log2_T, 2, device="cuda")
    for t_bins in (t_uniform, t_conc.contiguous()):
        for zero_frac in (0.0, 0.6):
            denc = torch.randn(spec.out_dim, M, device="cuda")
            denc = (denc * (torch.rand(M, device="cuda") >= zero_frac)).contiguous()
            P = N.make_points(None, o, d, t_bins, S)
            ws, ws_n = F._scatter_workspace(spec, torch.device("cuda"), M)
            assert ws_n > 0
            got, ref = torch.zeros_like(table), torch.zeros_like(table)
            N.check(lib.nsamd_hashgrid_encode_bwd(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), N.ptr(denc), 1,
                                                  M, N.ptr(got), None, N.ptr(ws), ws_n, N.stream()), "binned")
            N.check(lib.nsamd_hashgrid_encode_bwd(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), N.ptr(denc), 1,
                                                  M, N.ptr(ref), None, None, 0, N.stream()), "scan")
            scale = float(ref.abs().max())
            assert float((got - ref).abs().max()) <= 2e-5 * scale, (S, zero_frac, float((got - ref).abs().max()), scale)
            assert int((ref != 0).sum()) > 0
            # write-only variant: the gradient buffer starts as garbage and must come back complete
            ws2, ws2_n = F._scatter_workspace(spec, torch.device("cuda"), M, write_only=True)
            got2 = torch.full_like(table, float("nan"))
            N.check(lib.nsamd_hashgrid_encode_bwd_set(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(),
                                                      N.ptr(denc), 1, M, N.ptr(got2), None, N.ptr(ws2), ws2_n, N.stream()),
                    "binned, write-only")
            assert float((got2 - ref).abs().max()) <= 2e-5 * scale
    # degenerate batch: every ray identical -> a handful of tiles receive everything, their queues overflow and the
    # updates go through the direct-atomic (accumulate) / deferred-list (write-only) paths
    o1, d1 = o[:1].expand(n, 3).contiguous(), d[:1].expand(n, 3).contiguous()
    t1 = t_uniform[:1].expand(n, S + 1).contiguous()
    P = N.make_points(None, o1, d1, t1, S)
    denc = torch.randn(spec.out_dim, M, device="cuda")
    ws, ws_n = F._scatter_workspace(spec, torch.device("cuda"), M)
    ws2, ws2_n = F._scatter_workspace(spec, torch.device("cuda"), M, write_only=True)
    got, got2, ref = torch.zeros_like(table), torch.full_like(table, float("nan")), torch.zeros_like(table)
    for fn, out, w, wn in ((lib.nsamd_hashgrid_encode_bwd, got, ws, ws_n), (lib.nsamd_hashgrid_encode_bwd_set, got2, ws2, ws2_n),
                           (lib.nsamd_hashgrid_encode_bwd, ref, None, 0)):
        N.check(fn(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), N.ptr(denc), 1, M, N.ptr(out), None,
                   N.ptr(w) if w is not None else None, wn, N.stream()), "degenerate")
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-4 * scale and float((got2 - ref).abs().max()) <= 1e-4 * scale


