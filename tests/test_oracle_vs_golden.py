"""CPU: pins oracle/nerfacto_oracle.py against (i) the fixtures generated from the reference itself
(tests/golden/make_golden.py), (ii) the seed-free KATs of SURVEY.md §8(c), (iii) the numeric anchors the reference's
own tests hold (tests/utils/test_spherical_harmonics.py:8-16, tests/cameras/test_rays.py:11-30,
tests/model_components/test_renderers.py:12-83)."""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc

T = torch.from_numpy


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def small_cfg(main_log2, prop_log2, num_images):
    return orc.NerfactoCfg(
        main_grid=orc.HashGridCfg(16, 16, 2048, int(main_log2)),
        prop_grids=(orc.HashGridCfg(5, 16, 128, int(prop_log2)), orc.HashGridCfg(5, 16, 256, int(prop_log2))),
        num_images=int(num_images),
    )


# ---------------------------------------------------------------- KATs --------------------------------------------
def test_kat_hash(golden):
    g = golden("kat")
    xyz = g["hash_in"]
    got = [int(orc.hash_corner_index(xyz[i : i + 1, 0], xyz[i : i + 1, 1], xyz[i : i + 1, 2], i, 32)[0]) for i in range(2)]
    assert got == list(g["hash_out"]) == [19, 60]


def test_kat_scalings(golden):
    g = golden("kat")
    for name, (L, lo, hi) in {"main": (16, 16, 2048), "prop0": (5, 16, 128), "prop1": (5, 16, 256)}.items():
        np.testing.assert_array_equal(orc.hash_level_scalings(L, lo, hi).numpy(), g[f"scalings_{name}"])
    assert orc.hash_level_scalings(16, 16, 2048).tolist() == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]


def test_kat_sh_contraction(golden):
    g = golden("kat")
    close(orc.sh_levels4(T(g["sh_in"])), g["sh_out"], atol=1e-7)
    close(orc.contract_linf(T(g["contract_in"])), g["contract_out"], atol=0)
    close(orc.contract_linf(torch.tensor([[2.0, 0, 0], [-4.0, 2, 1]])), [[1.5, 0, 0], [-1.75, 0.875, 0.4375]], atol=0)


def test_kat_sampler_and_render(golden):
    g = golden("kat")
    nears, fars = torch.full((1, 1), 0.05), torch.full((1, 1), 1000.0)
    s, t = orc.piecewise_bins(nears, fars, 4, None)
    close(t[0, :-1], g["pw_starts"], atol=0, rtol=1e-7)
    close(t[0, 1:], g["pw_ends"], atol=0, rtol=1e-7)
    close(t[0, :-1], [0.05, 0.53724998, 1.02511537, 2.04813099], rtol=1e-6)
    w = orc.weights_from_density(t, torch.ones(1, 4))
    close(w[0], g["pw_weights"], atol=1e-7)
    s2, t2, _ = orc.pdf_resample(s, w, 3, None, nears, fars)
    close(t2[0, :-1], g["pdf_starts"], rtol=2e-6)
    close(t2[0, 1:], g["pdf_ends"], rtol=2e-6)
    rgbs = torch.tensor([[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1.0, 1, 1]]])
    close(orc.composite_rgb(rgbs, w), g["rgb_last_sample"][None], atol=1e-7)
    close(orc.depth_median(w, t)[0][0], g["depth_median"], rtol=1e-7)
    close(orc.depth_expected(w, t)[0], g["depth_expected"], rtol=1e-6)


def test_reference_test_anchors(golden):
    # tests/cameras/test_rays.py:11-30 — Frustums.get_positions KAT [0, 3.5, 2] ... here origins=1, dir=(0,1,0), 2..3
    g = golden("kat")
    o = torch.ones(5, 3)
    d = torch.ones(5, 3) * torch.tensor([0.0, 1.0, 0.0])
    t_bins = torch.tensor([[2.0, 3.0]]).expand(5, 2)
    pos = orc.sample_positions(o, d, t_bins)[:, 0]
    close(pos, g["frustum_positions"], atol=0)
    close(pos[0], [1.0, 3.5, 1.0], atol=0)
    # tests/utils/test_spherical_harmonics.py:8-16 — orthonormality of the basis on the sphere, atol 1.5e-2
    torch.manual_seed(0)
    n = 1_000_000
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    sh = orc.sh_levels4(dirs)
    gram = (sh.T @ sh) / n * 4 * np.pi
    close(gram, np.eye(16), atol=1.5e-2, rtol=0)
    # tests/model_components/test_renderers.py — rgb of an opaque white ray > 0.9; zero weights -> background only
    w = torch.zeros(3, 8)
    w[:, 0] = 1.0
    assert float(orc.composite_rgb(torch.ones(3, 8, 3), w, "black").max()) > 0.9
    assert float(orc.composite_rgb(torch.ones(3, 8, 3), torch.zeros(3, 8), "black").abs().max()) == pytest.approx(0)


# ---------------------------------------------------------------- hash grid ---------------------------------------
def test_hashgrid_fixture(golden):
    g = golden("hashgrid")
    L, lo, hi, log2T, F = [int(v) for v in g["cfg"]]
    scal = orc.hash_level_scalings(L, lo, hi)
    np.testing.assert_array_equal(scal.numpy(), g["scalings"])
    x = T(g["x"]).requires_grad_(True)
    table = T(g["table"]).requires_grad_(True)
    out = orc.hashgrid_encode(x, table, scal, 2**log2T)
    close(out, g["out"], atol=1e-6)
    (out * T(g["gout"])).sum().backward()
    close(x.grad, g["dx"], atol=1e-4, rtol=1e-4)
    close(table.grad, g["dtable"], atol=1e-5, rtol=1e-5)


# ---------------------------------------------------------------- fields ------------------------------------------
def _field_setup(g):
    cfg = small_cfg(g["cfg_main_log2"], g["cfg_prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    for v in params.values():
        v.requires_grad_(True)
    return cfg, params


def test_proposal_density_fixture(golden):
    g = golden("fields")
    cfg, params = _field_setup(g)
    for i in range(2):
        pos = T(g["positions"]).requires_grad_(True)
        dens = orc.proposal_density(pos, params, i, cfg)
        close(dens, g[f"prop{i}_density"], atol=1e-6, rtol=2e-5)
        (dens * T(g[f"prop{i}_g"])).sum().backward()
        close(pos.grad, g[f"prop{i}_dpos"], atol=1e-4, rtol=1e-3)
        close(params[f"proposal_networks.{i}.encoding.hash_table"].grad, g[f"prop{i}_dtable"], atol=1e-5, rtol=1e-4)
        for j in range(2):
            close(params[f"proposal_networks.{i}.mlp_base.1.layers.{j}.weight"].grad, g[f"prop{i}_dW{j}"], atol=1e-4, rtol=1e-4)
            close(params[f"proposal_networks.{i}.mlp_base.1.layers.{j}.bias"].grad, g[f"prop{i}_db{j}"], atol=1e-4, rtol=1e-4)


def test_nerfacto_field_fixture(golden):
    g = golden("fields")
    cfg, params = _field_setup(g)
    pos, dirs, cam = T(g["positions"]).requires_grad_(True), T(g["directions"]), T(g["cams"])
    dens, rgb, _ = orc.nerfacto_field(pos, dirs, cam, params, cfg, training=True)
    close(dens, g["main_train_density"], atol=1e-6, rtol=2e-5)
    close(rgb, g["main_train_rgb"], atol=2e-6)
    ((dens * T(g["main_g_density"])).sum() + (rgb * T(g["main_g_rgb"])).sum()).backward()
    close(pos.grad, g["main_dpos"], atol=2e-3, rtol=2e-3)
    close(params["field.mlp_base.model.0.hash_table"].grad, g["main_dtable"], atol=1e-5, rtol=1e-4)
    close(params["field.embedding_appearance.embedding.weight"].grad, g["main_demb"], atol=1e-5, rtol=1e-4)
    for j in range(2):
        close(params[f"field.mlp_base.model.1.layers.{j}.weight"].grad, g[f"main_base_dW{j}"], atol=1e-4, rtol=1e-4)
        close(params[f"field.mlp_base.model.1.layers.{j}.bias"].grad, g[f"main_base_db{j}"], atol=1e-4, rtol=1e-4)
    for j in range(3):
        close(params[f"field.mlp_head.layers.{j}.weight"].grad, g[f"main_head_dW{j}"], atol=1e-4, rtol=1e-4)
        close(params[f"field.mlp_head.layers.{j}.bias"].grad, g[f"main_head_db{j}"], atol=1e-4, rtol=1e-4)
    with torch.no_grad():
        dens_e, rgb_e, _ = orc.nerfacto_field(pos.detach(), dirs, cam, params, cfg, training=False)
    close(dens_e, g["main_eval_density"], atol=1e-6, rtol=2e-5)
    close(rgb_e, g["main_eval_rgb"], atol=2e-6)


# ---------------------------------------------------------------- samplers ----------------------------------------
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_sampler_fixture(golden, mode):
    g = golden("samplers")
    nears, fars = T(g["nears"]), T(g["fars"])
    tr = mode == "train"
    s0, t0 = orc.piecewise_bins(nears, fars, 256, T(g["j0"]) if tr else None)
    np.testing.assert_array_equal(s0.numpy(), g[f"{mode}_l0_s_bins"])  # elementwise IEEE ops only -> bit exact
    np.testing.assert_array_equal(t0.numpy(), g[f"{mode}_l0_t_bins"])
    w0 = orc.weights_from_density(t0, T(g[f"{mode}_l0_density"]))
    close(w0, g[f"{mode}_l0_weights"], atol=1e-7, rtol=1e-6)
    # feed the REFERENCE's weights so the index comparison isolates the resampler
    dbg = {}
    s1, t1, i1 = orc.pdf_resample(s0, T(g[f"{mode}_l0_weights"]), 96, T(g["j1"]) if tr else None, nears, fars, debug=dbg)
    _check_inds(i1, g[f"{mode}_l1_inds"], dbg)
    close(s1, g[f"{mode}_l1_s_bins"], atol=2e-6, rtol=0)
    _check_t_bins(t1, s1, g[f"{mode}_l1_s_bins"], g[f"{mode}_l1_t_bins"], nears, fars)
    w1 = T(g[f"{mode}_l1_weights"])
    close(orc.weights_from_density(T(g[f"{mode}_l1_t_bins"]), T(g[f"{mode}_l1_density"])), w1, atol=1e-7, rtol=1e-6)
    s2, t2, i2 = orc.pdf_resample(
        T(g[f"{mode}_l1_s_bins"]), torch.pow(w1, float(g["anneal"])), 48, T(g["j2"]) if tr else None, nears, fars, debug=dbg
    )
    _check_inds(i2, g[f"{mode}_l2_inds"], dbg)
    close(s2, g[f"{mode}_l2_s_bins"], atol=2e-6, rtol=0)
    _check_t_bins(t2, s2, g[f"{mode}_l2_s_bins"], g[f"{mode}_l2_t_bins"], nears, fars)


def _check_t_bins(t_mine, s_mine, s_ref, t_ref, nears, fars):
    """t = 1/(2-2s') is ill-conditioned as s'->1 (far field): an ulp of s moves t by t*ulp/(1-s'). So (i) mapping the
    REFERENCE's s_bins through the oracle's s->t map must be bit-exact, (ii) the oracle's own t may deviate only by
    what its s deviation explains."""
    np.testing.assert_array_equal(orc.spacing_to_euclidean(T(s_ref), nears, fars).numpy(), t_ref)
    ds = np.abs(s_mine.numpy() - s_ref)
    s_far = orc.spacing_fn(fars).numpy()
    cond = 2.0 * np.maximum(t_ref, 1.0) ** 2 * s_far  # |dt/ds| for the far branch t = 1/(2-2x), x = s*s_far+(1-s)*s_near
    assert np.all(np.abs(t_mine.numpy() - t_ref) <= cond * ds * 1.5 + 1e-6 * np.abs(t_ref))


def _check_inds(mine, ref, dbg):
    """Index equality with the reference. The oracle sums the weights left-to-right where the reference uses
    torch.sum (ATen's blocked order), so cdf values can differ by an ulp; an index may differ ONLY where u ties with
    every cdf entry between the two answers to within 2 ulp (on the fixtures: the eval-mode u = 0.5 of the two
    degenerate rays, whose cdf hits 0.5 exactly). Anything else fails."""
    mine = mine.numpy()
    bad = np.argwhere(mine != ref)
    assert len(bad) <= 4, f"{len(bad)} searchsorted indices differ from the reference"
    cdf, u = dbg["cdf"].numpy(), dbg["u"].numpy()
    for r, c in bad:
        lo, hi = sorted((int(mine[r, c]), int(ref[r, c])))
        gap = np.abs(cdf[r, lo:hi] - u[r, c])
        assert np.all(gap <= 2 * np.spacing(np.float32(u[r, c]))), (r, c, mine[r, c], ref[r, c], gap)


def test_weights_backward_fixture(golden):
    g = golden("samplers")
    dens = T(g["train_l0_density"]).requires_grad_(True)
    w = orc.weights_from_density(T(g["train_l0_t_bins"]), dens)
    (w * T(g["weights_g"])).sum().backward()
    close(dens.grad, g["weights_ddensity"], atol=1e-5, rtol=1e-4)


# ---------------------------------------------------------------- render / losses ---------------------------------
def test_render_fixture(golden):
    g = golden("render")
    t_bins = T(g["t_bins"])
    dens = T(g["density"]).requires_grad_(True)
    rgb = T(g["rgb"]).requires_grad_(True)
    w = orc.weights_from_density(t_bins, dens)
    close(w, g["weights"], atol=1e-7, rtol=1e-6)
    for bg in ("last_sample", "white", "black", "random"):
        close(orc.composite_rgb(rgb, w, bg, True), g[f"rgb_train_{bg}"], atol=1e-6)
    close(orc.composite_rgb(T(g["rgb_eval_in"]), w.detach(), "last_sample", False), g["rgb_eval_last_sample"], atol=1e-6)
    close(orc.accumulation(w), g["accumulation"], atol=1e-6)
    dm, idx = orc.depth_median(w.detach(), t_bins)
    np.testing.assert_array_equal(idx.numpy(), g["depth_median_idx"])
    close(dm, g["depth_median"], atol=0, rtol=1e-7)
    de = orc.depth_expected(w, t_bins)
    close(de, g["depth_expected"], rtol=1e-5)
    comp = orc.composite_rgb(rgb, w, "last_sample", True)
    ((comp * T(g["g_rgb"])).sum() + (orc.accumulation(w) * T(g["g_acc"])).sum() + (de * T(g["g_dep"])).sum()).backward()
    close(dens.grad, g["d_density"], atol=1e-4, rtol=1e-4)
    close(rgb.grad, g["d_rgb"], atol=1e-6, rtol=1e-5)


def test_losses_fixture(golden):
    g = golden("losses")
    ws = [T(g[f"w{i}"]).requires_grad_(True) for i in range(3)]
    bins = [T(g[f"s_bins{i}"]) for i in range(3)]
    li = orc.interlevel_loss(ws, bins)
    ld = orc.distortion_loss(ws[-1], bins[-1])
    close(li, g["interlevel"], rtol=1e-5)
    close(ld, g["distortion"], rtol=1e-5)
    (li + 0.5 * ld).backward()
    for i in range(3):
        close(ws[i].grad, g[f"dw{i}"], atol=1e-7, rtol=1e-4)


# ---------------------------------------------------------------- whole pipeline ----------------------------------
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pipeline_fixture(golden, mode):
    g = golden("pipeline")
    cfg = small_cfg(g["main_log2"], g["prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    for v in params.values():
        v.requires_grad_(True)
    jit = [T(g[f"j{i}"]) for i in range(3)]
    out = orc.nerfacto_forward(params, cfg, T(g["origins"]), T(g["directions"]), T(g["cams"]), jit, training=(mode == "train"))
    for i in range(3):
        close(out["s_bins_list"][i], g[f"{mode}_s_bins{i}"], atol=5e-6, rtol=0)
        close(out["t_bins_list"][i], g[f"{mode}_t_bins{i}"], atol=0, rtol=2e-3)  # far-field conditioning, see _check_t_bins
        close(out["weights_list"][i], g[f"{mode}_w{i}"], atol=2e-5, rtol=1e-3)
    for k, (mine, ref) in enumerate(zip(out["inds_list"], (g[f"{mode}_inds1"], g[f"{mode}_inds2"]))):
        frac = float((mine.numpy() != ref).mean())
        assert frac <= 2e-3, f"level {k + 1}: {frac:.2%} of indices differ (end-to-end, ulp-level weight noise)"
    close(out["rgb"], g[f"{mode}_rgb"], atol=1e-5, rtol=0)
    close(out["accumulation"], g[f"{mode}_acc"], atol=1e-5, rtol=0)
    close(out["expected_depth"], g[f"{mode}_expected_depth"], rtol=1e-4)
    close(out["depth"], g[f"{mode}_depth"], rtol=1e-4)
    for i in range(2):
        close(out[f"prop_depth_{i}"], g[f"{mode}_prop_depth_{i}"], rtol=1e-4)
    if mode == "train":
        losses = orc.nerfacto_losses(out, T(g["target"]), cfg)
        close(losses["rgb_loss"], g["loss_rgb"], rtol=1e-5)
        close(losses["interlevel_loss"], g["loss_interlevel"], rtol=1e-4)
        close(losses["distortion_loss"], g["loss_distortion"], rtol=1e-4)
        sum(losses.values()).backward()

        def gclose(name, ref, scale=1.0):
            a, b = params[name].grad.numpy(), g[ref]
            tol = 1e-4 * max(1e-12, float(np.abs(b).max()))
            np.testing.assert_allclose(a, b, atol=tol, rtol=1e-3)

        gclose("field.mlp_base.model.0.hash_table", "g_main_table")
        gclose("field.embedding_appearance.embedding.weight", "g_emb")
        for j in range(2):
            gclose(f"field.mlp_base.model.1.layers.{j}.weight", f"g_base_W{j}")
            gclose(f"field.mlp_base.model.1.layers.{j}.bias", f"g_base_b{j}")
        for j in range(3):
            gclose(f"field.mlp_head.layers.{j}.weight", f"g_head_W{j}")
            gclose(f"field.mlp_head.layers.{j}.bias", f"g_head_b{j}")
        for i in range(2):
            gclose(f"proposal_networks.{i}.encoding.hash_table", f"g_prop{i}_table")
            for j in range(2):
                gclose(f"proposal_networks.{i}.mlp_base.1.layers.{j}.weight", f"g_prop{i}_W{j}")
                gclose(f"proposal_networks.{i}.mlp_base.1.layers.{j}.bias", f"g_prop{i}_b{j}")


# ---------------------------------------------------------------- raygen ------------------------------------------
def test_raygen_fixture(golden):
    g = golden("raygen")
    out = orc.raygen_pinhole(T(g["ray_indices"]), T(g["c2w"]), T(g["fx"]), T(g["fy"]), T(g["cx"]), T(g["cy"]))
    close(out["origins"], g["origins"], atol=0)
    close(out["directions"], g["directions"], atol=1e-7)
    close(out["pixel_area"], g["pixel_area"], rtol=1e-4)
    close(out["directions_norm"], g["directions_norm"], rtol=1e-6)


# ---------------------------------------------------------------- C restatement -----------------------------------
def _c_oracle():
    import ctypes
    import os
    import subprocess

    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    so = os.path.join(here, "liboracle_c.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", here], check=True)
    return ctypes.CDLL(so)


def test_c_oracle_hash_and_cdf(golden):
    """oracle/hash_oracle.c (plain C, integer / ordering-critical pieces) == numpy/torch oracle == reference."""
    import ctypes as C

    lib = _c_oracle()
    g = golden("hashgrid")
    L, lo, hi, log2T, _ = [int(v) for v in g["cfg"]]
    x = np.ascontiguousarray(g["x"], dtype=np.float32)
    M = x.shape[0]
    scal = orc.hash_level_scalings(L, lo, hi).numpy()
    for lvl in range(L):
        out = np.zeros((M, 8), dtype=np.int64)
        lib.oracle_hash_corner_indices(x.ctypes.data_as(C.c_void_p), C.c_int64(M), C.c_float(float(scal[lvl])), lvl, log2T,
                                       out.ctypes.data_as(C.c_void_p))
        s = x * scal[lvl]
        f, c = np.floor(s).astype(np.int32), np.ceil(s).astype(np.int32)
        for corner in range(8):
            ix = (c if corner & 1 else f)[:, 0]
            iy = (c if corner & 2 else f)[:, 1]
            iz = (c if corner & 4 else f)[:, 2]
            np.testing.assert_array_equal(out[:, corner], orc.hash_corner_index(ix, iy, iz, lvl, 2**log2T))
    # KAT of SURVEY §8c through the C path: corners of (3,7,11) at level 0 and (1,2,3) at level 1, T = 32
    pts = np.array([[3, 7, 11], [1, 2, 3]], dtype=np.float32)
    out = np.zeros((2, 8), dtype=np.int64)
    lib.oracle_hash_corner_indices(pts.ctypes.data_as(C.c_void_p), C.c_int64(2), C.c_float(1.0), 0, 5, out.ctypes.data_as(C.c_void_p))
    assert out[0, 0] == 19 and out[1, 0] + 32 == 60

    gs = golden("samplers")
    w = np.ascontiguousarray(gs["train_l0_weights"], dtype=np.float32)
    n, S = w.shape
    cdf = np.zeros((n, S + 1), dtype=np.float32)
    lib.oracle_pdf_cdf(w.ctypes.data_as(C.c_void_p), C.c_int64(n), S, C.c_float(0.01), C.c_float(1e-5), cdf.ctypes.data_as(C.c_void_p))
    dbg = {}
    nears, fars = T(gs["nears"]), T(gs["fars"])
    _, _, inds = orc.pdf_resample(T(gs["train_l0_s_bins"]), T(w), 96, T(gs["j1"]), nears, fars, debug=dbg)
    np.testing.assert_array_equal(cdf, dbg["cdf"].numpy())  # same left-to-right fp32 arithmetic, bit for bit
    u = np.ascontiguousarray(dbg["u"].numpy(), dtype=np.float32)
    idx = np.zeros((n, 97), dtype=np.int32)
    lib.oracle_searchsorted_right(cdf.ctypes.data_as(C.c_void_p), C.c_int64(n), S + 1, u.ctypes.data_as(C.c_void_p), 97,
                                  idx.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(idx, inds.numpy())
    np.testing.assert_array_equal(idx, gs["train_l1_inds"])  # and equal to the reference's own indices


def test_psnr_fixture_is_reproducible_from_the_scene_definition(golden):
    """tests/golden/psnr_scene_s*.npz (oracle training runs on the procedural scene, eight seeds): the stored PSNRs are
    those of the stored images against the ground truth recomputed from tests/psnr_scene.py, the batch streams are
    deterministic and differ between seeds, and the oracle itself reaches the reference's acceptance level (PSNR > 20 dB,
    tests/test_nerfacto_integration.py:71) on training AND held-out views."""
    import numpy as np
    import psnr_scene as S

    twin_tr, twin_ho = [], []
    for seed in S.SEEDS:
        g = golden(f"psnr_scene_s{seed}")
        assert g["losses"].shape == (S.STEPS, 3) and g["losses"][-1].sum() < 0.02 * g["losses"][0].sum()
        assert int(g["cfg"][2]) == 41 + seed
        for k, cam_id in enumerate(S.EVAL_CAMERAS):
            _, _, gt = S.full_view(cam_id)
            assert abs(S.psnr(g["images"][k], gt) - float(g["psnr"][k])) < 1e-6
            assert abs(float(g["psnr_views"][cam_id]) - float(g["psnr"][k])) < 1e-9
        tr, ho = slice(0, S.N_TRAIN), slice(S.N_TRAIN, None)
        assert g["psnr_views"].shape == (S.N_TRAIN + S.N_HELD_OUT,)
        assert g["psnr_views"][tr].mean() > 30 and g["psnr_views"][ho].mean() > 20  # the reference's acceptance level
        # the twin run (1e-6 perturbation of the initial tables) — the spread of two correct fp32 trainings of this problem, what
        # an implementation difference is judged by: over the eight seeds the means over views differ by up to 0.30 dB (training;
        # s.d. 0.16 dB) and 0.65 dB (held out; s.d. 0.25 dB) per seed, single views by dB
        twin_tr.append(g["psnr_views_twin"][tr].mean() - g["psnr_views"][tr].mean())
        twin_ho.append(g["psnr_views_twin"][ho].mean() - g["psnr_views"][ho].mean())
        assert abs(twin_tr[-1]) < 0.4 and abs(twin_ho[-1]) < 0.8, (seed, twin_tr[-1], twin_ho[-1])
        assert np.abs(g["psnr_views"] - g["psnr_views_twin"]).max() > 0.5
    assert abs(np.mean(twin_tr)) < 0.15 and abs(np.mean(twin_ho)) < 0.25, (twin_tr, twin_ho)  # measured -0.06 / -0.14 dB
    b1, b2, b3 = S.batches(seed=9, steps=2), S.batches(seed=9, steps=2), S.batches(seed=10, steps=2)
    for x, y, z in zip(b1, b2, b3):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
        assert not np.array_equal(x[0], z[0])


@pytest.mark.parametrize("mode", ["SO3xR3", "SE3"])
def test_camera_optimizer_gradient_fixture(golden, mode):
    """with nerfacto's default camera optimiser the ray origins / directions are functions of
    `pose_adjustment` and the loss gradient reaches it through the sample positions of all three levels. The mirror
    (nerfstudio_amd/cameras/camera_optimizers.py, lie_groups.py — host torch, runs on CPU) + the oracle must reproduce
    the reference's corrected rays, losses, dL/d(origins, directions) and dL/dpose_adjustment (tests/golden/camera_opt.npz,
    generated from the reference's own CameraOptimizer)."""
    from nerfstudio_amd.cameras.camera_optimizers import CameraOptimizerConfig

    g = golden("camera_opt")
    cfg = small_cfg(g["main_log2"], g["prop_log2"], g["num_images"])
    params = orc.init_params(cfg, seed=int(g["seed"]), table_std=float(g["table_std"]))
    for v in params.values():
        v.requires_grad_(True)
    cam_opt = CameraOptimizerConfig(mode=mode).setup(num_cameras=cfg.num_images, device="cpu")
    with torch.no_grad():
        cam_opt.pose_adjustment.copy_(T(g["pose_adjustment"]))
    o, d = cam_opt.corrected_rays(T(g["origins"]), T(g["directions"]), T(g["cams"]))
    close(o, g[f"{mode}_origins"], atol=1e-7, rtol=1e-6)
    close(d, g[f"{mode}_directions"], atol=1e-7, rtol=1e-6)
    o.retain_grad()
    d.retain_grad()
    out = orc.nerfacto_forward(params, cfg, o, d, T(g["cams"]), [T(g[f"j{i}"]) for i in range(3)], training=True)
    losses = orc.nerfacto_losses(out, T(g["target"]), cfg)
    cam_opt.get_loss_dict(losses)
    close(out["rgb"], g[f"{mode}_rgb"], atol=1e-5, rtol=0)
    ref_l = g[f"{mode}_losses"]
    for k, name in enumerate(("rgb_loss", "interlevel_loss", "distortion_loss", "camera_opt_regularizer")):
        close(losses[name], ref_l[k], rtol=2e-4)
    sum(losses.values()).backward()

    def gclose(a, b, rel_l2, what):
        """Relative L2: the per-ray position gradients inherit the far-field conditioning of the sample positions (t up
        to 1000 through spacing_to_euclidean, see _check_t_bins: the oracle's t_bins agree with the reference's to 2e-3
        there), so a handful of far rays differ by ~1 % of the largest entry while everything else agrees to 1e-4."""
        err = float(np.linalg.norm(a.numpy() - b) / max(1e-30, np.linalg.norm(b)))
        assert err <= rel_l2, f"{what}: relative L2 {err:.2e}"

    gclose(o.grad, g[f"{mode}_d_origins"], 2e-2, "dL/d origins")
    gclose(d.grad, g[f"{mode}_d_directions"], 1e-2, "dL/d directions")
    gclose(cam_opt.pose_adjustment.grad, g[f"{mode}_g_pose"], 1e-2, "dL/d pose_adjustment")
    gclose(params["field.mlp_base.model.0.hash_table"].grad, g[f"{mode}_g_main_table"], 1e-3, "dL/d main table")
    # module contract (camera_optimizers.py:115-205)
    assert cam_opt(torch.tensor([0, 3])).shape == (2, 3, 4) and cam_opt.get_correction_matrices().shape == (cfg.num_images, 3, 4)
    groups = {}
    cam_opt.get_param_groups(groups)
    assert list(groups) == ["camera_opt"] and groups["camera_opt"][0] is cam_opt.pose_adjustment
    off = CameraOptimizerConfig(mode="off").setup(num_cameras=3, device="cpu")
    assert torch.equal(off(torch.tensor([1]))[0], torch.eye(4)[:3]) and off.corrected_rays(o, d, T(g["cams"]))[0] is o


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[0]: vanilla NeRF (oracle/vanilla_oracle.py) against the reference's own modules
# ---------------------------------------------------------------------------------------------------------------------
def test_vanilla_nerf_encoding_matches_reference(golden):
    from oracle import vanilla_oracle as van

    g = golden("vanilla")
    x = torch.from_numpy(g["enc_x"])
    for name, args in (("enc_pos", (10, 0.0, 8.0, True)), ("enc_dir", (4, 0.0, 4.0, True)), ("enc_plain", (6, 0.0, 5.0, False))):
        got = van.nerf_encoding(x, *args)
        assert got.shape == g[name].shape
        np.testing.assert_array_equal(got.numpy(), g[name])  # same torch ops in the same order: bit-equal


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_vanilla_nerf_pipeline_matches_reference(golden, mode):
    """NeRFModel.get_outputs wiring (uniform 64 -> coarse field -> PDF 128 + original edges -> fine field, white
    background, median depth), per-edge jitter (single_jitter=False, the samplers' default), losses and gradients."""
    from oracle import vanilla_oracle as van

    g = golden("vanilla")
    cfg = van.VanillaCfg()
    params = {}
    params.update(van.init_field_params(cfg, 92, "field_coarse."))
    params.update(van.init_field_params(cfg, 93, "field_fine."))
    np.testing.assert_allclose(np.array([float(v.double().sum()) for v in params.values()]), g["param_checksum"], rtol=1e-12)
    training = mode == "train"
    for p in params.values():
        p.requires_grad_(training)
    o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
    jit = [torch.from_numpy(g["j0"]), torch.from_numpy(g["j1"])] if training else None
    out = van.vanilla_forward(params, cfg, o, d, jit, training=training)
    pre = mode + "_"
    np.testing.assert_array_equal(out["t_bins_coarse"].numpy(), g[pre + "t_bins_coarse"])
    np.testing.assert_allclose(out["weights_coarse"].detach().numpy(), g[pre + "weights_coarse"], rtol=2e-4, atol=1e-7)
    assert out["s_bins_fine"].shape == (o.shape[0], 64 + 128 + 2)
    np.testing.assert_allclose(out["s_bins_fine"].numpy(), g[pre + "s_bins_fine"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["t_bins_fine"].numpy(), g[pre + "t_bins_fine"], rtol=0, atol=1e-5)
    for k in ("rgb_coarse", "rgb_fine", "accumulation_coarse", "accumulation_fine"):
        np.testing.assert_allclose(out[k].detach().numpy(), g[pre + k], rtol=0, atol=2e-6, err_msg=k)
    for k in ("depth_coarse", "depth_fine"):
        np.testing.assert_allclose(out[k].detach().numpy(), g[pre + k], rtol=2e-5, err_msg=k)
    if training:
        tgt = torch.from_numpy(g["target"])
        loss = sum(van.vanilla_losses(out, tgt, cfg).values())
        np.testing.assert_allclose(float(loss), float(g["train_loss"]), rtol=1e-5)
        loss.backward()
        for name, p in params.items():
            gr = p.grad.reshape(-1)
            idx = torch.from_numpy(g[f"train_gidx_{name}"])
            stat = g[f"train_gstat_{name}"]
            scale = max(float(stat[0]), 1e-30)
            ref_vals = g[f"train_gval_{name}"]
            atol = 3e-2 * max(float(np.abs(ref_vals).max()), scale / np.sqrt(gr.numel())) + 1e-12  # cancellation-heavy sums
            np.testing.assert_allclose(gr[idx].numpy(), ref_vals, rtol=0, atol=atol, err_msg=name)
            # (256-wide matmuls: MKL's blocking order differs between the two compositions of the same graph)
            np.testing.assert_allclose(float(gr.double().norm()), float(stat[0]), rtol=2e-3, err_msg=name)


# ---------------------------------------------------------------------------------------------------------------------
# instant-ngp packed arithmetic (oracle/packed_oracle.py): anchored on the pinned DENSE path
# ---------------------------------------------------------------------------------------------------------------------
def test_packed_weights_and_compositing_equal_the_dense_reference_path(golden):
    """With the same number of samples on every ray, packing is a reshape: render_weight_from_density / the packed
    renderer branches must reproduce the reference's dense get_weights / renderers (fixtures samplers.npz, render.npz)."""
    from oracle import packed_oracle as pk

    g = golden("render")
    w_ref = torch.from_numpy(g["weights"])                     # reference RaySamples.get_weights on ...
    t_bins, dens = torch.from_numpy(g["t_bins"]), torch.from_numpy(g["density"])
    n, s = dens.shape
    ray_idx = torch.arange(n).repeat_interleave(s)
    t0, t1 = t_bins[:, :-1].reshape(-1), t_bins[:, 1:].reshape(-1)
    w, trans, alphas = pk.render_weight_from_density(t0, t1, dens.reshape(-1), ray_idx, n)
    np.testing.assert_allclose(w.view(n, s).numpy(), w_ref.numpy(), rtol=2e-5, atol=1e-7)
    info = pk.pack_info(ray_idx, n)
    assert torch.equal(info[:, 1], torch.full((n,), s)) and torch.equal(info[:, 0], torch.arange(n) * s)
    rgb = torch.from_numpy(g["rgb"])
    for bg in ("white", "black"):
        comp, acc, depth = pk.composite_packed(rgb.reshape(-1, 3), w_ref.reshape(-1), t0, t1, ray_idx, n, background=bg)
        dense = orc.composite_rgb(rgb, w_ref, bg, training=True)
        np.testing.assert_allclose(comp.numpy(), dense.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(acc.numpy(), orc.accumulation(w_ref).numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(depth.numpy(), orc.depth_expected(w_ref, t_bins).numpy(), rtol=2e-5)
    np.testing.assert_allclose(acc.numpy(), g["accumulation"], rtol=0, atol=2e-6)  # and the reference's own renderer output


def test_packed_ragged_rays_and_visibility():
    """Ragged packing (rays with 0 .. many samples): per-ray results equal the dense formulas applied ray by ray; the
    visibility mask follows its definition (transmittance before the sample >= eps, alpha >= threshold)."""
    from oracle import packed_oracle as pk

    rs = np.random.RandomState(17)
    counts = [0, 5, 1, 0, 12, 3]
    ray_idx = torch.tensor([r for r, c in enumerate(counts) for _ in range(c)])
    m = int(ray_idx.numel())
    t0 = torch.from_numpy(np.concatenate([np.sort(rs.uniform(0.1, 4, c)) for c in counts if c]).astype(np.float32))
    t1 = t0 + torch.from_numpy(rs.uniform(0.01, 0.2, m).astype(np.float32))
    sig = torch.from_numpy((np.exp(rs.standard_normal(m) * 2)).astype(np.float32))
    w, trans, alphas = pk.render_weight_from_density(t0, t1, sig, ray_idx, len(counts))
    off = 0
    for r, c in enumerate(counts):
        if c == 0:
            continue
        sd = sig[off:off + c] * (t1[off:off + c] - t0[off:off + c])
        excl = torch.cumsum(sd, 0) - sd
        np.testing.assert_allclose(trans[off:off + c].numpy(), torch.exp(-excl).numpy(), rtol=1e-5)
        np.testing.assert_allclose(w[off:off + c].numpy(), (torch.exp(-excl) * (1 - torch.exp(-sd))).numpy(), rtol=1e-5, atol=1e-8)
        off += c
    acc = pk.accumulate_along_rays(w, None, ray_idx, len(counts))
    assert float(acc[0]) == 0.0 and float(acc[3]) == 0.0 and float(acc.max()) <= 1.0 + 1e-6
    vis = pk.render_visibility_from_density(t0, t1, sig, ray_idx, len(counts), early_stop_eps=1e-2, alpha_thre=0.05)
    assert torch.equal(vis, (trans >= 1e-2) & (alphas >= 0.05))
    assert torch.equal(pk.pack_info(ray_idx, len(counts))[:, 1], torch.tensor(counts))
