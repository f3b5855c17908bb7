"""GPU parity tests of the instant-ngp packed path (csrc/packed.hip) against oracle/packed_oracle.py.

Pinned through the oracle to the reference: packed weights / visibility / accumulation (their formulas are those of the
reference's dense path, fixtures render.npz / samplers.npz; the packed-vs-dense consistency is asserted here too).
Pinned only to the restatement: the marcher's sample placement (nerfacc 0.5.2 is not installable here) — bit-exact
indices and bin edges against oracle.occgrid_march, which is what "bit-exact for sample indices / occupancy masks" can
mean without nerfacc."""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc
from oracle import packed_oracle as po

pytestmark = pytest.mark.gpu
ROI = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]


def dev(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.cuda()


@pytest.fixture(scope="module")
def F():
    from nerfstudio_amd import _native, functional

    _native.load()
    return functional


def _rays(n, seed, scale=0.6):
    rs = np.random.RandomState(seed)
    o = (rs.standard_normal((n, 3)) * scale).astype(np.float32)
    d = rs.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("levels,res,cone,stratified", [(1, 16, 0.0, False), (3, 8, 0.0, True), (4, 16, 0.004, True), (2, 5, 0.02, False)])
def test_occgrid_march_bit_exact_vs_oracle(F, levels, res, cone, stratified):
    rs = np.random.RandomState(levels * 100 + res)
    B = rs.rand(levels, res, res, res) > 0.65
    n = 203
    o, d = _rays(n, 11 + levels, scale=1.2)
    d[0] = [0.0, 0.0, 1.0]                    # axis-parallel rays (infinite slab distances)
    d[1] = [1.0, 0.0, 0.0]
    o[2] = [5.0, 5.0, 5.0]                    # outside every level, pointing away or in
    jit = rs.uniform(0, 1, n).astype(np.float32) if stratified else None
    t_min = (rs.uniform(0, 0.3, n)).astype(np.float32)
    t_max = (rs.uniform(1.0, 9.0, n)).astype(np.float32)
    step = 0.031
    ref = po.occgrid_march(o, d, B, ROI, step, near_plane=0.05, far_plane=100.0, t_min=t_min, t_max=t_max, cone_angle=cone,
                           jitter=jit)
    got = F.occgrid_march(dev(o), dev(d), dev(B.astype(np.uint8)), ROI, step, 0.05, 100.0, dev(t_min), dev(t_max), cone,
                          None if jit is None else dev(jit))
    assert len(ref[0]) > 500
    np.testing.assert_array_equal(got[0].cpu().numpy(), ref[0])       # ray indices
    np.testing.assert_array_equal(got[1].cpu().numpy(), ref[1])       # bin edges: same fp32 operations in the same order
    np.testing.assert_array_equal(got[2].cpu().numpy(), ref[2])
    info = got[3].cpu().numpy()
    np.testing.assert_array_equal(info[:, 1], np.bincount(ref[0], minlength=n))
    np.testing.assert_array_equal(info[:, 0], np.cumsum(info[:, 1]) - info[:, 1])
    # no near/far arrays, empty grid, zero rays
    got2 = F.occgrid_march(dev(o), dev(d), dev(B.astype(np.uint8)), ROI, step, 0.2, 3.0, None, None, cone, None)
    ref2 = po.occgrid_march(o, d, B, ROI, step, near_plane=0.2, far_plane=3.0, cone_angle=cone)
    np.testing.assert_array_equal(got2[1].cpu().numpy(), ref2[1])
    empty = F.occgrid_march(dev(o), dev(d), torch.zeros((levels, res, res, res), dtype=torch.uint8, device="cuda"), ROI, step)
    assert empty[0].numel() == 0 and int(empty[3][:, 1].sum()) == 0


def _packed_case(seed, n_rays=97, max_count=300):
    rs = np.random.RandomState(seed)
    counts = rs.randint(0, max_count, n_rays)
    counts[::7] = 0                      # empty rays
    counts[3] = 64                       # exactly one / two wavefront chunks
    counts[4] = 128
    counts[5] = 1
    idx = np.repeat(np.arange(n_rays), counts).astype(np.int64)
    n = len(idx)
    ts = np.concatenate([np.sort(rs.uniform(0.05, 6.0, c)) for c in counts]).astype(np.float32) if n else np.zeros(0, np.float32)
    te = (ts + rs.uniform(0.005, 0.05, n)).astype(np.float32)
    sig = (rs.lognormal(0.0, 1.5, n)).astype(np.float32)
    return n_rays, counts, idx, ts, te, sig


def test_packed_weights_forward_backward_vs_oracle(F):
    n_rays, counts, idx, ts, te, sig = _packed_case(1)
    info, total = F.packed_info_from_counts(dev(counts.astype(np.int32)))
    assert total == len(idx)
    np.testing.assert_array_equal(info.cpu().numpy(), po.pack_info(torch.from_numpy(idx), n_rays).numpy())
    s_ref = torch.from_numpy(sig).requires_grad_(True)
    w_ref, _, _ = po.render_weight_from_density(torch.from_numpy(ts), torch.from_numpy(te), s_ref, torch.from_numpy(idx), n_rays)
    s_gpu = dev(sig).requires_grad_(True)
    w = F.packed_weights(s_gpu, dev(ts), dev(te), info)
    np.testing.assert_allclose(w.detach().cpu().numpy(), w_ref.detach().numpy(), atol=2e-7, rtol=2e-5)
    g = torch.from_numpy(np.random.RandomState(2).standard_normal(len(idx)).astype(np.float32))
    (w_ref * g).sum().backward()
    (w * g.cuda()).sum().backward()
    ref_g = s_ref.grad.numpy()
    np.testing.assert_allclose(s_gpu.grad.cpu().numpy(), ref_g, atol=1e-5 * np.abs(ref_g).max(), rtol=2e-3)


def test_packed_weights_equal_the_dense_kernel_on_equal_counts(F):
    """Anchor on the pinned dense path: with S samples on every ray the packed scan is RaySamples.get_weights."""
    rs = np.random.RandomState(5)
    n, S = 64, 48
    t_bins = np.sort(rs.uniform(0.05, 5.0, (n, S + 1)), axis=1).astype(np.float32)
    dens = rs.lognormal(0, 1.2, (n, S)).astype(np.float32)
    dense = F.weights_from_density(dev(t_bins), dev(dens))
    info, _ = F.packed_info_from_counts(torch.full((n,), S, dtype=torch.int32, device="cuda"))
    packed = F.packed_weights(dev(dens.reshape(-1)), dev(np.ascontiguousarray(t_bins[:, :-1]).reshape(-1)),
                              dev(np.ascontiguousarray(t_bins[:, 1:]).reshape(-1)), info)
    np.testing.assert_array_equal(packed.cpu().numpy().reshape(n, S), dense.cpu().numpy())  # same scan, same bits


def test_visibility_early_termination_and_compaction_vs_oracle(F):
    n_rays, counts, idx, ts, te, sig = _packed_case(7, max_count=400)
    sig = sig * 3.0  # dense enough that many rays terminate early
    info, _ = F.packed_info_from_counts(dev(counts.astype(np.int32)))
    eps, athre = 1e-4, 0.01
    _, trans, alphas = po.render_weight_from_density(torch.from_numpy(ts), torch.from_numpy(te), torch.from_numpy(sig),
                                                     torch.from_numpy(idx), n_rays)
    ref_mask = ((trans >= eps) & (alphas >= athre)).numpy()
    ri, s2, e2, info2, mask = F.packed_visibility_compact(dev(idx), dev(ts), dev(te), dev(sig), info, eps, athre)
    mask = mask.cpu().numpy().astype(bool)
    # identical except exactly at a threshold (device expf vs libm: 1 ulp)
    near = (np.abs(trans.numpy() - eps) < 1e-9) | (np.abs(alphas.numpy() - athre) < 1e-7)
    assert (mask == ref_mask)[~near].all() and 0.05 < mask.mean() < 0.95
    assert (trans.numpy() < eps).sum() > 100, "the case must exercise early termination"
    np.testing.assert_array_equal(ri.cpu().numpy(), idx[mask])        # survivors in order, per ray
    np.testing.assert_array_equal(s2.cpu().numpy(), ts[mask])
    np.testing.assert_array_equal(e2.cpu().numpy(), te[mask])
    np.testing.assert_array_equal(info2.cpu().numpy()[:, 1], np.bincount(idx[mask], minlength=n_rays))


@pytest.mark.parametrize("background", ["random", "white", "black"])
def test_packed_composite_forward_backward_vs_oracle(F, background):
    n_rays, counts, idx, ts, te, sig = _packed_case(9)
    rs = np.random.RandomState(10)
    rgb = rs.uniform(0, 1, (len(idx), 3)).astype(np.float32)
    info, _ = F.packed_info_from_counts(dev(counts.astype(np.int32)))
    w_ref = po.render_weight_from_density(torch.from_numpy(ts), torch.from_numpy(te), torch.from_numpy(sig),
                                          torch.from_numpy(idx), n_rays)[0].requires_grad_(True)
    c_ref = torch.from_numpy(rgb).requires_grad_(True)
    comp_r, acc_r, dep_r = po.composite_packed(c_ref, w_ref, torch.from_numpy(ts), torch.from_numpy(te), torch.from_numpy(idx),
                                               n_rays, background=background, training=True)
    w = dev(w_ref.detach().numpy()).requires_grad_(True)
    c = dev(rgb).requires_grad_(True)
    comp, acc, dep = F.packed_composite(c, w, dev(idx), info, dev(ts), dev(te), background)
    np.testing.assert_allclose(comp.detach().cpu().numpy(), comp_r.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(acc.detach().cpu().numpy(), acc_r.detach().numpy()[:, 0], atol=3e-6)
    steps = (torch.from_numpy(ts) + torch.from_numpy(te)) / 2
    dep_c = torch.clip(dep.detach().cpu(), steps.min(), steps.max())
    np.testing.assert_allclose(dep_c.numpy(), dep_r.detach().numpy()[:, 0], rtol=2e-5, atol=1e-6)
    g_rgb = torch.from_numpy(rs.standard_normal((n_rays, 3)).astype(np.float32))
    g_acc = torch.from_numpy(rs.standard_normal(n_rays).astype(np.float32))
    ((comp_r * g_rgb).sum() + (acc_r[:, 0] * g_acc).sum()).backward()
    ((comp * g_rgb.cuda()).sum() + (acc * g_acc.cuda()).sum()).backward()
    np.testing.assert_allclose(c.grad.cpu().numpy(), c_ref.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(w.grad.cpu().numpy(), w_ref.grad.numpy(), atol=2e-6, rtol=1e-5)
    # eval mode: nan_to_num on the colours, clamp of the result (renderers.py:225-231)
    bad = rgb.copy()
    bad[::11] = np.nan
    with torch.no_grad():
        ev = F.packed_composite(dev(bad), w.detach(), dev(idx), info, None, None, background, eval_mode=True)[0]
    ev_r = po.composite_packed(torch.from_numpy(bad), w_ref.detach(), torch.from_numpy(ts), torch.from_numpy(te),
                               torch.from_numpy(idx), n_rays, background=background, training=False)[0]
    np.testing.assert_allclose(ev.cpu().numpy(), ev_r.numpy(), atol=3e-6)


def test_ngp_model_outputs_vs_oracle_and_trains(F):
    """NGPModel.get_outputs on the MI355X path against the oracle evaluated on the SAME packed samples (field + packed
    weights + packed compositing), then a short optimisation: the loss falls and the occupancy grid thins the samples."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.instant_ngp import InstantNGPModelConfig, NGPModel

    # NGPModel builds its NerfactoField with the field's own defaults (models/instant_ngp.py:102-109): average_init_density 1
    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 12), prop_grids=(), num_images=4, average_init_density=1.0)
    params = orc.init_params(cfg, seed=21, table_std=0.5)
    mc = InstantNGPModelConfig(grid_resolution=16, grid_levels=2, log2_hashmap_size=12, background_color="white", cone_angle=0.0,
                               render_step_size=0.02)
    model = NGPModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
    missing, unexpected = model.load_state_dict({k: v.detach().clone() for k, v in params.items() if k.startswith("field.")},
                                                strict=False)
    assert not unexpected, unexpected
    model = model.cuda().train()
    n = 96
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=4)
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 1e-6).cuda(),
                   camera_indices=cam.cuda()[:, None])
    model.update_occupancy_grid(step=0)
    assert 0 < int(model.occupancy_grid.binaries.sum()) <= model.occupancy_grid.binaries.numel()
    torch.manual_seed(0)
    jit = torch.rand(n, device="cuda")
    # the sampler's own result (candidates thinned by visibility) ...
    with torch.no_grad():
        rs_, ray_indices = model.sampler(ray_bundle=rb, near_plane=mc.near_plane, far_plane=mc.far_plane,
                                         render_step_size=mc.render_step_size, alpha_thre=mc.alpha_thre, cone_angle=mc.cone_angle,
                                         jitter=jit)
    out = model(rb, jitter=jit)
    assert int(out["num_samples_per_ray"].sum()) == ray_indices.numel() > n
    # ... fed to the oracle
    idx = ray_indices.cpu()
    ts, te = rs_.frustums.starts[:, 0].cpu(), rs_.frustums.ends[:, 0].cpu()
    pos = o[idx] + d[idx] * ((ts + te) / 2)[:, None]
    with torch.no_grad():
        dens, rgb_s, _ = orc.nerfacto_field(pos, d[idx], cam[idx], params, cfg, training=True)
        w = po.render_weight_from_density(ts, te, dens, idx, n)[0]
        comp, acc, dep = po.composite_packed(rgb_s, w, ts, te, idx, n, background="white", training=True)
    np.testing.assert_allclose(out["rgb"].detach().cpu().numpy(), comp.numpy(), atol=1e-4)          # north_star: 1e-4 RGB
    np.testing.assert_allclose(out["accumulation"].detach().cpu().numpy(), acc.numpy(), atol=1e-4)
    np.testing.assert_allclose(out["depth"].detach().cpu().numpy(), dep.numpy(), rtol=1e-3, atol=1e-4)
    # the candidates before thinning, bit-exact against the marcher restatement
    B = model.occupancy_grid.binaries.cpu().numpy().astype(bool)
    ref = po.occgrid_march(o.numpy(), d.numpy(), B, ROI, mc.render_step_size, near_plane=mc.near_plane, far_plane=mc.far_plane,
                           jitter=jit.cpu().numpy())
    cand = F.occgrid_march(o.cuda(), d.cuda(), model.occupancy_grid.binaries, ROI, mc.render_step_size, mc.near_plane,
                           mc.far_plane, None, None, 0.0, jit)
    np.testing.assert_array_equal(cand[0].cpu().numpy(), ref[0])
    np.testing.assert_array_equal(cand[1].cpu().numpy(), ref[1])
    assert ray_indices.numel() <= cand[0].numel()
    # a short optimisation through the module path
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-15)
    batch = {"image": tgt.cuda()}
    losses, counts = [], []
    for step in range(40):
        model.update_occupancy_grid(step)
        opt.zero_grad(set_to_none=True)
        res = model(rb)
        loss = model.get_loss_dict(res, batch)["rgb_loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss))
        counts.append(int(res["num_samples_per_ray"].sum()))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.6 * np.mean(losses[:5]), losses[::8]
    model.eval()
    with torch.no_grad():
        ev = model.get_outputs_for_camera_ray_bundle(rb._map(lambda t: t.view(8, 12, -1)))
    assert ev["rgb"].shape == (8, 12, 3) and float(ev["rgb"].min()) >= 0 and float(ev["rgb"].max()) <= 1
