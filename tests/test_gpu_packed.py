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


def test_ngp_explicit_schedule_honours_the_bundle_near_far(F):
    """a collider puts per-ray nears / fars on the bundle and VolumetricSampler hands them to the marcher as
    t_min / t_max (model_components/ray_samplers.py:470-476). The explicit schedule (ngp_step) must sample the same interval as
    the module path — it used to march the global near / far planes whatever the bundle carried."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.instant_ngp import InstantNGPModelConfig, NGPModel

    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 12), prop_grids=(), num_images=4, average_init_density=1.0)
    params = orc.init_params(cfg, seed=23, table_std=0.5)
    mc = InstantNGPModelConfig(grid_resolution=16, grid_levels=2, log2_hashmap_size=12, background_color="white",
                               cone_angle=0.004, render_step_size=0.02)
    model = NGPModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
    model.load_state_dict({k: v.detach().clone() for k, v in params.items() if k.startswith("field.")}, strict=False)
    model = model.cuda().train()
    n = 96
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=6)
    o, d, cam = o.cuda(), d.cuda(), cam.cuda()
    model.update_occupancy_grid(step=0)
    g = torch.Generator().manual_seed(2)
    nears = (0.4 + 0.4 * torch.rand(n, 1, generator=g)).cuda()
    fars = nears + (0.3 + 0.5 * torch.rand(n, 1, generator=g)).cuda()
    jit = torch.rand(n, generator=g).cuda()

    def bundle(with_bounds):
        return RayBundle(origins=o, directions=d, pixel_area=torch.full((n, 1), 1e-6).cuda(), camera_indices=cam[:, None],
                         nears=nears if with_bounds else None, fars=fars if with_bounds else None)

    model.config.fused_train_step = False
    ref = model(bundle(True), jitter=jit)
    free = model(bundle(False), jitter=jit)
    assert not torch.equal(ref["num_samples_per_ray"], free["num_samples_per_ray"])  # the interval matters on this batch
    model.config.fused_train_step = True
    out = model(bundle(True), jitter=jit)
    assert "ngp_step" in out
    assert torch.equal(out["num_samples_per_ray"], ref["num_samples_per_ray"])
    np.testing.assert_allclose(out["rgb"].cpu().numpy(), ref["rgb"].detach().cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(out["depth"].cpu().numpy(), ref["depth"].detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    # ... and a bundle without them afterwards marches the planes again (the flag is per batch)
    out2 = model(bundle(False), jitter=jit)
    assert torch.equal(out2["num_samples_per_ray"], free["num_samples_per_ray"])


@pytest.mark.parametrize("background", ["random", "white"])
def test_ngp_explicit_schedule_equals_the_module_path(F, background):
    """ngp_step.NgpTrainStep (static capacity-sized buffers, back-to-back launches) against NGPModel's nn.Module / autograd
    path on the same rays, lattice offsets and background draw: identical sample placement (integers), outputs / loss /
    gradients to fp32 rounding (the kept samples' positions are an fma in one route and mul + add in the other); then the
    same schedule behind the Model API (config.fused_train_step) gives the runner's own bits, and trains."""
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.instant_ngp import InstantNGPModelConfig, NGPModel
    from nerfstudio_amd.ngp_step import NgpTrainStep

    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, 12), prop_grids=(), num_images=4, average_init_density=1.0)
    params = orc.init_params(cfg, seed=23, table_std=0.5)
    mc = InstantNGPModelConfig(grid_resolution=16, grid_levels=2, log2_hashmap_size=12, background_color=background,
                               cone_angle=0.004, render_step_size=0.02)
    model = NGPModel(mc, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg.num_images)
    missing, unexpected = model.load_state_dict({k: v.detach().clone() for k, v in params.items() if k.startswith("field.")},
                                                strict=False)
    assert not unexpected, unexpected
    model = model.cuda().train()
    n = 160
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=6)
    o, d, cam, tgt = o.cuda(), d.cuda(), cam.cuda(), tgt.cuda()
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.full((n, 1), 1e-6).cuda(), camera_indices=cam[:, None])
    batch = {"image": tgt}
    model.update_occupancy_grid(step=0)
    torch.manual_seed(1)
    jit = torch.rand(n, device="cuda")
    names = [k for k, _ in model.field.named_parameters()]

    def grads():
        return {k: p.grad.detach().clone() for k, p in model.field.named_parameters()}

    # ---- module path
    model.zero_grad(set_to_none=True)
    out = model(rb, jitter=jit)
    torch.manual_seed(5)
    loss_m = model.get_loss_dict(out, batch)["rgb_loss"]
    loss_m.backward()
    g_m = grads()
    # ---- explicit schedule, capacities far too small on purpose: the buffers grow on the first batch
    model.zero_grad(set_to_none=True)
    r = NgpTrainStep(model, n, o.device, cap_candidates=16, cap_kept=16)
    r.set_batch(o, d, cam, tgt)
    r.forward(jit)
    assert r.cap_c >= r.num_candidates > 16 and r.cap_k >= r.num_kept > 16
    torch.manual_seed(5)
    loss_r = r.loss()
    r.backward()
    g_r = grads()
    res = r.outputs()
    assert torch.equal(res["num_samples_per_ray"], out["num_samples_per_ray"]) and r.num_kept == int(out["num_samples_per_ray"].sum())
    np.testing.assert_allclose(res["rgb"].cpu().numpy(), out["rgb"].detach().cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(res["accumulation"].cpu().numpy(), out["accumulation"].detach().cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(res["depth"].cpu().numpy(), out["depth"].detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(loss_r), float(loss_m.detach()), rtol=1e-5)
    for k in names:
        a, b = g_r[k].cpu().numpy(), g_m[k].cpu().numpy()
        assert np.linalg.norm(a - b) <= 1e-3 * max(np.linalg.norm(b), 1e-20), (k, np.linalg.norm(a - b), np.linalg.norm(b))
    # ---- the same schedule behind the Model API: the runner's own bits
    model.config.fused_train_step = True
    model.zero_grad(set_to_none=True)
    out2 = model(rb, jitter=jit)
    assert "ngp_step" in out2 and torch.equal(out2["rgb"], res["rgb"])
    torch.manual_seed(5)
    loss_f = model.get_loss_dict(out2, batch)["rgb_loss"]
    assert loss_f.requires_grad and float(loss_f.detach()) == float(loss_r)
    loss_f.backward()
    g_f = grads()
    for k in names:
        assert torch.equal(g_f[k], g_r[k]), k
    metrics = model.get_metrics_dict(out2, batch)
    assert int(metrics["num_samples_per_batch"]) == r.num_kept and np.isfinite(float(metrics["psnr"]))
    with pytest.raises(RuntimeError, match="unit weight"):
        out3 = model(rb, jitter=jit)
        (2.0 * model.get_loss_dict(out3, batch)["rgb_loss"]).backward()
    # ---- DynamicBatchPipeline changes the ray count every step (pipelines/dynamic_batch.py:71-95): fewer rays reuse the
    # buffers, more grow them; same runner object, same results as the module path
    fused_obj = model._fused
    for n2 in (96, 300):
        o2, d2, cam2, tgt2 = (t.cuda() for t in orc.synthetic_rays(n2, cfg.num_images, seed=20 + n2))
        rb2 = RayBundle(origins=o2, directions=d2, pixel_area=torch.full((n2, 1), 1e-6).cuda(), camera_indices=cam2[:, None])
        jit2 = torch.rand(n2, device="cuda")
        model.zero_grad(set_to_none=True)
        model.config.fused_train_step = False
        ref2 = model(rb2, jitter=jit2)
        model.config.fused_train_step = True
        got2 = model(rb2, jitter=jit2)
        assert model._fused is fused_obj and fused_obj.runner.n == n2 <= fused_obj.runner.cap_n
        assert got2["rgb"].shape == (n2, 3) and torch.equal(got2["num_samples_per_ray"], ref2["num_samples_per_ray"])
        np.testing.assert_allclose(got2["rgb"].cpu().numpy(), ref2["rgb"].detach().cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(got2["depth"].cpu().numpy(), ref2["depth"].detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        model.get_loss_dict(got2, {"image": tgt2})["rgb_loss"].backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.field.parameters())
    assert fused_obj.runner.cap_n >= 300
    # ---- and it trains (torch Adam on the gradients the schedule leaves in .grad; zero_grad(set_to_none) as the trainer's)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-15)
    losses = []
    for step in range(40):
        model.update_occupancy_grid(step)
        opt.zero_grad(set_to_none=True)
        res = model(rb)
        loss = model.get_loss_dict(res, batch)["rgb_loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), losses[::8]
    # evaluation and no_grad calls stay on the module path
    model.eval()
    with torch.no_grad():
        assert "ngp_step" not in model(rb)


@pytest.mark.parametrize("levels,res", [(4, 16), (2, 32)])
def test_marcher_with_coarse_occupancy_bits_places_the_same_samples(F, levels, res):
    """Empty-space skipping (the 4x4x4-block occupancy bitfield staged in LDS) is invisible in the result: the marcher
    with and without it, on a clustered grid (most blocks empty) and on a random one, emits identical samples; and the
    bitfield nsamd_occgrid_binarise builds is the OR of each block's cells."""
    rs = np.random.RandomState(7 + res)
    n = 301
    o, d = _rays(n, 5, scale=1.5)
    jit = rs.uniform(0, 1, n).astype(np.float32)
    for kind in ("clustered", "random"):
        if kind == "clustered":
            occ = np.zeros((levels, res, res, res), np.float32)
            c, h = res // 2, res // 4
            occ[:, c - h:c + h - 1, c - h + 1:c + h, c - h:c + h] = rs.uniform(0.2, 1.0, (levels, 2 * h - 1, 2 * h - 1, 2 * h))
            occ[0, 1, 2, 3] = 1.0          # a lone cell far from the blob
        else:
            occ = (rs.rand(levels, res, res, res) > 0.9) * rs.uniform(0.2, 1.0, (levels, res, res, res))
        occs = dev(occ.astype(np.float32).reshape(-1))
        binaries = torch.zeros((levels, res, res, res), dtype=torch.uint8, device="cuda")
        words = F.occgrid_coarse_words(levels, res)
        assert words == (levels * (res // 4) ** 3 + 31) // 32
        coarse = torch.full((words,), -1, dtype=torch.int32, device="cuda")
        stats = torch.zeros(2, device="cuda")
        F.occgrid_binarise(occs, binaries, coarse, 0.01, torch.zeros(1024, dtype=torch.float64, device="cuda"), stats)
        B = po.occgrid_thresholds(occ.reshape(-1), 0.01).reshape(levels, res, res, res)
        np.testing.assert_array_equal(binaries.cpu().numpy().astype(bool), B)
        np.testing.assert_allclose(float(stats[1]), occ.astype(np.float64).mean(), rtol=1e-6)
        blocks = B.reshape(levels, res // 4, 4, res // 4, 4, res // 4, 4).any(axis=(2, 4, 6)).reshape(-1)
        bits = ((coarse.cpu().numpy().astype(np.int64)[:, None] >> np.arange(32)) & 1).astype(bool).reshape(-1)[: blocks.size]
        np.testing.assert_array_equal(bits, blocks)
        if kind == "clustered":
            assert blocks.mean() < 0.35
        with_bits = F.occgrid_march(dev(o), dev(d), binaries, ROI, 0.02, 0.05, 50.0, None, None, 0.004, dev(jit), coarse=coarse)
        without = F.occgrid_march(dev(o), dev(d), binaries, ROI, 0.02, 0.05, 50.0, None, None, 0.004, dev(jit))
        ref = po.occgrid_march(o, d, B, ROI, 0.02, near_plane=0.05, far_plane=50.0, cone_angle=0.004, jitter=jit)
        assert len(ref[0]) > 50
        for a, b, c_ in zip(with_bits[:3], without[:3], ref):
            np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
            np.testing.assert_array_equal(a.cpu().numpy(), c_)


@pytest.mark.parametrize("stash_cap", [3, 40, 4096])
def test_marcher_that_marches_every_ray_once_places_the_same_samples(F, stash_cap):
    """nsamd_occgrid_march_count_stash / _write_stashed (the count pass leaves a ray's first `stash_cap` kept steps behind, the
    write pass copies them and marches only the rays that kept more a second time) against the plain two-pass pair and the
    oracle: identical ray indices, bin edges and packed_info — with a stash nearly every ray overflows (3), one that some rays
    overflow (40) and one nobody does."""
    rs = np.random.RandomState(11)
    levels, res, n = 2, 32, 257
    o, d = _rays(n, 5, scale=1.5)
    jit = rs.uniform(0, 1, n).astype(np.float32)
    B = rs.rand(levels, res, res, res) > 0.8
    binaries = torch.from_numpy(B.astype(np.uint8)).cuda()
    plain = F.occgrid_march(dev(o), dev(d), binaries, ROI, 0.02, 0.05, 50.0, None, None, 0.004, dev(jit), stash_cap=0)
    once = F.occgrid_march(dev(o), dev(d), binaries, ROI, 0.02, 0.05, 50.0, None, None, 0.004, dev(jit), stash_cap=stash_cap)
    ref = po.occgrid_march(o, d, B, ROI, 0.02, near_plane=0.05, far_plane=50.0, cone_angle=0.004, jitter=jit)
    counts = plain[3][:, 1].cpu().numpy()
    assert counts.max() > 40 > np.median(counts) > 3  # the three stash sizes do exercise the three cases
    for a, b in zip(once, plain):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    for a, c_ in zip(once[:3], ref):
        np.testing.assert_array_equal(a.cpu().numpy(), c_)


def test_field_forward_without_the_head_gives_the_same_density(F):
    """nsamd_field_mlp_fwd with rgb == NULL (Field.density_fn: what the instant-ngp sampler asks of its candidates) against the
    full call: the density bit for bit, ragged M."""
    from nerfstudio_amd import _native as N

    g = torch.Generator().manual_seed(4)
    M = 16 * 37 + 5
    enc = (torch.randn(32, M, generator=g) * 0.5).cuda()
    sel = (torch.rand(M, generator=g) > 0.1).float().cuda()
    dirs = torch.nn.functional.normalize(torch.randn(1, 3, generator=g), dim=-1).cuda()
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 63), (64,), (64, 64), (64,), (3, 64), (3,)]
    prm = [(torch.randn(*sh, generator=g) * 0.3).cuda().contiguous() for sh in shapes]
    app0 = torch.zeros(32, device="cuda")
    fm = N.FieldMlp(*(N.ptr(p) for p in prm), None, 0, 1.0)
    lib = N.load()
    d_full, rgb = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    d_only = torch.full((M,), -1.0, device="cuda")
    N.check(lib.nsamd_field_mlp_fwd(N.ptr(enc), N.ptr(sel), N.ptr(dirs), None, N.ptr(app0), M, M, fm, N.ptr(d_full), N.ptr(rgb),
                                    N.stream()), "field_mlp_fwd")
    N.check(lib.nsamd_field_mlp_fwd(N.ptr(enc), N.ptr(sel), N.ptr(dirs), None, N.ptr(app0), M, M, fm, N.ptr(d_only), None,
                                    N.stream()), "field_mlp_fwd (density only)")
    torch.cuda.synchronize()
    assert torch.equal(d_only, d_full) and float(d_full.abs().max()) > 0 and bool(torch.isfinite(rgb).all())


def test_occupancy_grid_refresh_kernels_equal_the_torch_path(F):
    """OccGridEstimator.update_every_n_steps on the GPU (cell positions, decayed maximum with repeated cells, mean /
    threshold / binaries — csrc/packed.hip) against the same module on CPU tensors (plain torch), fed the same cells, offsets
    and density estimates: positions, occs and binaries equal bit for bit, warm-up refresh and partial refresh."""
    from nerfstudio_amd.model_components.occupancy import OccGridEstimator

    def blob(x):  # a density-like field; evaluated on the CPU for both grids (identical estimates)
        x = x.detach().cpu()
        return (torch.exp(-3.0 * (x * x).sum(-1, keepdim=True)) * 0.4 + 1e-6).float()  # (no denormals: density x step never is)

    res, levels = 16, 3
    total = levels * res ** 3
    g_cpu = OccGridEstimator(torch.tensor(ROI), resolution=res, levels=levels).train()
    g_gpu = OccGridEstimator(torch.tensor(ROI), resolution=res, levels=levels).cuda().train()
    gen = torch.Generator().manual_seed(3)
    # positions of every cell / of a list with repeats
    jit_all = torch.rand((total, 3), generator=gen)
    x_cpu = g_cpu._cell_positions(None, total, jit_all)
    x_gpu = g_gpu._cell_positions(None, total, jit_all.cuda())
    assert torch.equal(x_cpu, x_gpu.cpu())
    cells = torch.randint(total, (5000,), generator=gen)
    cells[100:200] = cells[0:100]  # repeats
    jit = torch.rand((5000, 3), generator=gen)
    assert torch.equal(g_cpu._cell_positions(cells, 5000, jit), g_gpu._cell_positions(cells.cuda(), 5000, jit.cuda()).cpu())
    # a warm-up refresh and two partial ones through the public entry point, with the random draws pinned
    for step, seed in ((0, 1), (256, 2), (272, 3)):
        outs = []
        if step >= 256:  # same cells / offsets on both grids: drawn once, on the CPU, from the state BEFORE the refresh
            k = total // 4
            gen2 = torch.Generator().manual_seed(100 + seed)
            uniform = torch.randint(total, (k,), generator=gen2)
            occupied = torch.nonzero(g_cpu.binaries.reshape(-1)).reshape(-1)
            flat = torch.cat([uniform, occupied[:k]])
            offs = torch.rand((flat.numel(), 3), generator=gen2)
            assert flat.numel() > flat.unique().numel()  # repeated cells are part of the case
        for g in (g_cpu, g_gpu):
            if step >= 256:
                x = g._cell_positions(flat.to(g.occs.device), flat.numel(), offs.to(g.occs.device))
                occ = blob(x).reshape(-1).to(g.occs.device)
                if g.occs.is_cuda:
                    F.occgrid_update(g.occs, flat.cuda(), occ, 0.95, g._buf("old", total, torch.float32))
                else:
                    new = g.occs.clone()
                    new[flat] = g.occs[flat] * 0.95
                    g.occs.copy_(new.scatter_reduce(0, flat, occ, "amax", include_self=True))
                g._refresh_derived(0.01)
            else:
                offs = torch.rand((total, 3), generator=torch.Generator().manual_seed(50))
                x = g._cell_positions(None, total, offs.to(g.occs.device))
                occ = blob(x).reshape(-1).to(g.occs.device)
                if g.occs.is_cuda:
                    F.occgrid_update(g.occs, None, occ, 0.95, g._buf("old", total, torch.float32))
                else:
                    g.occs.copy_(torch.maximum(g.occs * 0.95, occ))
                g._refresh_derived(0.01)
            outs.append((g.occs.detach().cpu().clone(), g.binaries.detach().cpu().clone(), g._occ_mean))
        assert torch.equal(outs[0][0], outs[1][0]), f"occs differ at step {step}"
        assert torch.equal(outs[0][1], outs[1][1]), f"binaries differ at step {step}"
        assert abs(outs[0][2] - outs[1][2]) <= 1e-7 * max(abs(outs[0][2]), 1e-9)
        assert 0 < int(outs[0][1].sum()) < total
    # and the public entry point runs end to end on the device
    g_gpu.update_every_n_steps(step=288, occ_eval_fn=lambda x: blob(x).cuda())
    assert g_gpu._coarse is not None and g_gpu._coarse_version == g_gpu.binaries._version


def test_ngp_bench_size_parity_vs_oracle(F):
    """BASELINE configs[3] at the BENCHMARK's own size, through the benchmark's own objects (scripts/bench_ngp.build_ngp:
    128^3 x 4 occupancy grid, 4096 rays, T = 2^19, the lifted synthetic field) — what tests/test_gpu_bench_parity.py is to
    the nerfacto line. One iteration of ngp_step.NgpTrainStep with injected lattice offsets and background draws:
      * the candidates of the first 192 rays bit-exact against the marcher restatement (the numpy marcher walks step by step);
      * the visibility mask bit-exact against the oracle's scan over the kernels' own candidate densities;
      * rgb / accumulation <= 1e-4 L-inf, the loss, and EVERY field gradient (the table per level) against the oracle
        evaluated on the kernels' own packed samples — bounded by 4 x the fp32 reference's own distance from a float64
        evaluation of the same graph (floor 5e-4), as on the nerfacto line;
      * the occupancy refresh of the training iteration runs on this state and (with the bench's hook) leaves the grid as it was."""
    import bench
    from scripts.bench_ngp import build_ngp

    dev_ = torch.device("cuda")
    F._SCATTER_WS.clear()
    model, arena, tr, (o, d, cam, tgt) = build_ngp(dev_, bench.synthetic_rays)
    cfg, grid, r, n = model.config, model.occupancy_grid, tr.runner, bench.RAYS_PER_GPU
    assert tuple(grid.binaries.shape) == (4, 128, 128, 128) and model.field.mlp_base.encoding.spec.log2_hashmap_size == 19
    before = grid.binaries.clone()
    tr.update_occupancy_grid(512)  # a refresh step of the schedule: 2.5 M cell densities, decayed maximum, threshold, bitfield
    assert len(tr.refreshes) == 1 and torch.equal(grid.binaries, before)
    rs = np.random.RandomState(17)
    jit = torch.from_numpy(rs.uniform(0, 1, n).astype(np.float32)).cuda()
    bg = torch.from_numpy(rs.uniform(0, 1, (n, 3)).astype(np.float32))
    arena.zero_grad(skip=[tr.table])
    r.forward(jit)
    loss = r.loss(background=bg.cuda())
    r.backward()
    torch.cuda.synchronize()
    mc, mk = r.num_candidates, r.num_kept
    assert mc > 20 * n and 10 * n < mk < mc
    # ---- marcher: the first rays' candidates, integers and bin edges bit for bit
    sub = 192
    B = grid.binaries.cpu().numpy().astype(bool)
    ref = po.occgrid_march(o[:sub], d[:sub], B, ROI, cfg.render_step_size, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
                           cone_angle=cfg.cone_angle, jitter=jit[:sub].cpu().numpy())
    info = r.info.cpu().numpy()
    m_sub = int(info[sub - 1, 0] + info[sub - 1, 1])
    np.testing.assert_array_equal(r.c_ri[:m_sub].cpu().numpy(), ref[0])
    np.testing.assert_array_equal(r.c_ts[:m_sub].cpu().numpy(), ref[1])
    np.testing.assert_array_equal(r.c_te[:m_sub].cpu().numpy(), ref[2])
    # ---- visibility scan + early termination on the kernels' own candidate densities
    c_ri, c_ts, c_te, c_sigma = (t[:mc].cpu() for t in (r.c_ri, r.c_ts, r.c_te, r.c_sigma))
    alpha = min(float(cfg.alpha_thre), float(grid._occ_mean))
    keep = po.render_visibility_from_density(c_ts, c_te, c_sigma, c_ri, n, 1e-4, alpha)
    assert int(keep.sum()) == mk and torch.equal(keep, r.c_mask[:mc].cpu().bool())
    # ---- the field, packed weights and compositing on the kernels' kept samples
    idx, ts, te = r.k_ri[:mk].cpu(), r.k_ts[:mk].cpu(), r.k_te[:mk].cpu()
    assert torch.equal(idx, c_ri[keep]) and torch.equal(ts, c_ts[keep])
    ocfg = orc.NerfactoCfg(prop_grids=(), num_images=100, average_init_density=1.0)
    keys = [k for k in orc.init_params(ocfg, seed=0) if k.startswith("field.")]
    sd = model.state_dict()
    base = {k: sd[k].detach().cpu().clone() for k in keys}
    to, td, tcam, ttgt = (torch.from_numpy(a) for a in (o, d, cam[:, 0], tgt))

    def oracle(dtype):
        prm = {k: v.to(dtype).clone().requires_grad_(True) for k, v in base.items()}
        t0, t1 = ts.to(dtype), te.to(dtype)
        pos = to.to(dtype)[idx] + td.to(dtype)[idx] * ((t0 + t1) / 2)[:, None]
        dens, rgb_s, _ = orc.nerfacto_field(pos, td.to(dtype)[idx], tcam[idx], prm, ocfg, training=True)
        w = po.render_weight_from_density(t0, t1, dens, idx, n)[0]
        comp, acc, dep = po.composite_packed(rgb_s, w, t0, t1, idx, n, background="random", training=True)
        pred = comp + bg.to(dtype) * (1.0 - acc)
        val = torch.mean((pred - ttgt.to(dtype)) ** 2)
        val.backward()
        return prm, comp.detach(), acc.detach(), val.detach()

    p32, comp, acc, val = oracle(torch.float32)
    p64, _, _, _ = oracle(torch.float64)
    rgb_err = float((r.rgb.cpu() - comp).abs().max())
    acc_err = float((r.acc.cpu() - acc[:, 0]).abs().max())
    assert rgb_err <= 1e-4 and acc_err <= 1e-4, (rgb_err, acc_err)       # north_star: 1e-4 RGB L-inf
    np.testing.assert_allclose(float(loss), float(val), rtol=2e-5)
    named = dict(model.named_parameters())
    worst = ("", 0.0, 0.0)
    T = 1 << 19

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))

    for k in keys:
        got = named[k].grad.detach().cpu().numpy().astype(np.float64)
        g32, g64 = p32[k].grad.numpy().astype(np.float64), p64[k].grad.numpy()
        parts = [(k, got, g32, g64)]
        if k.endswith("hash_table"):  # the main table level by level (its levels differ by orders of magnitude)
            parts = [(f"{k}[L{l}]", got[l * T:(l + 1) * T], g32[l * T:(l + 1) * T], g64[l * T:(l + 1) * T]) for l in range(16)]
        for name, a, b32, b64 in parts:
            if np.linalg.norm(b64) == 0:
                assert np.linalg.norm(a) == 0, name
                continue
            e_gpu, e_ref = rel(a, b64), rel(b32, b64)
            assert e_gpu <= 4 * max(e_ref, 5e-4), f"{name}: |gpu - f64| = {e_gpu:.2e}, |ref32 - f64| = {e_ref:.2e}"
            if e_gpu > worst[1]:
                worst = (name, e_gpu, e_ref)
    for ws in F._SCATTER_WS.values():
        ev = F.scatter_events(ws)
        assert ev[1] == 0 and ev[2] == 0, f"scatter records on an unordered path / lost: {ev}"
    print(f"\nngp bench-size parity: {mc} candidates, {mk} kept; rgb L-inf {rgb_err:.2e}, acc {acc_err:.2e}, loss {float(loss):.6f}; "
          f"worst gradient {worst[0]}: (gpu-f64, ref32-f64) rel-L2 = ({worst[1]:.2e}, {worst[2]:.2e}); refresh {tr.refreshes[0]:.2f} ms")
