"""The training iteration's merged launches against the launches they replace, BIT FOR BIT (round 5):

* nsamd_render_losses_train = nsamd_render_train + nsamd_proposal_losses + nsamd_render_train_bwd [+ nsamd_weights_bwd(_gate) per
  proposal level]: one launch, one wave per (ray, job), the stand-alone launches' device bodies run one after the other inside the
  wave (csrc/ray_bodies.h, csrc/fused_rays.hip) — so every output must be the same bits, including the ones the launch reads
  back itself (fine weights, MSE gradient, distortion gradient, interlevel gradient);
* nsamd_select_bins = nsamd_select_batch + nsamd_piecewise_bins.

The separate launches are pinned to the oracle / the reference's fixtures by tests/test_gpu_kernels.py; this file pins the merged
forms to them and the whole runner (a few optimisation steps, both update kinds, gated and ungated proposal chains, every
background mode) to the separate-launch runner."""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from nerfstudio_amd import _native, functional

    _native.load()
    return functional


def _bits(t):
    return t.detach().contiguous().view(torch.int32) if t.dtype == torch.float32 else t.detach()


def _same(a, b, what):
    assert a.shape == b.shape, what
    assert torch.equal(_bits(a), _bits(b)), f"{what}: {int((_bits(a) != _bits(b)).sum())} of {a.numel()} words differ"


def _runner(cfg, n, seed, background, fuse, gate=True, fold=True):
    from test_gpu_kernels import _hip_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    model = _hip_model(cfg, orc.init_params(cfg, seed=seed, table_std=0.4))
    model.config.background_color = background
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    r = NerfactoTrainStep(model, n, torch.device("cuda"))
    r.fuse_rays, r.fold_weights_bwd, r.gate_proposals = fuse, fold, gate
    r.side_stream = None
    return model, arena, r


@pytest.mark.parametrize("background", ["last_sample", "white", "random"])
@pytest.mark.parametrize("gate", [True, False])
def test_render_losses_train_equals_the_separate_launches(F, background, gate):
    from test_gpu_kernels import small_cfg

    cfg = small_cfg(12, 10, 6)
    n = 333  # not a multiple of the four rays of a workgroup: tail waves
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=3)
    rs = np.random.RandomState(5)
    jit = torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    bg = torch.from_numpy(rs.uniform(0, 1, (n, 3)).astype(np.float32)).cuda()
    runs = {}
    for fuse in (False, True):
        F._SCATTER_WS.clear()
        model, arena, r = _runner(cfg, n, 11, background, fuse, gate=gate)
        r.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
        r.jitter.copy_(jit)
        if r.bg_rays is not None:
            r.bg_rays.copy_(bg)
        r.anneal_dev.fill_(0.7)
        states = []
        for step, updated in enumerate((True, False, True)):
            arena.zero_grad(skip=r.written_params())
            r.forward_backward(updated, draw_jitter=False)
            torch.cuda.synchronize()
            L = r.n_prop
            snap = {"weights": r.weights[L], "rgb": r.rgb, "acc": r.acc, "depth_exp": r.depth_exp, "depth_med": r.depth_med[L],
                    "sq_err": r.sq_err, "d_rgb_out": r.d_rgb_out, "dist_per_ray": r.dist_per_ray, "dw_dist": r.dw_dist,
                    "d_rgb_s": r.d_rgb_s, "d_dens_main": r.d_dens_main, "grad": arena.grad, "minmax": r.minmax_ws[:2]}
            for lvl in range(L):
                snap[f"inter{lvl}"] = r.inter_per_ray[lvl]
                if updated:
                    snap[f"dw_prop{lvl}"], snap[f"p_ddens{lvl}"] = r.dw_prop[lvl], r.p_ddens[lvl]
                    if gate:
                        snap[f"mask{lvl}"] = r.prop_ray_masks[lvl]
            if updated and gate:
                snap["gates"] = r.prop_gates
            states.append({k: v.detach().clone() for k, v in snap.items()})
            arena.step(groups=["fields", "proposal_networks"] if updated else ["fields"])
        states.append({"params": arena.flat.clone(), "m": arena.exp_avg.clone(), "v": arena.exp_avg_sq.clone()})
        runs[fuse] = states
        del model, arena, r
    assert any(float(s["grad"].abs().max()) > 0 for s in runs[True][:3])
    assert float(runs[True][0]["p_ddens0"].abs().max()) > 0, "the proposal level carried no gradient: the test shows nothing"
    for i, (a, b) in enumerate(zip(runs[False], runs[True])):
        assert a.keys() == b.keys()
        for k in a:
            _same(a[k], b[k], f"iteration {i}, {k} ({background}, gated={gate})")


def test_folded_weights_backward_equals_its_own_launch_and_respects_the_switches(F):
    """The level's weights backward inside the losses launch against the stand-alone nsamd_weights_bwd_gate at the head of the
    level's chain (NSAMD_FOLD_WEIGHTS_BWD=0), and a chain that is asked for twice runs its own launch the second time."""
    from test_gpu_kernels import small_cfg

    cfg = small_cfg(12, 10, 6)
    n = 256
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=4)
    jit = torch.from_numpy(np.random.RandomState(1).uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    out = {}
    for fold in (False, True):
        F._SCATTER_WS.clear()
        model, arena, r = _runner(cfg, n, 12, "last_sample", True, fold=fold)
        r.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
        r.jitter.copy_(jit)
        arena.zero_grad(skip=r.written_params())
        r.forward_and_losses(True, draw_jitter=False)
        assert (len(r._wb_folded) == r.n_prop) == fold
        r.backward_all(True)
        assert not r._wb_folded
        torch.cuda.synchronize()
        first = arena.grad.clone()
        # the chains once more without a new `losses`: nothing is folded any more, each level launches its own weights backward
        a, b = arena.groups["proposal_networks"]
        arena.grad[a:b].zero_()
        r.prop_gates.zero_()
        r.backward_proposals()
        torch.cuda.synchronize()
        _same(arena.grad[a:b], first[a:b], f"second call of the proposal chains (fold={fold})")
        out[fold] = (first, [x.clone() for x in r.p_ddens], [x.clone() for x in r.prop_ray_masks])
    _same(out[False][0], out[True][0], "gradient arena")
    for lvl in range(2):
        _same(out[False][1][lvl], out[True][1][lvl], f"p_ddens[{lvl}]")
        _same(out[False][2][lvl], out[True][2][lvl], f"ray mask [{lvl}]")


@pytest.mark.parametrize("n", [333, 4096])
@pytest.mark.parametrize("with_pool", [False, True])
def test_proposal_sampler_one_launch_equals_the_per_level_launches(F, n, with_pool):
    """nsamd_proposal_sampler ([batch selection,] initial bins, then per level density -> weights -> median depth -> resampling,
    one wavefront per ray) against nsamd_select_batch / nsamd_piecewise_bins / nsamd_density_field_fwd / nsamd_proposal_resample:
    bin edges of every level, densities, weights, median depths and — on the steps that keep them for the backward — the encoded
    features, selectors and pre-activations, bit for bit; annealed resampling included."""
    from test_gpu_kernels import small_cfg

    from nerfstudio_amd import _native as N

    cfg = small_cfg(12, 10, 6)
    rs = np.random.RandomState(7)
    slots = 3
    pool = {"origins": torch.from_numpy((rs.standard_normal((slots, n, 3)) * 0.5).astype(np.float32)).cuda(),
            "directions": torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((slots, n, 3)).astype(np.float32)), dim=-1).cuda(),
            "cameras": torch.from_numpy(rs.randint(0, cfg.num_images, (slots, n))).cuda(),
            "target": torch.from_numpy(rs.uniform(0, 1, (slots, n, 3)).astype(np.float32)).cuda()}
    jit = torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    slot = torch.tensor([1.0], device="cuda")
    out = {}
    for fused in (False, True):
        F._SCATTER_WS.clear()
        model, arena, r = _runner(cfg, n, 13, "last_sample", True)
        r.fuse_sampler = fused
        r.fuse_select = fused
        r.jitter.copy_(jit)
        r.anneal_dev.fill_(0.6)
        snaps = []
        for need_enc in (True, False):
            for buf in (r.p_enc + r.p_sel + r.p_pre + r.p_dens + r.weights + r.s_bins + r.t_bins + r.depth_med):
                buf.fill_(-7.0)
            if with_pool:
                if fused:
                    r.pending_select = (N.ptr(slot), slots, pool)
                else:
                    N.check(N.load().nsamd_select_batch(N.ptr(slot), slots, n, N.ptr(pool["origins"]), N.ptr(pool["directions"]),
                                                        N.ptr(pool["cameras"]), N.ptr(pool["target"]), N.ptr(r.origins),
                                                        N.ptr(r.directions), N.ptr(r.camera_indices), N.ptr(r.target), N.stream()),
                            "select_batch")
            else:
                r.set_batch(pool["origins"][2], pool["directions"][2], pool["cameras"][2], pool["target"][2])
            r.forward_proposals(draw_jitter=False, need_enc=need_enc)
            torch.cuda.synchronize()
            assert r._sampler_ok and r.pending_select is None
            snap = {"origins": r.origins, "directions": r.directions, "cams": r.camera_indices, "target": r.target}
            for lvl in range(3):
                snap[f"s_bins{lvl}"], snap[f"t_bins{lvl}"] = r.s_bins[lvl], r.t_bins[lvl]
            for lvl in range(2):
                snap[f"dens{lvl}"], snap[f"w{lvl}"], snap[f"med{lvl}"] = r.p_dens[lvl], r.weights[lvl], r.depth_med[lvl]
                snap[f"enc{lvl}"], snap[f"sel{lvl}"], snap[f"pre{lvl}"] = r.p_enc[lvl], r.p_sel[lvl], r.p_pre[lvl]
            snaps.append({k: v.detach().clone() for k, v in snap.items()})
        out[fused] = snaps
        del model, arena, r
    for i, (a, b) in enumerate(zip(out[False], out[True])):
        assert float(a["dens0"].min()) > -7.0 and float(a["t_bins2"].min()) > -7.0
        if i == 0:
            assert float((a["enc0"] == -7.0).float().mean()) < 0.01, "the features must have been written on this pass"
        else:
            assert bool((b["enc0"] == -7.0).all()) and bool((b["pre1"] == -7.0).all()), "nothing kept for a backward that will not run"
        for k in a:
            _same(a[k], b[k], f"pass {i}, {k} (n={n}, pool={with_pool})")


@pytest.mark.parametrize("per_edge", [False, True])
def test_select_bins_equals_select_batch_plus_piecewise_bins(F, per_edge):
    from nerfstudio_amd import _native as N

    lib = N.load()
    n, slots, S = 1001, 3, 256
    g = torch.Generator().manual_seed(0)
    pool = {"origins": torch.randn(slots, n, 3, generator=g).cuda(), "directions": torch.randn(slots, n, 3, generator=g).cuda(),
            "cameras": torch.randint(0, 50, (slots, n), generator=g).cuda(), "target": torch.rand(slots, n, 3, generator=g).cuda()}
    nears, fars = torch.full((n,), 0.05).cuda(), torch.full((n,), 1000.0).cuda()
    edges = F._linspace("edges", S, torch.device("cuda"))
    jit = (torch.rand(n, S + 1, generator=g) if per_edge else torch.rand(n, generator=g)).cuda()
    st = N.stream()
    for slot_value in (0.0, 2.0, 7.0):  # (clamped to the last slot)
        slot = torch.tensor([slot_value], device="cuda")
        outs = []
        for merged in (False, True):
            o, d, t = (torch.full((n, 3), -1.0, device="cuda") for _ in range(3))
            c = torch.full((n,), -1, dtype=torch.int64, device="cuda")
            sb, tb = (torch.full((n, S + 1), -1.0, device="cuda") for _ in range(2))
            if merged:
                N.check(lib.nsamd_select_bins(N.ptr(slot), slots, n, N.ptr(pool["origins"]), N.ptr(pool["directions"]),
                                              N.ptr(pool["cameras"]), N.ptr(pool["target"]), N.ptr(o), N.ptr(d), N.ptr(c), N.ptr(t),
                                              N.ptr(nears), N.ptr(fars), N.ptr(edges), N.ptr(jit), int(per_edge), S, 0, N.ptr(sb),
                                              N.ptr(tb), st), "select_bins")
            else:
                N.check(lib.nsamd_select_batch(N.ptr(slot), slots, n, N.ptr(pool["origins"]), N.ptr(pool["directions"]),
                                               N.ptr(pool["cameras"]), N.ptr(pool["target"]), N.ptr(o), N.ptr(d), N.ptr(c), N.ptr(t),
                                               st), "select_batch")
                N.check(lib.nsamd_piecewise_bins(N.ptr(nears), N.ptr(fars), N.ptr(edges), N.ptr(jit), int(per_edge), n, S, 0,
                                                 N.ptr(sb), N.ptr(tb), st), "piecewise_bins")
            torch.cuda.synchronize()
            outs.append((o, d, c, t, sb, tb))
        k = min(int(slot_value), slots - 1)
        assert torch.equal(outs[1][0], pool["origins"][k]) and torch.equal(outs[1][2], pool["cameras"][k])
        for a, b, what in zip(outs[0], outs[1], ("origins", "directions", "cameras", "target", "s_bins", "t_bins")):
            _same(a, b, f"{what} (slot {slot_value})")


def test_trainer_with_merged_launches_trains_through_the_same_bits(F, monkeypatch):
    """bench.py's trainer (captured graphs, deferred main-field Adam, pool of batches, bench-size batch) with the merged
    launches against the same trainer on the separate launches: identical parameter and moment bits after 15 iterations of both
    update kinds, replayed from the captured graphs."""
    import hashlib

    import bench

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.trainer import HipTrainer

    digests = {}
    dev = torch.device("cuda")
    for arm, env in (("merged", {}), ("separate", {"NSAMD_FUSE_RAYS": "0", "NSAMD_FUSE_SELECT": "0", "NSAMD_FUSE_SAMPLER": "0"})):
        for k in ("NSAMD_FUSE_RAYS", "NSAMD_FUSE_SELECT", "NSAMD_FUSE_SAMPLER"):
            monkeypatch.delenv(k, raising=False)
        if arm == "merged":  # (both off by default: measured slower — they must still train the same bits)
            monkeypatch.setenv("NSAMD_FUSE_SAMPLER", "1")
            monkeypatch.setenv("NSAMD_FUSE_RAYS", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        F._SCATTER_WS.clear()
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        model = bench.build_model(dev, seed=0)
        arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
        rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
        tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=True, use_runner=True, pool=pool)
        assert tr.runner.fuse_rays == tr.runner.fuse_select == tr.runner.fuse_sampler == (arm == "merged")
        torch.manual_seed(1)
        torch.cuda.manual_seed(1)  # the jitter draws of the iterations below
        tr.train_iteration()
        tr.finish()
        assert tr.try_capture()
        torch.manual_seed(2)
        torch.cuda.manual_seed(2)
        kinds = set()
        for _ in range(14):  # (every iteration before step 10 updates the proposal networks, then every other one)
            kinds.add(bool(model.proposal_sampler.updated_this_step()))
            tr.train_iteration()
        tr.finish()
        torch.cuda.synchronize()
        assert kinds == {True, False}
        digests[arm] = tuple(hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest()
                             for x in (arena.flat, arena.exp_avg, arena.exp_avg_sq))
        assert bool(torch.isfinite(tr.last_loss()))
        del tr, arena, model
    assert digests["merged"] == digests["separate"], digests


def test_weight_gradient_reduce_riding_the_apply_pass_equals_its_own_launch(F):
    """nsamd_field_mlp_bwd_scatter carries the weight-gradient reduce as extra workgroups of the scatter's apply pass
    (csrc/field_reduce.h); the phase entry point runs the three launch groups one at a time (gradient kernel, reduce, apply):
    the whole fields slice of the gradient arena — MLP weights, biases, appearance embedding, the table — must be the same bits,
    at the benchmark's own size (1024-thread apply workgroups, 100 cameras)."""
    import bench

    from nerfstudio_amd import _native as N
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    F._SCATTER_WS.clear()
    dev = torch.device("cuda")
    model = bench.build_model(dev, seed=0)
    with torch.no_grad():  # (tables away from the all-but-zero initial state)
        model.field.mlp_base.encoding.hash_table.normal_(0.0, 0.3)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    o, d, cam, tgt = (torch.from_numpy(a).to(dev) for a in bench.synthetic_rays(1003))
    r = NerfactoTrainStep(model, bench.RAYS_PER_GPU, dev)
    r.side_stream = None
    r.set_batch(o, d, cam[:, 0], tgt)
    r.jitter.copy_(torch.from_numpy(np.random.RandomState(4).uniform(0, 1, (3, bench.RAYS_PER_GPU)).astype(np.float32)))
    r.forward_and_losses(False, draw_jitter=False)
    a, b = arena.groups["fields"]
    out = {}
    for mode in ("rider", "phases"):
        arena.zero_grad(["fields"], skip=r.written_params())
        N.PROFILE = {} if mode == "phases" else None  # (the per-kernel table's route: one launch group per call)
        try:
            r.backward_main()
        finally:
            N.PROFILE = None
        torch.cuda.synchronize()
        out[mode] = arena.grad[a:b].clone()
    assert float(out["rider"].abs().max()) > 0 and not torch.isnan(out["rider"]).any()
    _same(out["rider"], out["phases"], "fields slice of the gradient arena")


def test_loss_values_of_the_finishing_pass_and_of_their_own_launch(F):
    """The five floats a trainer reads every iteration (rgb / interlevel / distortion loss, psnr, distortion metric): written by
    nsamd_render_losses_train's finishing pass and by nsamd_train_loss_values behind the separate launches — equal to each other
    bit for bit (same per-ray terms, same fixed summation order) and to the float64 sums of the per-ray terms to 1e-6."""
    from test_gpu_kernels import small_cfg

    cfg = small_cfg(12, 10, 6)
    n = 1000
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=9)
    jit = torch.from_numpy(np.random.RandomState(2).uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    vals = {}
    for fuse in (False, True):
        F._SCATTER_WS.clear()
        model, arena, r = _runner(cfg, n, 14, "last_sample", fuse)
        r.want_loss_vals = True
        r.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
        r.jitter.copy_(jit)
        arena.zero_grad(skip=r.written_params())
        r.forward_and_losses(True, draw_jitter=False)
        torch.cuda.synchronize()
        assert r._loss_vals_fresh
        ld = r.loss_dict()
        assert ld["rgb_loss"].data_ptr() == r.loss_vals.data_ptr()
        mc, S = model.config, r.counts[-1]
        ref = {"rgb_loss": float(r.sq_err.double().sum()) / (3 * n),
               "interlevel_loss": mc.interlevel_loss_mult * float(sum(p.double().sum() for p in r.inter_per_ray)) / (n * S),
               "distortion_loss": mc.distortion_loss_mult * float(r.dist_per_ray.double().sum()) / n}
        for k, v in ref.items():
            assert abs(float(ld[k]) - v) <= 1e-6 * max(abs(v), 1e-6) + 1e-9, (k, float(ld[k]), v)
        assert abs(float(r.loss_vals[3]) + 10.0 * np.log10(ref["rgb_loss"])) <= 1e-4
        assert abs(float(r.loss_vals[4]) - float(r.dist_per_ray.double().sum()) / n) <= 1e-6 * float(r.loss_vals[4]) + 1e-9
        assert abs(float(r.loss_vals[5]) - sum(ref.values())) <= 2e-6 * sum(ref.values())
        vals[fuse] = r.loss_vals.clone()
        del model, arena, r
    _same(vals[False], vals[True], "loss values: own launch vs the merged launch's finishing pass")


def _philox_reference(seed, draw, q):
    """Philox-4x32-10 (Salmon et al. 2011) on counter (q, draw) with key `seed`, as csrc/misc.hip runs it -> 4 uint32."""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c = [q & mask, (q >> 32) & mask, draw & mask, (draw >> 32) & mask]
    k = [seed & mask, (seed >> 32) & mask]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & mask, p1 & mask, ((p0 >> 32) ^ c[3] ^ k[1]) & mask, p0 & mask]
        k = [(k[0] + W0) & mask, (k[1] + W1) & mask]
    return c


def test_step_prologue_rows_follow_the_table_and_draws_are_philox(F):
    """nsamd_step_prologue: row `counter[0] % rows` of the host's table lands in `hyper`, both counters advance by one per launch,
    and the step's uniforms are Philox-4x32-10 keyed by (seed, draw counter) — the numbers of a Python restatement, bit for bit —,
    uniform on [0, 1) and different from step to step."""
    from nerfstudio_amd import _native as N

    lib = N.load()
    rows, n0, n1, seed = 3, 1003, 501, 0x1234567890ABCDEF
    table = torch.arange(rows * 8, dtype=torch.float32, device="cuda") * 0.5 + 1.0
    counter = torch.tensor([1, 7], dtype=torch.int64, device="cuda")
    hyper = torch.zeros(8, device="cuda")
    u0, u1 = torch.full((n0,), -1.0, device="cuda"), torch.full((n1,), -1.0, device="cuda")
    draws = []
    for launch in range(4):
        N.check(lib.nsamd_step_prologue(N.ptr(counter), N.ptr(table), rows, N.ptr(hyper), N.ptr(u0), n0, N.ptr(u1), n1, seed,
                                        N.stream()), "step_prologue")
        torch.cuda.synchronize()
        row = (1 + launch) % rows
        assert torch.equal(hyper, table[row * 8:(row + 1) * 8]) and counter.tolist() == [2 + launch, 8 + launch]
        both = torch.cat([u0, u1]).cpu().numpy()
        assert both.min() >= 0.0 and both.max() < 1.0
        draws.append(both)
        for i in (0, 1, 2, 3, 4, 1001, 1002, 1003, 1004, n0 + n1 - 1):  # (both sides of the seam between the two outputs)
            want = np.float32(_philox_reference(seed, 7 + launch, i // 4)[i % 4] >> 8) * np.float32(2.0 ** -24)
            assert both[i] == want, (launch, i, both[i], want)
    allv = np.concatenate(draws)
    assert abs(allv.mean() - 0.5) < 0.02 and abs(allv.var() - 1.0 / 12.0) < 0.01
    assert not np.array_equal(draws[0], draws[1]) and len(np.unique(allv)) > 0.99 * len(allv)
    # no table: the draws only; no draws: the row only
    N.check(lib.nsamd_step_prologue(N.ptr(counter), None, 0, None, N.ptr(u0), n0, None, 0, seed, N.stream()), "step_prologue")
    hyper.fill_(-3.0)
    N.check(lib.nsamd_step_prologue(N.ptr(counter), N.ptr(table), rows, N.ptr(hyper), None, 0, None, 0, seed, N.stream()), "step_prologue")
    torch.cuda.synchronize()
    assert counter.tolist() == [7, 13] and torch.equal(hyper, table[0:8])  # (row 6 % 3 = 0)
    # rows in pinned HOST memory, written by the host right before each launch (the trainer's "ring" mode): the device reads what
    # the host wrote last, launch after launch, also when a row is rewritten with the previous launch still in flight
    ring = torch.zeros(4, 8).pin_memory()
    ring_np = ring.numpy()
    counter.zero_()
    seen = torch.zeros(64, 8, device="cuda")
    for launch in range(64):
        ring_np[launch % 4] = np.arange(8, dtype=np.float32) + 100.0 * launch
        N.check(lib.nsamd_step_prologue(N.ptr(counter), ring.data_ptr(), 4, N.ptr(hyper), None, 0, None, 0, seed, N.stream()), "step_prologue")
        seen[launch].copy_(hyper)
        if launch % 4 == 3:
            torch.cuda.synchronize()  # (the trainer's guard event: the host never laps the device)
    torch.cuda.synchronize()
    want = np.arange(8, dtype=np.float32)[None, :] + 100.0 * np.arange(64, dtype=np.float32)[:, None]
    assert np.array_equal(seen.cpu().numpy(), want)


@pytest.mark.parametrize("mode", ["ring", "table"])
def test_trainer_prologue_hands_every_iteration_the_scalars_the_host_computes(F, monkeypatch, mode):
    """trainer.HipTrainer with the device-side prologue (scalars from the ring in host memory / from the predicted table): over
    300 / 150 replayed iterations — past the end of the 256-row ring / of the 128-row table, across a `finish()` in the middle and
    a rewind of the training state (bench.py's repeated windows) — `hyper` holds after every iteration exactly the scalars the
    host computes for it (what the per-iteration upload used to carry); and graph replay trains through the same bits as eager
    launches."""
    import hashlib

    import bench

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.trainer import HipTrainer

    monkeypatch.setenv("NSAMD_STEP_PROLOGUE", mode)
    dev = torch.device("cuda")
    digests = {}
    for arm in ("graph", "eager"):
        F._SCATTER_WS.clear()
        torch.manual_seed(0)
        model = bench.build_model(dev, seed=0)
        arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
        rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
        tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=arm == "graph", use_runner=True, pool=pool)
        assert tr.prologue and tr.prologue_table == (mode == "table") and tr.prologue_ring == (mode == "ring")
        if arm == "graph":
            tr.train_iteration()
            tr.finish()
            assert tr.try_capture()
        else:
            tr.train_iteration()
            tr.finish()
            tr.warm_variants()  # (the same real iterations the capture's warm-up runs)
        state = bench.TrainingState(tr, arena, model)
        want = np.zeros(8, dtype=np.float32)
        checked, refills = 0, []
        for i in range((300 if mode == "ring" else 150) if arm == "graph" else 12):
            if arm == "graph" and i == 40:
                tr.finish()
            if arm == "graph" and i == 90:
                state.restore()
            model.set_step(tr.step)  # (what `_prologue` is about to do: the anneal exponent of this iteration)
            tr._hyper_row(want, tr.step, arena.step_counts, tr._have_pending)
            # (an iteration without a pending main-field update launches no main-field Adam and may be handed the row predicted
            #  for one with: its first two scalars are not read)
            read = slice(2, 8) if (tr.defer and not tr._have_pending) else slice(0, 8)
            before = (tr._table_base if tr._table_valid else None) if mode == "table" else 0
            tr.train_iteration()
            refills.append((i, before != (tr._table_base if mode == "table" else 0)))
            if arm == "graph" and (i < 12 or i % 7 == 0 or 38 <= i <= 44 or 85 <= i <= 95 or 125 <= i <= 135 or 250 <= i <= 262):
                torch.cuda.synchronize()
                got = tr.hyper.cpu().numpy()
                assert np.array_equal(got[read], want[read]), (i, got, want)
                checked += 1
            if i == 11:
                tr.finish()
                torch.cuda.synchronize()
                digests[arm] = tuple(hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest()
                                     for x in (arena.flat, arena.exp_avg, arena.exp_avg_sq))
        assert arm == "eager" or checked > 40
        if arm == "graph" and mode == "table":
            # a new table only when the rows run out — not after `finish()` (i = 40) and not after the rewind (i = 90), whose
            # iterations' rows are still in the table
            new_tables = [i for i, refilled in refills if refilled]
            assert len(new_tables) <= 2 and 40 not in new_tables and 90 not in new_tables, new_tables
        del tr, arena, model
    assert digests["graph"] == digests["eager"], digests
