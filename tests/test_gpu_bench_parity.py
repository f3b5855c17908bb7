"""Parity at the BENCHMARK's own size, through the BENCHMARK's own path.

bench.py measures `nerfstudio_amd.trainer.HipTrainer` (the object the `nerfacto-hip` pipeline drives under the reference's own
trainer): the explicit kernel schedule (train_step.NerfactoTrainStep) captured in hipGraphs — four
variants, proposal update x pending main-field Adam — at 4096 rays x (256, 96, 48) samples, T = 2^19 / 2^17, 100 cameras,
ray batches selected out of the HBM-resident pool by a device-side slot index. The small fixtures do not reach the parts of
that path that depend on M and T (the scatter's per-level queue capacities, the 8x coarse queues, the spill path, the
fixed-point scale), so here:

1. one proposal-UPDATE and one NON-update iteration REPLAYED from the captured graphs (deferred Adam pending) against the
   CPU oracle evaluated on the same rays / jitter / parameters: rgb <= 1e-4 L-inf (north_star), the three losses, every
   gradient tensor (the main table PER LEVEL) as close to the float64 evaluation of the same graph as the fp32 reference
   or a half-ulp perturbation of the parameters is (the gradient is ill-conditioned at that level), no scatter record on
   an unordered path;
2. six iterations replayed from the graphs against the same six launched eagerly (Adam in order, one stream):
   parameters and both Adam moments equal BIT FOR BIT (DESIGN §4.2 claims "same bits"; round 2 compared a loss rounded to
   six decimals);
3. the zero-gradient gating of the proposal chains (include/nsamd.h): gated == ungated, bit for bit, on an all-zero, a
   one-ray and a NaN-density upstream gradient.
"""
import numpy as np
import pytest
import torch

from oracle import nerfacto_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from nerfstudio_amd import _native, functional

    _native.load()
    functional.DIRECT_GRAD = True
    yield functional
    functional.DIRECT_GRAD = False


def _bench_trainer(F, params, cfg, use_graph, seed=1000):
    """bench.py's own workload (model with the oracle's parameters loaded, arena, ray pool) on the product's trainer."""
    import bench
    from test_gpu_kernels import _hip_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.trainer import HipTrainer

    dev = torch.device("cuda")
    model = _hip_model(cfg, params)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    rb, batch, pool = bench.synthetic_batch(dev, seed=seed)
    trainer = HipTrainer(model, arena, rb, batch, world=1, use_graph=use_graph, use_runner=True, pool=pool)
    trainer.draw_jitter = False  # the jitter buffer is filled by the test (the oracle gets the same draws)
    return bench, model, arena, trainer


def _oracle_params(model, keys):
    sd = model.state_dict()
    out = {}
    for k in keys:
        src = k.replace(".encoding.hash_table", ".mlp_base.0.hash_table") if k.startswith("proposal_networks") else k
        out[k] = sd[src if src in sd else k].detach().cpu().clone()
    return out


def _grads_by_oracle_name(model, keys):
    named = dict(model.named_parameters())
    out = {}
    for k in keys:
        cand = [k, k.replace(".encoding.hash_table", ".mlp_base.0.hash_table")]
        p = next((named[c] for c in cand if c in named), None)
        assert p is not None, f"no parameter for {k} in {sorted(named)[:6]}..."
        out[k] = p.grad.detach().cpu().numpy().copy()
    return out


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


@pytest.mark.parametrize("init", ["default", "n(0,0.3)"])
def test_bench_configuration_parity_through_replayed_graph(F, init):
    """(1) of the module docstring. `default` = the benchmark's own initial state (tables U(-1e-3, 1e-3): near-uniform
    densities); `n(0,0.3)` = tables ~ N(0, 0.3) so that densities, weights and resampling are far from uniform."""
    cfg = orc.NerfactoCfg()  # T = 2^19 / 2^17, L = 16 / 5, 100 cameras: BASELINE configs[1]
    assert cfg.num_images == 100 and cfg.main_grid.log2_hashmap_size == 19
    params = orc.init_params(cfg, seed=0, table_std=None if init == "default" else 0.3)
    keys = list(params.keys())
    F._SCATTER_WS.clear()
    bench, model, arena, tr = _bench_trainer(F, params, cfg, use_graph=True)
    n = bench.RAYS_PER_GPU
    rs = np.random.RandomState(11)
    tr.runner.jitter.copy_(torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)))
    tr.capture()  # two eager warm-up iterations (parameters move), then the four variants
    assert tr.defer and set(tr.graphs) == {("all", u, p) for u in (True, False) for p in (True, False)}
    # one replayed iteration first, so that the iterations under test run with the main-field Adam PENDING (the steady
    # state of the benched schedule: the update of iteration k-1 is the first node of iteration k's graph)
    tr.train_iteration()
    ps = model.proposal_sampler
    pa, pb = arena.groups["proposal_networks"]
    checked = []
    for forced in (True, False):  # a proposal-update iteration, then a non-update iteration
        jit = rs.uniform(0, 1, (3, n)).astype(np.float32)
        tr.runner.jitter.copy_(torch.from_numpy(jit))
        assert tr._pending_main
        slot = tr.step % bench.BATCH_SLOTS
        ps.updated_this_step = lambda forced=forced: forced  # choose the schedule variant
        before = arena.flat[pa:pb].clone()  # the proposal parameters this iteration's forward sees
        tr.train_iteration()  # replays graph ("all", forced, pending=True)
        del ps.updated_this_step
        torch.cuda.synchronize()
        got_rgb = tr.runner.outputs()["rgb"].cpu().numpy()
        got_losses = {k: float(v) for k, v in tr.runner.loss_dict().items()}
        grads = _grads_by_oracle_name(model, keys)
        # ---- the oracle on the same rays / jitter / parameters: the main field as it stands (its Adam of THIS iteration is
        # still pending), the proposal networks as they were before the proposal Adam at the end of the replayed graph ----
        after = arena.flat[pa:pb].clone()
        arena.flat[pa:pb].copy_(before)
        base_params = _oracle_params(model, keys)
        arena.flat[pa:pb].copy_(after)
        if forced:
            assert not torch.equal(before, after), "the update iteration's graph steps the proposal networks"
        else:
            assert torch.equal(before, after), "a non-update iteration leaves the proposal networks alone"
        o, d, cam, tgt = (torch.from_numpy(a) for a in bench.synthetic_rays(1000 + slot))
        j = [torch.from_numpy(jit[i])[:, None] for i in range(3)]

        def oracle(dtype, perturb=None):
            prm = {k: v.detach().to(dtype).clone() for k, v in base_params.items()}
            if perturb is not None:  # every parameter moved by half an ulp of its fp32 value, random sign
                gen = torch.Generator().manual_seed(perturb)
                prm = {k: v * (1.0 + (torch.randint(0, 2, v.shape, generator=gen).to(dtype) * 2 - 1) * 2.0 ** -24) for k, v in prm.items()}
            prm = {k: v.requires_grad_(True) for k, v in prm.items()}
            res = orc.nerfacto_forward(prm, cfg, o.to(dtype), d.to(dtype), cam[:, 0], [x.to(dtype) for x in j], training=True,
                                       anneal=ps._anneal, proposal_requires_grad=forced)
            losses = orc.nerfacto_losses(res, tgt.to(dtype), cfg)
            res["density"].retain_grad()      # upstream gradients of the main field: what render_train_bwd produces
            res["rgb_samples"].retain_grad()
            sum(losses.values()).backward()
            return prm, res, losses

        oparams, out, ld = oracle(torch.float32)          # the reference's own arithmetic (torch path, fp32)
        truth, out64, _ = oracle(torch.float64)           # the same graph in float64: the arbiter for the gradients
        # per-sample forward values and the upstream gradients the field backward consumes (diagnostics + assertions)
        S_f = tr.runner.counts[-1]
        stage = {}
        for name, got_t, key, is_grad in (("density", tr.runner.f_dens, "density", False), ("rgb samples", tr.runner.f_rgb, "rgb_samples", False),
                                          ("dL/d density", tr.runner.d_dens_main, "density", True),
                                          ("dL/d rgb samples", tr.runner.d_rgb_s, "rgb_samples", True)):
            g = got_t.detach().cpu().double().numpy().reshape(-1)
            r32 = (out[key].grad if is_grad else out[key]).detach().double().numpy().reshape(-1)
            r64 = (out64[key].grad if is_grad else out64[key]).detach().numpy().reshape(-1)
            stage[name] = (_rel_l2(g, r64), _rel_l2(r32, r64))
        print(f"\n[updated={forced}] per-sample stages, rel-L2 vs float64 (kernels, fp32 reference): " +
              ", ".join(f"{k}: {v[0]:.1e} / {v[1]:.1e}" for k, v in stage.items()))
        err = float(np.abs(got_rgb - out["rgb"].detach().numpy()).max())
        assert err <= 1e-4, f"rgb L-inf {err:.2e} (updated={forced})"
        for k in ("rgb_loss", "interlevel_loss", "distortion_loss"):
            np.testing.assert_allclose(got_losses[k], float(ld[k]), rtol=5e-4, atol=1e-9, err_msg=f"{k} (updated={forced})")
        # Gradients, per tensor (the main table per LEVEL: queue capacities and the fixed-point scale are per level).
        # The gradient of this graph is ILL-CONDITIONED at the fp32 level: it is piecewise smooth in the parameters (ReLU
        # kinks of 12.6 M hidden units per step, searchsorted edges of the resampling), and moving every parameter by HALF AN
        # ULP of its fp32 value moves the exact (float64) gradient of the head / table tensors by 2e-4 ... 1e-3 (relative
        # L2; measured with the oracle, profiles/r03_gradient_conditioning.txt) — as much as the reference's own fp32
        # evaluation is away from float64. A bound in the spirit of backward error analysis is therefore the honest one:
        # with e_ref = |fp32 reference - float64| and e_cond = |float64 at half-ulp-perturbed parameters - float64|, the
        # kernels must satisfy |kernels - float64| <= 4 max(e_ref, e_cond) (floor below) and stay within 2e-2 of the fp32
        # reference. The perturbed pass is only evaluated when a tensor is outside 2 e_ref. (The floor: the proposal
        # networks' gradient comes through lossfun_outer's clip(w - w_outer, 0) — a difference of nearly equal numbers,
        # which amplifies the 1e-6-level fp32 differences of the forward; measured 1.3-1.6e-4 on the first proposal network
        # against 2e-5 for the fp32 reference.) That the MLP backward kernel itself carries no such error is shown on
        # identical inputs by test_field_mlp_backward_at_bench_size_vs_float64.
        # The floor. A ReLU whose pre-activation lies within the forward discrepancy of zero comes out on either side,
        # and each such event moves its sample's gradient by a finite amount. With near-uniform default tables there is
        # essentially no such event (floor 5e-4: everything sits at the reference's own 3e-4). With N(0, 0.3) tables the
        # forward itself is ill-conditioned — a 1e-7 relative difference of a far sample's bin edge is 2e-4 of a cell on
        # the 2048-resolution level, where neighbouring table entries differ by O(0.4): per-sample densities agree with
        # float64 to 1e-5 (kernels) / 4e-6 (fp32 reference), rgb to 3e-7, the upstream gradients to the reference's own
        # 1e-4 (printed above) — and 25 M head pre-activations ~ N(0, 0.3) meet a 1e-6 discrepancy ~70 times per step:
        # measured 0.6-1.1e-3 on the head / geo tensors (3e-3 allowed). The MLP backward kernel on IDENTICAL inputs with
        # such samples removed is within 4e-7 of float64 (test_field_mlp_backward_at_bench_size_vs_float64).
        floor = 5e-4 if init == "default" else 3e-3
        report, bad, cond = {}, [], {}

        def conditioning():
            if not cond:
                cond.update({k: v.grad.numpy() for k, v in oracle(torch.float64, perturb=17)[0].items() if v.grad is not None})
            return cond

        def check(name, got, ref32, ref64, key, sl=slice(None), e_family=0.0):
            if np.abs(ref64).max() == 0.0:  # an exact zero must be an exact zero
                report[name] = (float(np.abs(got).max()), 0.0, 0.0, 0.0)
                if np.abs(got).max() != 0.0:
                    bad.append(name)
                return
            e_gpu, e_ref, e_pair = _rel_l2(got, ref64), _rel_l2(ref32, ref64), _rel_l2(got, ref32)
            e_cond = 0.0
            if not e_gpu <= max(2.0 * e_ref, floor):
                e_cond = _rel_l2(conditioning()[key][sl], ref64)
            report[name] = (e_gpu, e_ref, e_pair, e_cond)
            if not (e_gpu <= max(4.0 * max(e_ref, e_cond), 3.0 * e_family, floor) and e_pair <= 2e-2):
                bad.append(name)

        for k in keys:
            ref = oparams[k].grad
            if k.startswith("proposal_networks") and not forced:
                assert ref is None or float(ref.abs().max()) == 0.0
                continue
            r32, r64 = ref.numpy(), truth[k].grad.numpy()
            if k == "field.mlp_base.model.0.hash_table":
                T = 1 << cfg.main_grid.log2_hashmap_size
                # (the levels of one table are a family: which of them a given ReLU / resampling event hits hardest is
                # chance, so each level is also allowed 3x the family's median reference error)
                levels = [slice(lvl * T, (lvl + 1) * T) for lvl in range(cfg.main_grid.num_levels)]
                family = float(np.median([_rel_l2(r32[sl], r64[sl]) for sl in levels if np.abs(r64[sl]).max() > 0] or [0.0]))
                for lvl, sl in enumerate(levels):
                    check(f"{k}[level {lvl}]", grads[k][sl], r32[sl], r64[sl], k, sl, family)
            else:
                check(k, grads[k], r32, r64, k)
        worst = max(report.items(), key=lambda kv: kv[1][0])
        table = "\n".join(f"  {n}: gpu-f64 {v[0]:.2e}  ref32-f64 {v[1]:.2e}  gpu-ref32 {v[2]:.2e}  f64(half-ulp)-f64 {v[3]:.2e}"
                          for n, v in report.items())
        print(f"[updated={forced}] gradients, rel-L2:\n{table}")
        assert not bad, f"updated={forced}: {bad}\n{table}"
        checked.append((forced, err, worst))
    for ws in F._SCATTER_WS.values():
        ev = F.scatter_events(ws)
        assert ev[1] == 0 and ev[2] == 0, f"scatter records on an unordered path / lost: {ev}"
    print("\nbench-size parity [init %s] (updated, rgb L-inf, worst tensor: (gpu-f64, ref32-f64, gpu-ref32) rel-L2):" % init, checked)


def test_graph_replay_trains_through_the_same_bits_as_eager_launches(F):
    """(2) of the module docstring: K = 6 iterations at the benchmark configuration, replayed from the captured hipGraphs
    with the main-field Adam deferred, against eager launches with Adam in order on one stream. torch.equal on the
    parameter arena and both moments."""
    cfg = orc.NerfactoCfg()
    K = 6
    rs = np.random.RandomState(3)
    jit = rs.uniform(0, 1, (K + 1, 3, 4096)).astype(np.float32)
    states = []
    for use_graph in (True, False):
        F._SCATTER_WS.clear()
        params = orc.init_params(cfg, seed=0)
        bench, model, arena, tr = _bench_trainer(F, params, cfg, use_graph=use_graph)
        tr.runner.jitter.copy_(torch.from_numpy(jit[K]))
        if use_graph:
            tr.capture()
            assert tr.defer and len(tr.graphs) == 4
        else:
            assert not tr.defer
            tr.runner.side_stream = None  # one stream, kernels in program order
            tr.warm_variants()  # the same two warm-up iterations the capture runs
        for k in range(K):
            tr.runner.jitter.copy_(torch.from_numpy(jit[k]))
            tr.train_iteration()
        tr.finish()
        torch.cuda.synchronize()
        states.append((arena.flat.clone(), arena.exp_avg.clone(), arena.exp_avg_sq.clone(),
                       float(sum(tr.runner.loss_dict().values()))))
        del tr, arena, model
    g, e = states
    assert np.isfinite(g[3]) and g[3] == e[3], (g[3], e[3])
    for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq"), g[:3], e[:3]):
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} elements differ between graph replay and eager"


@pytest.mark.parametrize("case", ["all_zero", "one_ray", "nan_density"])
def test_gated_proposal_chain_equals_ungated(F, case):
    """(3): the proposal networks' backward with the zero-gradient gating against the ungated entry points, bit for bit —
    all-zero upstream gradient (flags stay clear, gradients stay the zero fill, `denc` untouched), one ray with gradient
    (flag raised, everything equal, the per-ray / per-chunk / per-workgroup early-outs all taken), and a NaN density under
    a zero gradient (the full path must run: 0 * NaN = NaN as in autograd)."""
    from test_gpu_kernels import _hip_model, small_cfg

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = small_cfg(12, 10, 6)
    n = 700
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=8)
    rs = np.random.RandomState(2)
    jit = torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    results = []
    for gated in (True, False):
        F._SCATTER_WS.clear()
        model = _hip_model(cfg, orc.init_params(cfg, seed=7, table_std=0.4))
        arena = ParamArena(model.get_param_groups_ordered())
        step = NerfactoTrainStep(model, n, torch.device("cuda"))
        step.gate_proposals = gated
        step.side_stream = None  # (the interlevel gradient is replaced below, AFTER the losses launch)
        step.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
        step.jitter.copy_(jit)
        step.anneal_dev.fill_(1.0)
        arena.zero_grad()
        step.forward_and_losses(True, draw_jitter=False)
        for lvl in range(step.n_prop):  # replace the interlevel gradient by the case's
            step.dw_prop[lvl].zero_()
            step.p_denc[lvl].fill_(123.0)  # sentinel: a gated chain with a clear flag must not touch it
            if case == "one_ray":
                step.dw_prop[lvl][137] = torch.linspace(-1e-3, 2e-3, step.counts[lvl], device="cuda")
            elif case == "nan_density":
                step.p_dens[lvl][5 * step.counts[lvl] + 3] = float("nan")
        arena.zero_grad(["proposal_networks"])
        step.backward_proposals()
        torch.cuda.synchronize()
        a, b = arena.groups["proposal_networks"]
        results.append((arena.grad[a:b].clone(), [t.clone() for t in step.p_denc], [t.clone() for t in step.p_ddens],
                        step.prop_gates.clone(), [t.clone() for t in step.prop_ray_masks]))
    (g_grad, g_denc, g_ddens, flags, masks), (u_grad, u_denc, u_ddens, _, _) = results

    def same(a, b):  # bit equality of every non-NaN value, NaN exactly where the other has NaN
        na, nb = torch.isnan(a), torch.isnan(b)
        return torch.equal(na, nb) and torch.equal(a[~na].view(torch.int32), b[~nb].view(torch.int32))

    assert same(g_grad, u_grad), f"{int((g_grad.view(torch.int32) != u_grad.view(torch.int32)).sum())} gradient words differ"
    for lvl in range(2):
        assert same(g_ddens[lvl], u_ddens[lvl])
        if case == "one_ray":  # rays without upstream gradient take the early-out: exact zeros
            z = g_ddens[lvl].view(n, -1).clone()
            z[137] = 0.0
            assert float(z.abs().max()) == 0.0 and float(g_ddens[lvl].view(n, -1)[137].abs().max()) > 0.0
    if case == "all_zero":
        assert int(flags[0]) == 0 and int(flags[4]) == 0 and not any(bool(m.any()) for m in masks)
        assert float(g_grad.abs().max()) == 0.0
        for lvl in range(2):
            assert bool((g_denc[lvl] == 123.0).all()), "a gated chain with a clear flag must not write denc"
    else:
        assert int(flags[0]) == 1 and int(flags[4]) == 1
        for lvl in range(2):
            # the per-ray mask marks exactly the carrying ray; the feature gradients of its samples are what the scatter
            # reads (chunks without a marked ray are not even written: the sentinel survives there)
            carrying = 137 if case == "one_ray" else 5
            expect = torch.zeros(n, dtype=torch.uint8, device="cuda")
            expect[carrying] = 1
            assert torch.equal(masks[lvl], expect)
            S_l = step.counts[lvl]
            rows = slice(carrying * S_l, (carrying + 1) * S_l)
            assert same(g_denc[lvl][:, rows].contiguous(), u_denc[lvl][:, rows].contiguous())
            assert bool((g_denc[lvl] == 123.0).any()), "chunks without a marked ray must stay unwritten"
        if case == "one_ray":
            assert float(g_grad.abs().max()) > 0.0
        else:
            assert bool(torch.isnan(g_grad).any()), "a NaN density must reach the gradients as it does in autograd"


@pytest.mark.parametrize("live", [0.03, 0.4, 1.0])
def test_merged_proposal_levels_backward_equals_level_by_level(F, live):
    """nsamd_proposal_levels_bwd (every stage of the two proposal levels' backward chains as one launch across the levels)
    against the per-level gated entry points, bit for bit: weight gradients, table gradients, feature and density gradients,
    flags and ray masks — with 3 %, 40 % and all of the rays carrying interlevel gradient (different rays per level)."""
    from test_gpu_kernels import _hip_model, small_cfg

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = small_cfg(12, 10, 6)
    n = 1500
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=11)
    rs = np.random.RandomState(4)
    jit = torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    results = []
    for merged in (True, False):
        F._SCATTER_WS.clear()
        model = _hip_model(cfg, orc.init_params(cfg, seed=7, table_std=0.4))
        arena = ParamArena(model.get_param_groups_ordered())
        step = NerfactoTrainStep(model, n, torch.device("cuda"))
        assert step.gate_proposals
        step.merge_prop_levels = merged
        step.side_stream = None
        step.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
        step.jitter.copy_(jit)
        step.anneal_dev.fill_(1.0)
        arena.zero_grad()
        step.forward_and_losses(True, draw_jitter=False)
        g = torch.Generator(device="cuda").manual_seed(5)
        for lvl in range(step.n_prop):
            S_l = step.counts[lvl]
            dw = torch.randn(n, S_l, device="cuda", generator=g) * 1e-3
            keep = torch.rand(n, device="cuda", generator=g) < live
            step.dw_prop[lvl].copy_((dw * keep[:, None]).view_as(step.dw_prop[lvl]))
            step.p_denc[lvl].fill_(123.0)
        for _ in range(2):  # twice: the workspaces' self-cleaning state must serve the next call
            arena.zero_grad(["proposal_networks"])
            if step.gates_precleared:
                step.prop_gates.zero_()
            step.backward_proposals()
        torch.cuda.synchronize()
        a, b = arena.groups["proposal_networks"]
        results.append((arena.grad[a:b].clone(), [t.clone() for t in step.p_denc], [t.clone() for t in step.p_ddens],
                        step.prop_gates.clone(), [t.clone() for t in step.prop_ray_masks]))
    (m_grad, m_denc, m_ddens, m_flags, m_masks), (s_grad, s_denc, s_ddens, s_flags, s_masks) = results
    assert float(m_grad.abs().max()) > 0.0
    assert torch.equal(m_grad.view(torch.int32), s_grad.view(torch.int32)), \
        f"{int((m_grad.view(torch.int32) != s_grad.view(torch.int32)).sum())} gradient words differ"
    assert torch.equal(m_flags, s_flags)
    for lvl in range(2):
        assert torch.equal(m_masks[lvl], s_masks[lvl]) and int(m_masks[lvl].sum()) > 0
        assert torch.equal(m_ddens[lvl].view(torch.int32), s_ddens[lvl].view(torch.int32))
        assert torch.equal(m_denc[lvl].view(torch.int32), s_denc[lvl].view(torch.int32))


def test_field_mlp_backward_at_bench_size_vs_float64(F):
    """The main-field MLP kernels alone at M = 196 608 (4096 rays x 48), fed the training step's own buffers (encoded
    features, selector, directions, camera ids, upstream dL/d density and dL/d rgb): every weight gradient, the
    appearance-embedding gradient and the encoded-feature gradient against a float64 evaluation of the same MLPs on the
    same inputs, next to what torch's fp32 CPU evaluation achieves against it. A ReLU whose pre-activation is within 1e-5
    of zero may come out on either side in fp32 (MFMA k-order vs BLAS blocking) and then moves its sample's gradient by a
    finite amount — no arithmetic error, and no statement about the kernel: the samples that own such a unit (a fraction
    of a percent) are taken out of BOTH sides by zeroing their upstream gradients, so what is left measures arithmetic."""
    from test_gpu_kernels import _hip_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0, table_std=0.3)
    model = _hip_model(cfg, params)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    n = 4096
    step = NerfactoTrainStep(model, n, torch.device("cuda"))
    step.side_stream = None
    step.keep_denc = True  # the default launch (field backward + scatter records) leaves `f_denc` unwritten otherwise
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=21)
    step.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
    rs = np.random.RandomState(5)
    step.jitter.copy_(torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)))
    step.anneal_dev.fill_(1.0)
    arena.zero_grad(skip=step.written_params())
    step.forward_backward(updated=False, draw_jitter=False)
    torch.cuda.synchronize()
    S = step.counts[-1]
    enc = step.f_enc.t().contiguous().cpu()          # [M, 32]
    sel = step.f_sel.cpu()
    dirs = d.repeat_interleave(S, dim=0)             # per-sample view directions
    cams = cam.repeat_interleave(S, dim=0)
    keys = [k for k in params if k.startswith("field.") and "hash_table" not in k]

    def forward(prm, x, dtype):
        pre0 = x @ prm["field.mlp_base.model.1.layers.0.weight"].t() + prm["field.mlp_base.model.1.layers.0.bias"]
        h = torch.relu(pre0) @ prm["field.mlp_base.model.1.layers.1.weight"].t() + prm["field.mlp_base.model.1.layers.1.bias"]
        density = cfg.average_init_density * orc.trunc_exp(h[:, 0]) * sel.to(dtype)
        sh = orc.sh_levels4((dirs.to(dtype) + 1.0) / 2.0)
        app = prm["field.embedding_appearance.embedding.weight"][cams]
        p0 = torch.cat([sh, h[:, 1:], app], dim=-1) @ prm["field.mlp_head.layers.0.weight"].t() + prm["field.mlp_head.layers.0.bias"]
        p1 = torch.relu(p0) @ prm["field.mlp_head.layers.1.weight"].t() + prm["field.mlp_head.layers.1.bias"]
        rgb = torch.sigmoid(torch.relu(p1) @ prm["field.mlp_head.layers.2.weight"].t() + prm["field.mlp_head.layers.2.bias"])
        return density, rgb, (pre0, p0, p1)

    with torch.no_grad():
        pres = forward({k: params[k].double() for k in keys}, enc.double(), torch.float64)[2]
        ambiguous = torch.zeros(enc.shape[0], dtype=torch.bool)
        for p_ in pres:
            ambiguous |= (p_.abs() < 1e-5).any(dim=1)
    frac = float(ambiguous.float().mean())
    assert 0 < int(ambiguous.sum()) and frac < 0.02, frac
    amb_dev = ambiguous.cuda()
    step.d_dens_main[amb_dev] = 0.0
    step.d_rgb_s[amb_dev] = 0.0
    g_dens, g_rgb = step.d_dens_main.cpu(), step.d_rgb_s.cpu()
    arena.zero_grad(["fields"], skip=step.written_params())
    step.backward_field_and_table()  # the field MLP backward (+ table scatter) on the edited upstream gradients
    torch.cuda.synchronize()
    got = _grads_by_oracle_name(model, keys)
    got["denc"] = step.f_denc.t().contiguous().cpu().numpy()

    def evaluate(dtype):
        prm = {k: params[k].detach().to(dtype).clone().requires_grad_(True) for k in keys}
        x = enc.to(dtype).clone().requires_grad_(True)
        density, rgb, _ = forward(prm, x, dtype)
        ((density * g_dens.to(dtype)).sum() + (rgb * g_rgb.to(dtype)).sum()).backward()
        out = {k: v.grad.numpy() for k, v in prm.items()}
        out["denc"] = x.grad.numpy()
        return out

    r32, r64 = evaluate(torch.float32), evaluate(torch.float64)
    rows, bad = [], []
    for k in list(keys) + ["denc"]:
        e_gpu, e_ref = _rel_l2(got[k], r64[k]), _rel_l2(r32[k], r64[k])
        rows.append(f"  {k}: gpu-f64 {e_gpu:.2e}  cpu32-f64 {e_ref:.2e}")
        if not e_gpu <= max(3.0 * e_ref, 2e-6):
            bad.append(k)
    print(f"\nfield MLP backward at M = 196608 against float64 ({int(ambiguous.sum())} samples = {100 * frac:.2f} % with a "
          "ReLU pre-activation within 1e-5 of zero excluded):\n" + "\n".join(rows))
    assert not bad, f"{bad}\n" + "\n".join(rows)


@pytest.mark.parametrize("init", ["default", "n(0,0.3)"])
def test_backward_that_emits_the_scatter_records_equals_the_two_launches(F, init):
    """nsamd_field_mlp_bwd_scatter (the main field's backward emits the table scatter's pass-1 records from its registers)
    against nsamd_field_mlp_bwd + nsamd_hashgrid_encode_bwd_set at the benchmark's size: the same records reach the same
    order-independent fixed-point sums, so every MLP gradient is bit-equal and the table gradient differs at most by where
    the fixed-point truncation sits (the queue capacity sets the headroom): <= 1e-6 of the level's largest entry. No record
    on an unordered path, and two calls give the same bits."""
    from test_gpu_kernels import _hip_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep
    import bench

    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0, table_std=None if init == "default" else 0.3)
    F._SCATTER_WS.clear()
    dev = torch.device("cuda")
    model = _hip_model(cfg, params)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    n = bench.RAYS_PER_GPU
    o, d, cam, tgt = (torch.from_numpy(a).to(dev) for a in bench.synthetic_rays(1003))
    r = NerfactoTrainStep(model, n, dev)
    r.side_stream = None
    r.set_batch(o, d, cam[:, 0], tgt)
    rs = np.random.RandomState(4)
    r.jitter.copy_(torch.from_numpy(rs.uniform(0, 1, (3, n)).astype(np.float32)))
    r.forward_and_losses(False, draw_jitter=False)
    a, b = arena.groups["fields"]
    table = model.field.mlp_base.encoding.hash_table
    off = next(o_ for p_, o_ in zip(arena.params, arena.offsets) if p_ is table)
    T, out = 1 << 19, {}
    for mode in ("two", "fused", "fused_again"):
        r.fuse_route = mode != "two"
        arena.zero_grad(["fields"])
        arena.grad[off:off + table.numel()].fill_(float("nan"))  # write-only: every entry must be written
        r.backward_main()
        torch.cuda.synchronize()
        out[mode] = arena.grad[a:b].clone()
    assert any(k[3] == "producer" for k in F._SCATTER_WS), "the fused entry point was not taken"
    two, fused, again = out["two"], out["fused"], out["fused_again"]
    assert not torch.isnan(fused).any()
    assert torch.equal(fused, again), "two calls of the fused backward differ"
    lo, hi = off - a, off - a + table.numel()
    assert torch.equal(two[:lo], fused[:lo]) and torch.equal(two[hi:], fused[hi:]), "MLP / embedding gradients differ"
    t2, tf = two[lo:hi].view(16, T, 2), fused[lo:hi].view(16, T, 2)
    worst = 0.0
    for l in range(16):
        m = float(t2[l].abs().max())
        assert m > 0
        e = float((t2[l] - tf[l]).abs().max()) / m
        worst = max(worst, e)
        assert e <= 1e-6, f"level {l}: table gradient differs by {e:.2e} of the level's largest entry"
    for key, ws in F._SCATTER_WS.items():
        ev = F.scatter_events(ws)
        assert ev[1] == 0 and ev[2] == 0, f"scatter records on an unordered path / lost: {key[3]} {ev}"
    spilled = {k[3]: F.scatter_events(ws)[0] for k, ws in F._SCATTER_WS.items()}
    print(f"\nfused route [{init}]: worst table-gradient difference {worst:.2e} of a level's maximum; spilled records {spilled}")


def test_field_backward_with_compute_units_left_free_gives_the_same_gradients(F):
    """nsamd_field_mlp_bwd_reserve_cus: the persistent field backward on fewer workgroups (update iterations leave compute
    units to the proposal levels' chains; -1 = one more sweep: 220 workgroups at 196 608 points, 36 = the same here, 100 = an
    uneven last sweep). The table gradient's fixed-point sums do not depend on who emitted a record — bit-equal up to the
    headroom the queue capacity sets (<= 1e-6 of the level's largest entry) —, the MLP weight gradients are the same partial
    sums added in another fixed order (<= 1e-6 relative L2), and a repeated call gives the same bits."""
    from test_gpu_kernels import _hip_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep
    import bench

    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0, table_std=0.3)
    F._SCATTER_WS.clear()
    dev = torch.device("cuda")
    model = _hip_model(cfg, params)
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    n = bench.RAYS_PER_GPU
    o, d, cam, tgt = (torch.from_numpy(a).to(dev) for a in bench.synthetic_rays(1003))
    r = NerfactoTrainStep(model, n, dev)
    r.side_stream = None
    r.set_batch(o, d, cam[:, 0], tgt)
    r.jitter.copy_(torch.from_numpy(np.random.RandomState(4).uniform(0, 1, (3, n)).astype(np.float32)))
    r.forward_and_losses(False, draw_jitter=False)
    a, b = arena.groups["fields"]
    table = model.field.mlp_base.encoding.hash_table
    off = next(o_ for p_, o_ in zip(arena.params, arena.offsets) if p_ is table)
    lo, hi = off - a, off - a + table.numel()
    out = {}
    for mode in (0, -1, 36, 100, "again"):
        r.bwd_reserve_cus = -1 if mode == "again" else mode
        arena.zero_grad(["fields"])
        arena.grad[off:off + table.numel()].fill_(float("nan"))
        r.backward_main(reserve=True)
        torch.cuda.synchronize()
        out[mode] = arena.grad[a:b].clone()
    from nerfstudio_amd import _native as N

    assert N.load().nsamd_field_mlp_bwd_reserve_cus(0) == 0, "the reservation must not outlive the call"
    ref = out[0]
    assert not torch.isnan(ref).any() and float(ref.abs().max()) > 0
    assert torch.equal(out[-1], out["again"]) and torch.equal(out[-1], out[36])
    T = 1 << 19
    for mode in (-1, 100):
        g = out[mode]
        t0, t1 = ref[lo:hi].view(16, T, 2), g[lo:hi].view(16, T, 2)
        for l in range(16):
            m = float(t0[l].abs().max())
            assert float((t0[l] - t1[l]).abs().max()) <= 1e-6 * m, f"reserve {mode}: table level {l}"
        for sl in (slice(0, lo), slice(hi, None)):
            x, y = ref[sl].double(), g[sl].double()
            assert float((x - y).norm() / x.norm()) <= 1e-6, f"reserve {mode}: MLP / embedding gradients"
