"""The training iteration's merged launches against the launches they replace, BIT FOR BIT:

* nsamd_select_bins = nsamd_select_batch + nsamd_piecewise_bins;
* the main field's weight-gradient reduce riding the table scatter's apply pass = its own launch;
* nsamd_train_loss_values (the five floats a trainer logs) against float64 sums of the per-ray terms;
* nsamd_step_prologue (the step's scalars and Philox draws as the first node of the replayed graph).

(Round 5's fully merged per-ray launches — nsamd_render_losses_train, nsamd_proposal_sampler — were bit-identical to the launches
they replaced and measured slower; they and their tests live in nerfstudio_amd/csrc/experiments/rounds2to5_opt_in_variants.patch.)"""
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


def _runner(cfg, n, seed, background, gate=True):
    from test_gpu_kernels import _hip_model

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.train_step import NerfactoTrainStep

    model = _hip_model(cfg, orc.init_params(cfg, seed=seed, table_std=0.4))
    model.config.background_color = background
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    r = NerfactoTrainStep(model, n, torch.device("cuda"))
    r.gate_proposals = gate
    r.side_stream = None
    return model, arena, r


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
    """bench.py's trainer (captured graphs, deferred main-field Adam, pool of batches, bench-size batch) with batch selection +
    initial bins as ONE launch (the default) against the same trainer on the two launches: identical parameter and moment bits
    after 15 iterations of both update kinds, replayed from the captured graphs."""
    import hashlib

    import bench

    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.trainer import HipTrainer

    digests = {}
    dev = torch.device("cuda")
    for arm, env in (("merged", {}), ("separate", {"NSAMD_FUSE_SELECT": "0"})):
        monkeypatch.delenv("NSAMD_FUSE_SELECT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        F._SCATTER_WS.clear()
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        model = bench.build_model(dev, seed=0)
        arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
        rb, batch, pool = bench.synthetic_batch(dev, seed=1000)
        tr = HipTrainer(model, arena, rb, batch, world=1, use_graph=True, use_runner=True, pool=pool)
        assert tr.runner.fuse_select == (arm == "merged")
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


def test_loss_values_launch_against_float64_sums(F):
    """The five floats a trainer reads every iteration (rgb / interlevel / distortion loss, psnr, distortion metric), written by
    nsamd_train_loss_values behind the compositing and loss launches: the float64 sums of the per-ray terms to 1e-6, and the same
    bits on a second call (fixed summation order)."""
    from test_gpu_kernels import small_cfg

    cfg = small_cfg(12, 10, 6)
    n = 1000
    o, d, cam, tgt = orc.synthetic_rays(n, cfg.num_images, seed=9)
    jit = torch.from_numpy(np.random.RandomState(2).uniform(0, 1, (3, n)).astype(np.float32)).cuda()
    F._SCATTER_WS.clear()
    model, arena, r = _runner(cfg, n, 14, "last_sample")
    r.want_loss_vals = True
    r.set_batch(o.cuda(), d.cuda(), cam.cuda(), tgt.cuda())
    r.jitter.copy_(jit)
    vals = []
    for _ in range(2):
        arena.zero_grad(skip=r.written_params())
        r.forward_and_losses(True, draw_jitter=False)
        torch.cuda.synchronize()
        assert r._loss_vals_fresh
        ld = r.loss_dict()
        first = ld["rgb_loss"].clone()
        r.loss_vals[:3].fill_(-1.0)  # the dictionary holds its own copy: a later iteration does not rewrite it
        assert float(ld["rgb_loss"]) == float(first)
        r.forward_and_losses(True, draw_jitter=False)
        torch.cuda.synchronize()
        ld = r.loss_dict()
        mc, S = model.config, r.counts[-1]
        ref = {"rgb_loss": float(r.sq_err.double().sum()) / (3 * n),
               "interlevel_loss": mc.interlevel_loss_mult * float(sum(p.double().sum() for p in r.inter_per_ray)) / (n * S),
               "distortion_loss": mc.distortion_loss_mult * float(r.dist_per_ray.double().sum()) / n}
        for k, v in ref.items():
            assert abs(float(ld[k]) - v) <= 1e-6 * max(abs(v), 1e-6) + 1e-9, (k, float(ld[k]), v)
        assert abs(float(r.loss_vals[3]) + 10.0 * np.log10(ref["rgb_loss"])) <= 1e-4
        assert abs(float(r.loss_vals[4]) - float(r.dist_per_ray.double().sum()) / n) <= 1e-6 * float(r.loss_vals[4]) + 1e-9
        assert abs(float(r.loss_vals[5]) - sum(ref.values())) <= 2e-6 * sum(ref.values())
        vals.append(r.loss_vals[:8].clone())
    _same(vals[0], vals[1], "loss values of two calls on the same per-ray terms")


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
            # a new table when the rows run out or the state is rewound to BEFORE the table's first row — not after `finish()`
            # (i = 40), whose next iteration is served by the row predicted for an iteration with a pending update
            new_tables = [i for i, refilled in refills if refilled]
            assert len(new_tables) <= 3 and 40 not in new_tables and 41 not in new_tables, new_tables
        del tr, arena, model
    assert digests["graph"] == digests["eager"], digests
