"""CPU: host-side logic of the plugin mirror — hyper-parameter tables, state-dict names, error behaviour, schedules.
No kernels are launched."""
import numpy as np
import pytest
import torch

from nerfstudio_amd import functional as F


def test_scalings_match_reference(golden):
    g = golden("kat")
    for name, (L, lo, hi, t) in {"main": (16, 16, 2048, 19), "prop0": (5, 16, 128, 17), "prop1": (5, 16, 256, 17)}.items():
        spec = F.HashGridSpec(L, lo, hi, t)
        np.testing.assert_array_equal(spec.scalings().numpy(), g[f"scalings_{name}"])
    assert F.HashGridSpec(16, 16, 2048, 19).scalings().tolist()[-1] == 2047.0  # fp32 pow quirk, SURVEY §8 a8
    assert F.HashGridSpec(16, 16, 2048, 19).native().num_levels == 16


def test_state_dict_names_match_reference_torch_path():
    """SURVEY.md §5 checkpoint contract + Appendix A shapes."""
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    m = NerfactoModel(NerfactoModelConfig(log2_hashmap_size=8, proposal_net_args_list=[
        {"hidden_dim": 16, "log2_hashmap_size": 6, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 6, "num_levels": 5, "max_res": 256, "use_linear": False}]),
        torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=100)
    sd = m.state_dict()
    expect = {
        "field.mlp_base.model.0.hash_table": (16 * 256, 2),
        "field.mlp_base.model.1.layers.0.weight": (64, 32),
        "field.mlp_base.model.1.layers.1.weight": (16, 64),
        "field.mlp_head.layers.0.weight": (64, 63),
        "field.mlp_head.layers.1.weight": (64, 64),
        "field.mlp_head.layers.2.weight": (3, 64),
        "field.embedding_appearance.embedding.weight": (100, 32),
        "proposal_networks.0.encoding.hash_table": (5 * 64, 2),
        "proposal_networks.0.mlp_base.0.hash_table": (5 * 64, 2),
        "proposal_networks.0.mlp_base.1.layers.0.weight": (16, 10),
        "proposal_networks.1.mlp_base.1.layers.1.weight": (1, 16),
    }
    for k, shape in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shape, (k, sd[k].shape)
    groups = m.get_param_groups()
    assert set(groups) == {"proposal_networks", "fields"}
    n_field = sum(p.numel() for p in groups["fields"])
    assert n_field == 16 * 256 * 2 + 64 * 32 + 64 + 16 * 64 + 16 + 64 * 63 + 64 + 64 * 64 + 64 + 3 * 64 + 3 + 100 * 32


def test_full_size_parameter_count():
    """field 16 792 019 (100 images), each proposal net 1 310 913."""
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    m = NerfactoModel(NerfactoModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=100)
    g = m.get_param_groups()
    assert sum(p.numel() for p in g["fields"]) == 16_792_019
    assert sum(p.numel() for p in g["proposal_networks"]) == 2 * 1_310_913


def test_no_silent_fallback():
    from nerfstudio_amd.field_components.encodings import HashEncoding, SHEncoding
    from nerfstudio_amd.field_components.mlp import MLP

    for impl in ("torch", "tcnn"):
        with pytest.raises(ValueError, match="implementation='hip' only"):
            HashEncoding(implementation=impl)
        with pytest.raises(ValueError, match="implementation='hip' only"):
            MLP(in_dim=4, num_layers=2, layer_width=8, implementation=impl)
    enc = HashEncoding(num_levels=2, log2_hashmap_size=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.rand(4, 3))  # CPU tensor: the product never computes on the host
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SHEncoding()(torch.rand(4, 3))
    with pytest.raises(ValueError):
        SHEncoding(levels=7)  # reference error behaviour (encodings.py:762-765)
    with pytest.raises(ValueError):
        F.HashGridSpec(16, 16, 2048, 19, features_per_level=4)


def test_encoding_api_contract():
    """tests/field_components/test_encodings.py:143-169 / test_mlp.py:11-28 shape contracts (host side)."""
    from nerfstudio_amd.field_components.encodings import HashEncoding
    from nerfstudio_amd.field_components.mlp import MLP, MLPWithHashEncoding

    enc = HashEncoding(num_levels=8, features_per_level=2, log2_hashmap_size=5, min_res=2, max_res=4)
    assert enc.get_out_dim() == 16 and enc.hash_table.shape == (8 * 32, 2)
    assert float(enc.hash_table.abs().max()) <= 1e-3
    mlp = MLP(in_dim=6, num_layers=2, layer_width=8, out_dim=10)
    assert mlp.get_out_dim() == 10 and [tuple(p.shape) for p in mlp.param_tensors()] == [(8, 6), (8,), (10, 8), (10,)]
    m = MLPWithHashEncoding(num_levels=4, log2_hashmap_size=4, layer_width=64, out_dim=16)
    assert list(dict(m.named_parameters())) == ["model.0.hash_table", "model.1.layers.0.weight", "model.1.layers.0.bias",
                                                 "model.1.layers.1.weight", "model.1.layers.1.bias"]


def test_ray_datastructures_shapes():
    from nerfstudio_amd.cameras.rays import Frustums, RayBundle, samples_from_bins

    n, s = 5, 7
    rb = RayBundle(origins=torch.zeros(n, 3), directions=torch.ones(n, 3), pixel_area=torch.ones(n, 1),
                   camera_indices=torch.arange(n)[:, None], nears=torch.zeros(n, 1), fars=torch.ones(n, 1))
    assert len(rb) == n and len(rb[:3]) == 3 and len(rb.get_row_major_sliced_ray_bundle(1, 4)) == 3
    t = torch.linspace(0, 1, s + 1)[None].expand(n, s + 1).contiguous()
    rs = samples_from_bins(rb, t, t * 2, None)
    assert rs.frustums.starts.shape == (n, s, 1) and rs.deltas.shape == (n, s, 1)
    # TensorDataclass semantics of the reference (utils/tensor_dataclass.py:67-92): every field is broadcast to the batch
    # shape — as zero-copy views (stride 0 along the sample axis), the dense per-ray arrays live in the pack
    assert rs.frustums.origins.shape == (n, s, 3) and rs.camera_indices.shape == (n, s, 1)
    assert rs.frustums.origins.stride(1) == 0 and rs.camera_indices.stride(1) == 0
    assert tuple(rs.frustums.shape) == (n, s) == tuple(rs.shape) and rs.pack.origins.shape == (n, 3)
    assert rs.frustums.get_positions().shape == (n, s, 3)
    # Frustums.get_positions KAT of the reference's tests/cameras/test_rays.py:11-30
    fr = Frustums(origins=torch.ones(5, 3), directions=torch.ones(5, 3) * torch.tensor([0.0, 1.0, 0.0]),
                  starts=torch.ones(5, 1) * 2, ends=torch.ones(5, 1) * 3, pixel_area=torch.ones(5, 1))
    assert torch.allclose(fr.get_positions()[0], torch.tensor([1.0, 3.5, 1.0]))


def test_proposal_schedule_and_anneal():
    """Control flow of ProposalNetworkSampler / NerfactoModel callbacks (ray_samplers.py:567-574,590;
    models/nerfacto.py:208-213, 270-280)."""
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    m = NerfactoModel(NerfactoModelConfig(log2_hashmap_size=6, proposal_net_args_list=[
        {"hidden_dim": 16, "log2_hashmap_size": 5, "num_levels": 5, "max_res": 128, "use_linear": False}] * 2),
        torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=3)
    ps = m.proposal_sampler
    assert ps.update_sched(0) == 1 and ps.update_sched(5000) == 5 and ps.update_sched(2500) == 2.5
    m.set_step(0)
    assert ps._anneal == 0.0
    m.set_step(1000)
    assert ps._anneal == pytest.approx(1.0)
    m.set_step(100)
    assert ps._anneal == pytest.approx(10 * 0.1 / (9 * 0.1 + 1))
    updated = []
    ps._step, ps._steps_since_update = 0, 0
    for step in range(1, 30):
        upd = ps._steps_since_update > ps.update_sched(ps._step) or ps._step < 10
        updated.append(upd)
        if upd:
            ps._steps_since_update = 0
        ps.step_cb(step)
    assert all(updated[:10]) and updated[10:16] == [False, True, False, True, False, True]


def test_lr_schedule_matches_reference_lambda_lr(golden):
    """ExponentialDecayScheduler vs the lr torch's LambdaLR sets with the reference's scheduler (fixture generated from
    /root/reference/nerfstudio/engine/schedulers.py by running optimizer.step(); scheduler.step() up to 250 000 times)."""
    import numpy as np

    from nerfstudio_amd.schedulers import ExponentialDecayScheduler as S, ExponentialDecaySchedulerConfig as C

    g = golden("schedulers")
    cfgs = {"nerfacto": C(lr_final=1e-4, max_steps=200000),
            "warm_cos": C(lr_final=1e-4, max_steps=5000, warmup_steps=100, lr_pre_warmup=1e-8),
            "warm_lin": C(lr_final=None, max_steps=5000, warmup_steps=100, ramp="linear")}
    for name, cfg in cfgs.items():
        got = np.array([S(cfg).get_lr(int(s), 1e-2) for s in g["steps"]])
        np.testing.assert_allclose(got, g[name], rtol=1e-12, err_msg=name)


def test_checkpoint_interchange_with_reference_layout():
    """Model tensors under `_model.` (with and without DDP's `module.`), Adam state in torch.optim.Adam's own layout —
    torch's optimiser must accept it — and a round trip through the arena."""
    import torch

    from nerfstudio_amd import checkpoint as C
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    def small():
        cfg = NerfactoModelConfig(log2_hashmap_size=8, num_levels=16, max_res=64,
                                  proposal_net_args_list=[{"hidden_dim": 16, "log2_hashmap_size": 7, "num_levels": 3, "max_res": 32,
                                                           "use_linear": False}] * 2)
        return NerfactoModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=5)

    torch.manual_seed(0)
    a = small()
    arena = ParamArena(a.get_param_groups_ordered())
    arena.exp_avg.uniform_(-1, 1)
    arena.exp_avg_sq.uniform_(0, 1)
    arena.step_counts.update({"fields": 12, "proposal_networks": 7})
    ckpt = C.make_checkpoint(a, arena, step=12)
    assert all(k.startswith("_model.") for k in ckpt["pipeline"])
    assert "_model.field.mlp_base.model.0.hash_table" in ckpt["pipeline"]
    # torch's own optimiser accepts the state (what Optimizers.load_optimizers does, engine/optimizers.py:195-203)
    for name, params in a.get_param_groups().items():
        opt = torch.optim.Adam(params, lr=1e-2, eps=1e-15)
        opt.load_state_dict(ckpt["optimizers"][name])
        st = opt.state[params[0]]
        assert float(st["step"]) == arena.step_counts[name] and st["exp_avg"].shape == params[0].shape
    # the reference trainer loads `scalers` into an ENABLED GradScaler unconditionally (trainer.py:137, :439; nerfacto
    # trains with mixed_precision=True): an empty dict raises there, so the checkpoint carries a valid scaler state
    scaler = torch.amp.GradScaler("cpu", enabled=True)
    scaler.load_state_dict(ckpt["scalers"])
    assert scaler.get_scale() == 65536.0
    resumed = {"scale": 1024.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 17}
    assert C.make_checkpoint(a, arena, step=13, scalers=resumed)["scalers"] == resumed  # passed through unchanged
    # into a second model + arena, through DDP-style keys and some foreign entries
    torch.manual_seed(1)
    b = small()
    arena_b = ParamArena(b.get_param_groups_ordered())
    foreign = {"module." + k: v for k, v in ckpt["pipeline"].items()}
    foreign["module.datamanager.train_camera_optimizer.pose_adjustment"] = torch.zeros(5, 6)
    assert C.load_model_state(b, foreign) == []
    C.load_optimizer_states(b, arena_b, ckpt["optimizers"])
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    for p_, off in zip(arena.params, arena.offsets):  # (the alignment padding between tensors is not part of a checkpoint)
        sl = slice(off, off + p_.numel())
        assert torch.equal(arena.exp_avg[sl], arena_b.exp_avg[sl]) and torch.equal(arena.exp_avg_sq[sl], arena_b.exp_avg_sq[sl])
    assert arena_b.step_counts == {"fields": 12, "proposal_networks": 7}
    assert b.field.mlp_base.encoding.hash_table.data_ptr() >= arena_b.flat.data_ptr()  # still views of the arena
    with pytest.raises(KeyError):
        C.load_model_state(small(), {"_model.field.mlp_head.layers.0.weight": torch.zeros(64, 63)})


def test_reachable_prefix_covers_every_corner_the_oracle_can_produce():
    """HashGridSpec.reachable_prefix (the rows a data-parallel exchange needs on the coarse levels) against the oracle's
    hash: every corner index of random and extreme positions lies in the list, and the list is exactly the lattice hash."""
    import numpy as np
    import torch

    from nerfstudio_amd import functional as F
    from oracle import nerfacto_oracle as orc

    spec = F.HashGridSpec(16, 16, 2048, 19)
    rows, idx = spec.reachable_prefix()
    T = spec.table_size
    assert rows == 5 * T and idx.numel() == 288066  # levels 16, 22, 30, 42, 58: (res+1)^3 < 2^19; 59^3 = 205 379 -> fewer after collisions
    scal = spec.scalings()
    assert torch.equal(scal, orc.hash_level_scalings(16, 16, 2048))
    allowed = set(idx.tolist())
    rs = np.random.RandomState(3)
    x = np.concatenate([rs.uniform(0, 1, (5000, 3)), np.array([[0, 0, 0], [1 - 2**-24] * 3, [0.5, 0, 1 - 2**-24]])]).astype(np.float32)
    for lvl in range(5):
        s_ = np.float32(scal[lvl])
        sc = x * s_
        lo, hi = np.floor(sc).astype(np.int64), np.ceil(sc).astype(np.int64)
        for cx in (lo[:, 0], hi[:, 0]):
            for cy in (lo[:, 1], hi[:, 1]):
                for cz in (lo[:, 2], hi[:, 2]):
                    got = orc.hash_corner_index(cx, cy, cz, lvl, T)
                    assert set(got.tolist()) <= allowed
        res = int(s_)
        c = np.arange(res + 1)
        gx, gy, gz = np.meshgrid(c, c, c, indexing="ij")
        lattice = np.unique(orc.hash_corner_index(gx.ravel(), gy.ravel(), gz.ravel(), lvl, T))
        mine = idx[(idx >= lvl * T) & (idx < (lvl + 1) * T)].numpy()
        assert np.array_equal(lattice, mine)


def test_committed_bench_line_follows_the_contract():
    """profiles/r02_final_bench_default.json is what `python bench.py` printed on the MI355X box: every field of the
    driver's contract is there with the right type, the roofline fraction is achieved / peak, value = rays / step time."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_final_bench_default.json")
    d = json.load(open(path))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)


def test_profiler_hooks_time_function_and_roctx_ranges(monkeypatch):
    """utils/profiler.py: the reference's two forms of `time_function` (decorator / context), its running-mean Profiler, and
    the roctx ranges behind the same hooks (push / pop pair up, nest, survive exceptions; straight-through when nothing
    listens)."""
    from nerfstudio_amd.utils import profiler as P

    calls = []

    class FakeRoctx:
        def roctxRangePushA(self, name):
            calls.append(("push", name.decode()))
            return 0

        def roctxRangePop(self):
            calls.append(("pop",))
            return 0

    monkeypatch.setattr(P, "_ROCTX", FakeRoctx())
    monkeypatch.setattr(P, "PROFILER", [])

    @P.time_function
    def work(x):
        with P.time_function("inner block"):
            return x * 2

    P.enable_ranges(False)
    assert work(3) == 6 and calls == []  # nobody listens: no range, no timing
    assert P.enable_ranges(True) and P.ranges_enabled()
    assert work(4) == 8
    assert calls == [("push", work.__qualname__), ("push", "inner block"), ("pop",), ("pop",)]
    calls.clear()

    @P.time_function
    def boom():
        raise KeyError("x")

    with pytest.raises(KeyError):
        boom()
    assert calls == [("push", boom.__qualname__), ("pop",)]
    prof = P.setup_profiler()
    for _ in range(3):
        work(1)
    assert prof.profiler_dict[work.__qualname__]["step"] == 3 and prof.profiler_dict["inner block"]["step"] == 3
    P.enable_ranges(False)
    # the phases of the explicit kernel schedule carry the hook
    from nerfstudio_amd.train_step import NerfactoTrainStep

    for name in ("forward_proposals", "forward_main", "losses", "backward_main", "backward_proposals"):
        assert hasattr(getattr(NerfactoTrainStep, name), "__wrapped__"), name


def test_ngp_engine_runs_the_collider_and_falls_back_when_the_schedule_refuses_the_model():
    """pipeline.NgpEngine: (1) Model.forward applies the collider before get_outputs
    (models/base_model.py:140-141) — the engine must hand the COLLIDED bundle to the schedule, or enable_collider configs march
    the config's near / far planes instead of the collider's; (2) a model shape NgpTrainStep refuses must select the module
    path (a `reason`), not crash the first training iteration."""
    import types

    import torch

    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.pipeline import NgpEngine

    seen = {}

    class _Runner:
        num_kept = 7
        target = torch.zeros(4, 3)

        def outputs(self):
            return {"rgb": torch.zeros(4, 3)}

    class _Trainer:
        runner = _Runner()

        def set_batch(self, rb, batch):
            seen["nears"], seen["fars"] = rb.nears, rb.fars

        def train_iteration(self, step):
            return torch.zeros(())

    def collider(rb):
        rb.nears = torch.full_like(rb.origins[..., :1], 0.25)
        rb.fars = torch.full_like(rb.origins[..., :1], 3.5)
        return rb

    eng = object.__new__(NgpEngine)
    eng.trainer, eng.arena = _Trainer(), types.SimpleNamespace(lr=0.0)
    eng.optimizers = types.SimpleNamespace(optimizers={"fields": types.SimpleNamespace(param_groups=[{"lr": 1e-2}])})
    eng.pipeline = types.SimpleNamespace(model=types.SimpleNamespace(collider=collider))
    eng._bind_grads, eng._grad_views, eng._anchor = False, [], torch.zeros((), requires_grad=True)
    rb = RayBundle(origins=torch.zeros(4, 3), directions=torch.ones(4, 3), pixel_area=torch.ones(4, 1),
                   camera_indices=torch.zeros(4, 1, dtype=torch.long))
    eng.train_iteration(0, rb, {"image": torch.zeros(4, 3)})
    assert float(seen["nears"][0]) == 0.25 and float(seen["fars"][0]) == 3.5 and eng.arena.lr == 1e-2

    # (2) the schedule's constructor refuses: build() answers with a reason and leaves no trainer behind
    eng2 = object.__new__(NgpEngine)
    eng2.reason, eng2.trainer, eng2.runner_factory, eng2.on_build = None, None, None, None
    eng2._optimizer_reason = lambda: None
    eng2._adopt_optimizer_state = lambda: None
    p = torch.nn.Parameter(torch.zeros(8))
    eng2.optimizers = types.SimpleNamespace(
        parameters={"fields": [p]},
        optimizers={"fields": types.SimpleNamespace(param_groups=[{"lr": 1e-2, "betas": (0.9, 0.999), "eps": 1e-15}])})
    eng2.pipeline = types.SimpleNamespace(model=types.SimpleNamespace(config=types.SimpleNamespace(use_gradient_scaling=False)))

    def refuse(ray_bundle, batch):
        raise NotImplementedError("hash grid out_dim != 32")

    eng2.build_trainer_only = refuse
    eng2.runner_factory = object()  # (rays on the CPU are fine with a stand-in runner factory)
    reason = eng2.build(rb, {"image": torch.zeros(4, 3)})
    assert reason is not None and "out_dim" in reason and eng2.reason == reason and eng2.trainer is None


def test_pinhole_camera_args_picks_only_cameras_the_grid_generator_covers():
    """eval_render.pinhole_camera_args (round 6): Model.get_outputs_for_camera generates rays inside the chunk loop only for ONE
    undistorted perspective camera; everything else (other lens types, distortion, several cameras, missing fields) must take the
    reference's generate_rays route (cameras/cameras.py:321-503)."""
    import types

    import torch

    from nerfstudio_amd.eval_render import pinhole_camera_args

    def cam(**kw):
        base = dict(camera_to_worlds=torch.eye(4)[None, :3], fx=torch.tensor([[100.0]]), fy=torch.tensor([[90.0]]),
                    cx=torch.tensor([[32.0]]), cy=torch.tensor([[24.0]]), height=torch.tensor([[48]]), width=torch.tensor([[64]]),
                    camera_type=torch.tensor([[1]]), distortion_params=None)
        base.update(kw)
        return types.SimpleNamespace(**base)

    c2w, fx, fy, cx, cy, h, w = pinhole_camera_args(cam())
    assert c2w.shape == (3, 4) and (fx, fy, cx, cy, h, w) == (100.0, 90.0, 32.0, 24.0, 48, 64)
    assert pinhole_camera_args(cam(camera_to_worlds=torch.eye(4)[:3])) is not None          # an unbatched [3, 4] pose
    assert pinhole_camera_args(cam(distortion_params=torch.zeros(1, 6))) is not None          # all-zero distortion = none
    assert pinhole_camera_args(cam(camera_type=torch.tensor([[2]]))) is None                  # fisheye
    assert pinhole_camera_args(cam(distortion_params=torch.tensor([[0.1, 0, 0, 0, 0, 0]]))) is None
    assert pinhole_camera_args(cam(camera_to_worlds=torch.eye(4)[None, :3].repeat(2, 1, 1))) is None  # two cameras
    assert pinhole_camera_args(cam(fx=torch.tensor([[100.0], [101.0]]))) is None
    assert pinhole_camera_args(types.SimpleNamespace(camera_to_worlds=torch.eye(4)[None, :3])) is None  # no intrinsics
