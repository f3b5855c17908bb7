#!/usr/bin/env python3
"""bench.py — training rays/sec of the nerfacto hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full nerfacto training iteration on a batch of 4096 synthetic rays PER GPU (BASELINE configs[1]:
L=16 hash T=2^19 F=2, 64-wide MLPs, proposal sampler 256 -> 96 -> 48 samples/ray): proposal sampling, main field,
compositing, MSE + interlevel + distortion losses, backward of all of it, gradient exchange (N > 1), Adam over all
19.4 M parameters, and the reference's per-step callbacks (proposal update schedule, weight anneal). The iteration is the
PRODUCT's: nerfstudio_amd.trainer.HipTrainer, the object the `nerfacto-hip` method's pipeline drives under the reference's
own trainer (nerfstudio_amd/pipeline.py); this script owns only the workload (synthetic rays resident in HBM before the
timed region), the clock and the JSON line. value = world_size * rays_per_batch / time_per_step, the reference's own rays/s
definition (engine/trainer.py:276-284) with a device sync around the timed region.

The timed region — K steps after W warm-up steps, barrier + device sync on both sides, max over ranks — is run `--windows`
times (default 7) ON THE SAME K ITERATIONS: the training state at the start of the window (parameters, Adam moments, step
counters, sampler state) is restored before every repeat, so every repeat measures the same work in the same phase of the
proposal-update schedule; `ms_per_step` is the MEDIAN repeat, config.window_ms holds min / max / n. A single 15 ms window
moved by +-5 % from run to run. `long_run` continues from there for `--long-steps` iterations (default 300).

Extra objects on the JSON line:
  roofline      — the kernel with the largest share of the step (runner-up riding along), timed live with HIP events on the
                  launch stream (separate, untimed profiling steps after the timed region), against its algorithmic work.
  roofline_step — SURVEY.md §8(d) whole-step algorithmic HBM bytes / ms_per_step against the 8 TB/s peak.
  cpu_baseline  — the CPU oracle (oracle/nerfacto_oracle.py, kind "port") running the same training step on one full
                  batch on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RAYS_PER_GPU = 4096
BATCH_SLOTS = 8  # pre-generated ray batches resident in HBM; the timed loop takes a different one every step
NGP_DENSITY = 60.0  # --workload ngp (scripts/bench_ngp.py): the synthetic field's density


def build_model(device, seed, camera_optimizer="off"):
    from nerfstudio_amd.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    torch.manual_seed(seed)
    cfg = NerfactoModelConfig(camera_optimizer=CameraOptimizerConfig(mode=camera_optimizer))
    model = NerfactoModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=100)
    return model.to(device).train()


def synthetic_rays(seed, workload="bounded"):
    """One batch of RAYS_PER_GPU synthetic rays (numpy): origins, unit directions, camera ids U{0..99}, targets U(0,1).
    "bounded"   — BASELINE.md §2: origins ~ N(0, 0.5^2) inside the [-1,1]^3 box, isotropic directions (configs[1]/[2]).
    "unbounded" — configs[4] (mipnerf-360-style capture): cameras on a shell of radius ~3 around the box
                  looking inward with a wide field of view, so most samples fall in the contracted region ||x||_inf > 1
                  (far plane 1000, L-inf scene contraction); same model and sampler (256 -> 96 -> 48)."""
    rs = np.random.RandomState(seed)
    n = RAYS_PER_GPU
    if workload == "unbounded":
        u = rs.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=-1, keepdims=True)
        o = (3.0 * u + 0.1 * rs.standard_normal((n, 3))).astype(np.float32)
        d = (-u + 0.45 * rs.standard_normal((n, 3))).astype(np.float32)
    else:
        o = (rs.standard_normal((n, 3)) * 0.5).astype(np.float32)
        d = rs.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cam = rs.randint(0, 100, size=(n, 1)).astype(np.int64)
    tgt = rs.uniform(0, 1, size=(n, 3)).astype(np.float32)
    return o, d.astype(np.float32), cam, tgt


def synthetic_batch(device, seed, workload="bounded"):
    """-> (RayBundle of slot 0, {"image": targets of slot 0}, pool): the pool holds BATCH_SLOTS batches in HBM as
    [slots, N, 3] / [slots, N] tensors (seeds seed, seed + 1, ...)."""
    from nerfstudio_amd.cameras.rays import RayBundle

    n = RAYS_PER_GPU
    parts = [synthetic_rays(seed + k, workload) for k in range(BATCH_SLOTS)]
    pool = {"origins": torch.from_numpy(np.stack([p[0] for p in parts])).to(device),
            "directions": torch.from_numpy(np.stack([p[1] for p in parts])).to(device),
            "cameras": torch.from_numpy(np.stack([p[2][:, 0] for p in parts])).to(device),
            "target": torch.from_numpy(np.stack([p[3] for p in parts])).to(device)}
    rb = RayBundle(origins=pool["origins"][0].clone(), directions=pool["directions"][0].clone(),
                   pixel_area=torch.full((n, 1), 1e-6, device=device), camera_indices=pool["cameras"][0].clone()[:, None])
    return rb, {"image": pool["target"][0].clone()}, pool


def Trainer(model, arena, ray_bundle, batch, **kw):
    """The product's training iteration (nerfstudio_amd/trainer.py); kept under this name for the probe scripts."""
    from nerfstudio_amd.trainer import HipTrainer

    return HipTrainer(model, arena, ray_bundle, batch, **kw)


class TrainingState:
    """Everything a training iteration changes, so that a timed window can be repeated on the same iterations."""

    def __init__(self, trainer, arena, model):
        trainer.finish()
        ps = model.proposal_sampler
        self.t, self.a, self.m = trainer, arena, model
        self.tensors = [x.clone() for x in (arena.flat, arena.exp_avg, arena.exp_avg_sq)]
        self.counts = dict(arena.step_counts)
        self.scalars = (trainer.step, trainer.opt_step, ps._step, ps._steps_since_update, ps._anneal, model.step)

    def restore(self):
        t, a, m = self.t, self.a, self.m
        t.finish()
        for dst, src in zip((a.flat, a.exp_avg, a.exp_avg_sq), self.tensors):
            dst.copy_(src)
        a.step_counts.update(self.counts)
        t._true_steps = dict(self.counts)
        ps = m.proposal_sampler
        t.step, t.opt_step, ps._step, ps._steps_since_update, ps._anneal, m.step = self.scalars


def cpu_baseline(n_rays=RAYS_PER_GPU, steps=3, threads=None, workload="bounded"):
    """The CPU oracle (oracle/nerfacto_oracle.py, a restatement of the reference's torch path pinned to it by
    tests/golden) running the same training step — forward, losses, backward, Adam over all 19.4 M parameters — on the
    METRIC'S configuration: one batch of 4096 rays (BASELINE configs[1]); 1 warm-up step + `steps` timed ones, median
    (about 2-4 s per step). profiles/r02_cpu_reference_vs_port.txt holds the authoring-container comparison of this port
    with the reference's own modules on the same 4096 rays (port / reference = 0.97).
    Thread count: torch's CPU ops on this workload peak at ~16 threads on the MI355X host (measured 8/16/32/64/128
    threads: 135/141/111/63/30 rays/s on the 256-ray sample of round 1), so 16 is used rather than all cores."""
    from oracle import nerfacto_oracle as orc

    torch.set_num_threads(threads if threads is not None else min(16, os.cpu_count() or 16))
    cfg = orc.NerfactoCfg()
    params = orc.init_params(cfg, seed=0)
    plist = list(params.values())
    for p in plist:
        p.requires_grad_(True)
    opt = torch.optim.Adam(plist, lr=1e-2, eps=1e-15)
    o, d, cam, tgt = (torch.from_numpy(a) for a in synthetic_rays(1000, workload))
    o, d, cam, tgt = o[:n_rays], d[:n_rays], cam[:n_rays, 0], tgt[:n_rays]
    rs = np.random.RandomState(1)
    times = []
    for it in range(steps + 1):
        jit = [torch.from_numpy(rs.uniform(0, 1, (n_rays, 1)).astype(np.float32)) for _ in range(3)]
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = orc.nerfacto_forward(params, cfg, o, d, cam, jit, training=True)
        sum(orc.nerfacto_losses(out, tgt, cfg).values()).backward()
        opt.step()
        if it > 0:  # first step pays allocator / thread-pool warm-up
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(n_rays / med, 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_rays} rays x (256,96,48) samples = one full batch of the metric's configuration, full nerfacto "
                      f"tables, fwd+losses+bwd+Adam, median of {steps} steps after 1 warm-up ({med:.2f} s/step)"}


def cpu_baseline_ngp(o, d, cam, tgt, binaries, cfg, n_rays=256, steps=2):
    """The packed path's CPU restatement (oracle/packed_oracle.py marcher + visibility, oracle field, packed compositing,
    MSE, backward, torch Adam) on a bounded sample: `n_rays` of the same rays against the same grid (the numpy marcher walks
    rays step by step: ~10 s per 256 rays)."""
    from oracle import nerfacto_oracle as orc
    from oracle import packed_oracle as po

    torch.set_num_threads(min(16, os.cpu_count() or 16))
    ocfg = orc.NerfactoCfg(prop_grids=(), num_images=100, average_init_density=1.0)
    params = {k: v for k, v in orc.init_params(ocfg, seed=0).items() if k.startswith("field.")}
    with torch.no_grad():
        params["field.mlp_base.model.1.layers.1.bias"][0] = float(np.log(NGP_DENSITY))
    for p in params.values():
        p.requires_grad_(True)
    opt = torch.optim.Adam(list(params.values()), lr=1e-2, eps=1e-15)
    o, d, cam, tgt = o[:n_rays], d[:n_rays], cam[:n_rays, 0], tgt[:n_rays]
    to, td, tcam, ttgt = torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(cam), torch.from_numpy(tgt)
    rs = np.random.RandomState(3)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        jit = rs.uniform(0, 1, n_rays).astype(np.float32)
        idx, ts, te = po.occgrid_march(o, d, binaries, [-1, -1, -1, 1, 1, 1], cfg.render_step_size, near_plane=cfg.near_plane,
                                       far_plane=cfg.far_plane, cone_angle=cfg.cone_angle, jitter=jit)
        idx, ts, te = torch.from_numpy(idx), torch.from_numpy(ts), torch.from_numpy(te)
        pos = to[idx] + td[idx] * ((ts + te) / 2)[:, None]
        with torch.no_grad():
            sig = orc.nerfacto_field(pos, td[idx], tcam[idx], params, ocfg, training=True)[0]
            keep = po.render_visibility_from_density(ts, te, sig, idx, n_rays, 1e-4, min(cfg.alpha_thre, float(binaries.mean())))
        idx, ts, te = idx[keep], ts[keep], te[keep]
        pos = to[idx] + td[idx] * ((ts + te) / 2)[:, None]
        opt.zero_grad(set_to_none=True)
        dens, rgb_s, _ = orc.nerfacto_field(pos, td[idx], tcam[idx], params, ocfg, training=True)
        w = po.render_weight_from_density(ts, te, dens, idx, n_rays)[0]
        comp, acc, _ = po.composite_packed(rgb_s, w, ts, te, idx, n_rays, background="random", training=True)
        bg = torch.rand_like(comp)
        loss = torch.mean((ttgt - (comp + bg * (1.0 - acc))) ** 2)
        loss.backward()
        opt.step()
        if it > 0:
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(n_rays / med, 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_rays} of the step's 4096 rays against the same occupancy grid: numpy marcher + visibility, oracle field, "
                      f"packed compositing, MSE, backward, Adam over the field's parameters (no grid refresh); median of {steps} "
                      f"steps after 1 warm-up ({med:.2f} s/step)"}


def secondary_lines(steps, warmup, headline_ms):
    """The lines of the other workloads this repository covers, timed in the DEFAULT run so that the driver's one JSON line
    carries them: one window of `steps` iterations each, measured by this same script (or scripts/bench_seam.py)
    in a process of its own on the same GPU right after the headline — the reference's nerfacto default with the camera
    optimiser on (models/nerfacto.py:131 SO3xR3), BASELINE configs[4] (unbounded / contraction), configs[3] (instant-ngp incl.
    its occupancy refresh), and the iteration through the reference's Trainer / Pipeline seam against the direct line.
    Never the headline; a failed sub-run is reported as such instead of failing the bench."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    common = ["--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-secondary", "--long-steps", "0",
              "--windows", "3"]
    runs = {"camera_optimizer_SO3xR3": [os.path.join(here, "bench.py"), "--camera-optimizer", "SO3xR3"] + common,
            "unbounded_configs4": [os.path.join(here, "bench.py"), "--workload", "unbounded"] + common,
            "instant_ngp_configs3": [os.path.join(here, "bench.py"), "--workload", "ngp"] + common,
            "trainer_pipeline_seam": [os.path.join(here, "scripts", "bench_seam.py"), "--steps", str(steps), "--warmup", str(warmup),
                                      "--windows", "3"]}
    out = {}
    for name, cmd in runs.items():
        t0 = time.perf_counter()
        try:
            res = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=240, env=env)
            line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith("{")), None)
            if res.returncode != 0 or line is None:
                out[name] = {"error": f"rc {res.returncode}: {(res.stderr or res.stdout)[-200:]}"}
                continue
            j = json.loads(line)
        except Exception as e:  # noqa: BLE001 - a secondary line must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            continue
        wall = round(time.perf_counter() - t0, 1)
        if name == "trainer_pipeline_seam":
            out[name] = {"seam_ms_per_step": j["seam_ms"], "direct_ms_per_step": j["direct_pool_ms"],
                         "seam_over_direct": j["seam_over_direct_pool"], "windows": j["windows"], "process_s": wall}
        else:
            out[name] = {"ms_per_step": j["ms_per_step"], "value": j["value"], "unit": j["unit"], "steps": j["steps"],
                         "over_headline": round(float(j["ms_per_step"]) / headline_ms, 4), "workload": j["config"].get("workload"),
                         "process_s": wall}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=7, help="repeats of the timed K-step window on the same iterations (median)")
    ap.add_argument("--long-steps", type=int, default=300, help="secondary figure: this many further iterations (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--autograd", action="store_true",
                    help="drive the step through the nn.Module / autograd API instead of the explicit kernel schedule")
    ap.add_argument("--fused-model-api", action="store_true",
                    help="with the nn.Module driver (implies --autograd): config.fused_train_step")
    ap.add_argument("--camera-optimizer", choices=["off", "SO3xR3", "SE3"], default="off",
                    help="pose refinement of the training cameras (models/nerfacto.py:131; the reference's nerfacto default is "
                         "SO3xR3, its Blender benchmark recipe and this headline run it off)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL on ROCm); gloo only for functional tests")
    ap.add_argument("--share-gpu", action="store_true", help="functional test: every rank uses cuda:0 (needs gloo)")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--dp-graph", action="store_true", help="N > 1: replay captured hipGraph segments instead of eager launches")
    ap.add_argument("--ngp-module-path", action="store_true", help="--workload ngp through the nn.Module / autograd classes")
    ap.add_argument("--ngp-no-refresh", action="store_true", help="--workload ngp without the occupancy-grid refresh (kernel schedule only)")
    ap.add_argument("--workload", choices=["bounded", "unbounded", "ngp"], default="bounded",
                    help="bounded = BASELINE configs[1]/[2] (the metric's configuration); unbounded = configs[4]; ngp = configs[3]")
    ap.add_argument("--start-step", type=int, default=0,
                    help="not the headline: start the step counter (proposal update schedule, anneal, learning rate) here")
    ap.add_argument("--fixed-batch", action="store_true", help="train on one fixed ray batch instead of rotating the pool")
    ap.add_argument("--force-dp", action="store_true",
                    help="N = 1 only: the data-parallel schedule over a ONE-rank RCCL communicator (rehearsal of the N > 1 path)")
    ap.add_argument("--dp-mode", choices=["allreduce", "sharded"], default="allreduce")
    ap.add_argument("--param-checksum", action="store_true", help="add sha256 digests of the arenas after the FIRST window to config")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: rehearse the multi-GPU launch plumbing over gloo")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` object of the default N = 1 run (the other workloads' 20-step windows, see secondary_lines)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if args.dry_run:
        from nerfstudio_amd.dp_schedule import rehearse_on_cpu

        return rehearse_on_cpu(build_model, args.steps, args.warmup, args.dp_mode, rank, world)
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.force_dp:
        assert world == 1, "--force-dp is the single-GPU rehearsal of the data-parallel path"
        os.environ["NSAMD_FORCE_COLLECTIVES"] = "1"
        for k, v in (("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=args.dist_backend)

    from nerfstudio_amd import _native
    from nerfstudio_amd import functional as F
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.trainer import HipTrainer
    from nerfstudio_amd.utils import roofline as RL

    _native.load()  # fail loudly if the HIP extension is missing
    F.DIRECT_GRAD = True  # backward kernels accumulate straight into the arena's gradient views
    if args.workload == "ngp":
        assert world == 1 and not args.force_dp, "--workload ngp is a single-GPU line"
        from scripts.bench_ngp import run_ngp  # the instant-ngp line: its workload and JSON live in scripts/bench_ngp.py

        return run_ngp(args, device, synthetic_rays, cpu_baseline_ngp)
    model = build_model(device, seed=0, camera_optimizer=args.camera_optimizer)  # same init on every rank (replicated model)
    if args.fused_model_api:
        args.autograd = True
        model.config.fused_train_step = True
    # the reference's optimiser groups (models/nerfacto.py:255-260), AdamOptimizerConfig(lr=1e-2, eps=1e-15) each
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    arena.broadcast_params()
    same = os.environ.get("NSAMD_BENCH_SAME_RAYS") == "1"  # functional check: N ranks, identical rays == the N=1 run
    # each rank its own rays (scripts/train.py:98): BATCH_SLOTS batches per rank, disjoint seeds
    rb, batch, pool = synthetic_batch(device, seed=1000 + (0 if same else 100 * rank), workload=args.workload)
    trainer = HipTrainer(model, arena, rb, batch, world=world, use_graph=not args.no_graph, use_runner=not args.autograd,
                         pool=None if args.fixed_batch else pool, force_dp=args.force_dp, dp_mode=args.dp_mode)
    if args.start_step:
        trainer.step = args.start_step
        model.proposal_sampler._step = args.start_step
    for _ in range(max(1, args.warmup // 2)):  # eager warm-up: lazy kernel attributes, caches, allocator
        trainer.train_iteration()
    trainer.finish()
    # N > 1 launches the segments eagerly by default (the host keeps ahead of the GPU at this kernel granularity, and
    # captured segments between eager collectives could only be exercised over gloo on a shared GPU); --dp-graph opts in
    graphed = trainer.try_capture() if ((world == 1 and not args.force_dp) or args.dp_graph) else False
    for _ in range(args.warmup - max(1, args.warmup // 2)):
        trainer.train_iteration()
    trainer.finish()

    def timed_window(steps):
        """K iterations between barrier + device sync on both sides; -> (seconds, max over ranks; proposal-update steps)."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        updates = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            updates += int(model.proposal_sampler.updated_this_step())
            trainer.train_iteration()
        trainer.finish()  # the last step's pending main-field update is part of the timed work
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        sec = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([sec], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t.item())
        return sec, updates

    state = TrainingState(trainer, arena, model) if args.windows > 1 else None
    windows, checksum, updates = [], None, 0
    for w in range(max(1, args.windows)):
        if w > 0:
            state.restore()
        sec, updates = timed_window(args.steps)
        windows.append(sec)
        if w == 0:
            loss = trainer.last_loss()
            assert bool(torch.isfinite(loss)), "training diverged"
            if args.param_checksum and rank == 0:  # state right after the first window
                import hashlib

                checksum = {name: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:32]
                            for name, t in (("params", arena.flat), ("exp_avg", arena.exp_avg), ("exp_avg_sq", arena.exp_avg_sq))}
    elapsed = float(np.median(windows))
    long_run = None
    if args.long_steps > 0:
        sec, upd = timed_window(args.long_steps)
        long_run = {"steps": args.long_steps, "ms_per_step": round(sec / args.long_steps * 1e3, 4),
                    "value": round(world * RAYS_PER_GPU / (sec / args.long_steps), 1), "proposal_update_steps": upd,
                    "from_step": trainer.step - args.long_steps}
    if getattr(trainer, "_seg_times", None) and rank == 0:
        for k, v in trainer._seg_times.items():
            print(f"[dp-timing] {k:16s} n={len(v):4d} median {sorted(v)[len(v) // 2]:9.3f} ms  max {max(v):9.3f} ms", file=sys.stderr)

    main_points = RAYS_PER_GPU * model.config.num_nerf_samples_per_ray
    roof, table = (None, [])
    if rank == 0:
        if state is not None:
            state.restore()  # the per-kernel table is taken on the window's own iterations
        roof, table = RL.measure_roofline(trainer, arena, max(1, args.profile_steps), main_points)
    elif world > 1:  # keep the collective pattern identical on every rank during the profiling steps
        if state is not None:
            state.restore()
        for _ in range(max(1, args.profile_steps) + RL.ROOFLINE_WARMUP_ITERS):  # (+ measure_roofline's untimed warm-up on rank 0)
            trainer.train_iteration()
        trainer.finish()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        step_bytes = RL.step_algorithmic_bytes(RAYS_PER_GPU, params=arena.numel, updated_fraction=updates / args.steps)
        out = {
            "metric": "training rays/sec (4096 rays x 48 samples per GPU)",
            "value": round(world * RAYS_PER_GPU / (elapsed / args.steps), 1),
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("nerfacto 1xMI355X" if world == 1 else f"nerfacto {world}xMI355X data-parallel") +
                                   ": L=16 hash (T=2^19, F=2), 64x2 MLP, 48 samples/ray, 4096 rays/batch per GPU " +
                                   ("(BASELINE configs[1]/[2])" if args.workload == "bounded" else
                                    "(BASELINE configs[4]: unbounded scene, cameras at radius ~3, L-inf contraction)") +
                                   "; full training step incl. proposal nets 256->96, losses, Adam; "
                                   f"{1 if args.fixed_batch else BATCH_SLOTS} ray batches resident in HBM, rotated per step",
                       "rays": args.workload, "camera_optimizer": args.camera_optimizer,
                       "rays_per_gpu": RAYS_PER_GPU, "global_rays": world * RAYS_PER_GPU,
                       "window_ms": {"median": round(ms, 4), "min": round(min(windows) / args.steps * 1e3, 4),
                                     "max": round(max(windows) / args.steps * 1e3, 4), "n": len(windows),
                                     "proposal_update_steps": updates,
                                     "note": "the same K iterations repeated from the restored training state"},
                       "parallelism": f"dp{world}: rays sharded by batch; RCCL all-reduce of the gradient arena slices "
                                      "(main field 48 MB async — the coarse table levels go as their 288 k reachable rows — "
                                      "pipelined under the proposal backward and the next proposal forward; proposal slice "
                                      "only on update steps)",
                       "params": arena.numel, "final_loss": round(float(loss), 6),
                       "launch": (("hipGraph replay (4 captured variants: proposal update x pending main-field "
                                   + ("table scatter + Adam, which run" if trainer.defer_scatter else "Adam, which runs")
                                   + " beside the next proposal forward)" if trainer.defer else
                                   "hipGraph replay (2 captured variants)") if not trainer.pipelined else
                                  "hipGraph replay (6 captured segments)") if graphed else "eager",
                       "driver": ("Model API over the explicit kernel schedule (fused_step.py)" if args.fused_model_api else
                                  "autograd modules") if args.autograd else
                                 "nerfstudio_amd.trainer.HipTrainer over the explicit kernel schedule (train_step.py)"},
            "roofline": roof,
            "roofline_step": {"bound": "hbm", "achieved": round(step_bytes / (ms * 1e-3) / 1e9, 1), "peak": RL.HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(step_bytes / (ms * 1e-3) / 1e9 / RL.HBM_PEAK_GBS, 4),
                              "algorithmic_bytes_per_step": int(step_bytes),
                              "note": "SURVEY.md 8(d): hash gathers of all levels + main-table scatter + Adam (28 B/param) + the "
                                      "proposal tables' scatter on the window's update steps"},
        }
        if long_run is not None:
            out["long_run"] = long_run
        if args.start_step:
            out["config"]["start_step"] = args.start_step
        if checksum is not None:
            out["config"]["param_checksum"] = checksum
        if world > 1 or args.force_dp:  # what the process group itself reports (not the --gpus argument)
            out["config"].update(dp_mode=args.dp_mode, rccl_ranks=dist.get_world_size(), dist_backend=dist.get_backend())
        if args.force_dp:
            out["config"]["force_dp"] = "data-parallel schedule over a one-rank RCCL communicator (rehearsal of the N > 1 path)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(workload=args.workload)
        if (world == 1 and not args.no_secondary and not args.force_dp and args.workload == "bounded" and args.camera_optimizer == "off"
                and not (args.autograd or args.no_graph or args.start_step or args.kernel_table or args.param_checksum)):
            out["secondary"] = secondary_lines(args.steps, args.warmup, float(out["ms_per_step"]))
        if args.kernel_table:
            for r in table:
                print(f"{r['kernel']:64s} {r['calls_per_step']:5.1f}/step {r['ms_per_step']:9.4f} ms/step "
                      f"{r['mean_ms']:9.4f} ms/launch", file=sys.stderr)
        print(json.dumps(out))
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
