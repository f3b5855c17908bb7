"""`bench.py --workload ngp`: BASELINE configs[3] — instant-ngp (occupancy-grid ray marching + early termination, packed
samples) on 4096 synthetic rays, through nerfstudio_amd.ngp_trainer.NgpTrainer. The workload, the clock and the JSON line;
the CPU baseline leg (the only user of oracle/) is handed in by bench.py."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RAYS_PER_GPU = 4096
NGP_DENSITY = 60.0  # sigma ~ 60 -> alpha ~ 0.19 per step, a ray is opaque (T < 1e-4) after ~45 samples


def ngp_lattice_steps(o, d, step, cone, near, far, levels):
    """Lattice steps every ray walks through the outermost grid level (numpy, fp32, the marcher's own recurrence without the
    cell lookups): the algorithmic work of the occupancy march — one occupancy byte per step."""
    f = np.float32
    half = f(1 << (levels - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f(1.0) / d).astype(f)
        ta, tb = ((-half - o) * inv).astype(f), ((half - o) * inv).astype(f)
    lo, hi = np.minimum(ta, tb), np.maximum(ta, tb)
    t0 = np.maximum(np.nanmax(lo, axis=1), f(near)).astype(f)
    t1 = np.minimum(np.nanmin(hi, axis=1), f(far)).astype(f)
    t, n = t0.copy(), np.zeros(len(o), np.int64)
    alive = t < t1
    while alive.any():
        n += alive
        dt = np.minimum(np.maximum((t * f(cone)).astype(f), f(step)), f(1e10)).astype(f)
        t = np.where(alive, (t + dt).astype(f), t)
        alive &= t < t1
    return n


def build_ngp_model(device):
    """NGPModel + the synthetic occupancy state -> (model, keep_synthetic_grid).
    The grid is SYNTHETIC (random 5 %-occupied 128^3 x 4 levels) and the random field's density is lifted to
    ~60 so that rays become opaque after ~45 kept samples; `keep_synthetic_grid()` puts the synthetic `occs` back (after a
    refresh replaced them by the random field's own occupancy: every cell occupied)."""
    from nerfstudio_amd.instant_ngp import InstantNGPModelConfig, NGPModel

    torch.manual_seed(0)
    cfg = InstantNGPModelConfig()  # grid 128^3 x 4 levels, T = 2^19, cone_angle 0.004, alpha_thre 0.01, random background
    model = NGPModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=100).to(device).train()
    with torch.no_grad():  # lift the density head: sigma = exp(pre), pre ~ log(NGP_DENSITY)
        model.field.mlp_base.mlp.layers[-1].bias[0] = float(np.log(NGP_DENSITY))
    grid = model.occupancy_grid
    g = torch.Generator(device="cpu").manual_seed(7)
    occupied = torch.rand(grid.occs.shape, generator=g) < 0.05
    grid.occs.copy_(torch.where(occupied, torch.tensor(1.0), torch.tensor(0.0)).to(device))
    grid._refresh_derived(0.01)
    assert abs(float(grid.binaries.float().mean()) - 0.05) < 5e-3
    occs0 = grid.occs.clone()

    def keep_synthetic_grid():
        grid.occs.copy_(occs0)
        grid._refresh_derived(0.01)

    return model, keep_synthetic_grid


def build_ngp(device, synthetic_rays, module_path=False, refresh=True):
    """`build_ngp_model` + nerfstudio_amd.ngp_trainer.NgpTrainer on one batch of synthetic rays.
    The refresh of the grid runs inside the iteration exactly as in training (every 16th step: cells_per_lvl / 4 uniform + the
    occupied cells of each level, density of 2.5 M cell points, decayed maximum, threshold, coarse bitfield); the bench puts
    the synthetic `occs` back after each refresh — two more device copies INSIDE the timed region, no work skipped — so that
    every step marches the same 5 % grid."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
    from nerfstudio_amd.ngp_trainer import NgpTrainer

    model, keep_synthetic_grid = build_ngp_model(device)
    arena = ParamArena({"fields": list(model.field.parameters())}, lr=1e-2, eps=1e-15)
    o, d, cam, tgt = synthetic_rays(1000)
    n = RAYS_PER_GPU
    rb = RayBundle(origins=torch.from_numpy(o).to(device), directions=torch.from_numpy(d).to(device),
                   pixel_area=torch.full((n, 1), 1e-6, device=device), camera_indices=torch.from_numpy(cam).to(device))
    trainer = NgpTrainer(model, arena, n, device, module_path=module_path, after_refresh=keep_synthetic_grid, refresh=refresh)
    trainer.set_batch(rb, {"image": torch.from_numpy(tgt).to(device)})
    return model, arena, trainer, (o, d, cam, tgt)


def run_ngp(args, device, synthetic_rays, cpu_baseline_ngp):
    """One step = NGPModel's training iteration on 4096 synthetic rays INCLUDING the occupancy-grid refresh of every 16th
    step (models/instant_ngp.py:149-163): `value` is the amortised rate (the line without it overstated the
    training rate by 1.8 x). config.ms_per_step_excluding_refresh isolates the kernel schedule."""
    from nerfstudio_amd import functional as F
    from nerfstudio_amd.utils import roofline as RL

    model, arena, tr, (o, d, cam, tgt) = build_ngp(device, synthetic_rays, args.ngp_module_path, refresh=not args.ngp_no_refresh)
    cfg, grid, n = model.config, model.occupancy_grid, RAYS_PER_GPU
    step0 = 512  # past the grid's warm-up (256 steps): a quarter of the cells + the occupied ones are refreshed
    args.steps = (args.steps + 15) // 16 * 16  # whole refresh periods: exactly K / 16 refreshes in the K timed steps
    it = step0
    for _ in range(max(1, args.warmup)):
        tr.train_iteration(it)
        it += 1
    it = step0 + ((it - step0 + 15) // 16) * 16 + 1  # the timed region starts right after a refresh step: K / 16 refreshes in K steps
    torch.cuda.synchronize()
    tr.refreshes.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.train_iteration(it)
        it += 1
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(loss)), "training diverged"
    refreshes = list(tr.refreshes)
    kept = tr.kept_per_step(args.steps)
    # ---- per-kernel table (eager launches through the binding, HIP events on the launch stream; no refresh steps) ----
    prof_steps = max(1, args.profile_steps)
    tr.refresh = False
    prof = RL.profile_table(lambda: [tr.train_iteration(0) for _ in range(prof_steps)], prof_steps)
    table = sorted(((k, c / prof_steps, tot / prof_steps, mean) for k, (c, tot, mean) in prof.items()), key=lambda r: -r[2])
    if args.kernel_table:
        for k, c, ms, mean in table:
            print(f"{k:64s} {c:5.1f}/step {ms:9.4f} ms/step {mean:9.4f} ms/launch", file=sys.stderr)
    cand = F.occgrid_march(tr.rb.origins, tr.rb.directions, grid.binaries, grid._roi, cfg.render_step_size, cfg.near_plane,
                           cfg.far_plane, None, None, cfg.cone_angle, torch.rand(n, device=device), coarse=grid._coarse)
    n_cand = int(cand[0].numel())
    lattice = int(ngp_lattice_steps(o, d, cfg.render_step_size, cfg.cone_angle, cfg.near_plane, cfg.far_plane, cfg.grid_levels).sum())
    # roofline of the dominant PACKED kernel: the occupancy march. Algorithmic bytes per step: one occupancy byte per lattice
    # step and marching pass, 16 B per emitted sample (ray index, t_start, t_end), 24 B in + 20 B out per ray
    # (the explicit schedule marches every ray ONCE — nsamd_occgrid_march_count_stash — and its second launch copies the
    #  stashed steps; the module path marches twice)
    march = [(c, mean) for k, c, _, mean in table if k.startswith("nsamd_occgrid_march")]
    march_ms = sum(c * mean for c, mean in march)
    passes = sum(c for k, c, _, _ in table if k.startswith("nsamd_occgrid_march") and "write_stashed" not in k)
    packed = [(k, ms) for k, _, ms, _ in table if "occgrid" in k or "packed" in k]
    march_bytes = passes * lattice + 16 * n_cand + 44 * n
    roof = {"bound": "hbm", "achieved": round(march_bytes / (march_ms * 1e-3) / 1e9, 2), "peak": RL.HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(march_bytes / (march_ms * 1e-3) / 1e9 / RL.HBM_PEAK_GBS, 5), "traffic": None,
            "kernel": "nsamd_occgrid_march*", "avg_launch_ms": round(march_ms, 4), "algorithmic_per_launch": march_bytes,
            "rocprof_kernel": "nsamd::occgrid_march_kernel", "marching_passes_per_step": passes,
            "note": "latency-bound by construction: 1 B of grid per lattice step; lattice steps/s = "
                    f"{passes * lattice / (march_ms * 1e-3):.3e}"}
    top = next(((k, mean) for k, _, _, mean in table if RL.algorithmic_model_ngp(k, kept) is not None), None)
    roof_step = None
    if top is not None:
        bound, work = RL.algorithmic_model_ngp(top[0], kept)
        roof_step = RL.roofline_entry(top[0], top[1], bound, work, kept)
    ms = elapsed / args.steps * 1e3
    refresh_ms = float(np.median(refreshes)) if refreshes else None
    out = {
        "metric": "training rays/sec (4096 rays per GPU, instant-ngp packed path)",
        "value": round(RAYS_PER_GPU / (elapsed / args.steps), 1), "unit": "rays/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "instant-ngp 1xMI355X (BASELINE configs[3]): occupancy-grid refresh every 16th step INSIDE the timed "
                               "iteration, occupancy-grid ray marching (128^3 x 4 levels, random 5 % occupied), packed transmittance "
                               "scan with early termination + compaction, NerfactoField (L=16 hash T=2^19, 64x2 MLP) on the surviving "
                               "samples, packed compositing, MSE, backward, Adam; 4096 rays/batch",
                   "rays_per_gpu": RAYS_PER_GPU, "lattice_steps_per_ray": round(lattice / n, 1),
                   "candidate_samples_per_ray": round(n_cand / n, 2), "kept_samples_per_ray": round(kept / n, 2),
                   "field_density": NGP_DENSITY, "render_step_size": cfg.render_step_size, "cone_angle": cfg.cone_angle,
                   "alpha_thre": cfg.alpha_thre, "params": arena.numel, "final_loss": round(float(loss), 6),
                   "grid_refreshes_in_timed_region": len(refreshes), "grid_refresh_ms": None if refresh_ms is None else round(refresh_ms, 3),
                   "ms_per_step_excluding_refresh": round(ms - sum(refreshes) / args.steps, 4),
                   "launch": "eager (module / autograd path)" if tr.runner is None else
                   "explicit kernel schedule over capacity-sized buffers (ngp_step.py), eager launches",
                   "packed_kernels_ms_per_step": {k: round(v, 4) for k, v in packed}},
        "roofline": roof, "roofline_step": roof_step,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_ngp(o, d, cam, tgt, grid.binaries.cpu().numpy().astype(bool), cfg)
    print(json.dumps(out))
