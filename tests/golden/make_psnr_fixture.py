#!/usr/bin/env python3
"""CPU oracle training runs on the procedural scene of tests/psnr_scene.py -> tests/golden/psnr_scene_s<seed>.npz: per-step
losses (rgb, interlevel, distortion), the update schedule / anneal values, the PSNR of EVERY view (120 training, 20 held
out), four rendered images — and the same for a TWIN run whose initial tables differ by 1e-6 relative: the spread between
two correct fp32 trainings of this (chaotic) optimisation, which is what an implementation difference must be judged by.
The oracle is pinned to the reference by the other fixtures (make_golden.py); these pin the GPU path's TRAINING OUTCOME
to the oracle's (PSNR stand-in), on three independent seeds.
Run from the repository root:  python tests/golden/make_psnr_fixture.py [seed ...]   (about 8 minutes per seed on 2 threads)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import psnr_scene as S  # noqa: E402
from oracle import nerfacto_oracle as orc  # noqa: E402

torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "8")))
MAIN_LOG2, PROP_LOG2, SEED0 = 14, 12, 41


def schedule(cb_step, steps_since_update):
    """ProposalNetworkSampler update rule (ray_samplers.py:590) with nerfacto's schedule (nerfacto.py:208-213).
    `cb_step` is the sampler's own `_step`: set by step_cb AFTER a training iteration (ray_samplers.py:571-574), so during
    iteration k it still holds k - 1 (0 for the first two iterations)."""
    every = float(np.clip(np.interp(cb_step, [0, 5000], [0, 5]), 1, 5))
    return steps_since_update > every or cb_step < 10


def anneal_at(step, slope=10.0, n=1000):
    frac = float(np.clip(step / n, 0, 1))
    return slope * frac / ((slope - 1) * frac + 1)


def train(seed, perturb=0.0):
    """One oracle training run -> (params, per-step [rgb, interlevel, distortion] losses, schedule, anneals).
    perturb > 0 multiplies the initial hash tables by (1 + perturb * N(0,1)): the "twin" run that measures how far two
    CORRECT fp32 trainings of this problem drift apart (the optimisation is chaotic; tests compare against that spread)."""
    cfg = orc.NerfactoCfg(main_grid=orc.HashGridCfg(16, 16, 2048, MAIN_LOG2),
                          prop_grids=(orc.HashGridCfg(5, 16, 128, PROP_LOG2), orc.HashGridCfg(5, 16, 256, PROP_LOG2)),
                          num_images=S.N_TRAIN, appearance_embed_dim=0)  # eval and training see the same network
    params = orc.init_params(cfg, seed=SEED0 + seed)
    if perturb > 0:
        rs = np.random.RandomState(1234 + seed)
        for k, p in params.items():
            if "hash_table" in k:
                p.mul_(torch.from_numpy(1.0 + perturb * rs.standard_normal(tuple(p.shape)).astype(np.float32)))
    names = list(params)
    for p in params.values():
        p.requires_grad_(True)
    groups = {"proposal_networks": [params[k] for k in names if k.startswith("proposal_networks")],
              "fields": [params[k] for k in names if k.startswith("field")]}
    opts = {g: torch.optim.Adam(ps, lr=1e-2, eps=1e-15) for g, ps in groups.items()}
    losses, sched, anneals = [], [], []
    since, cb_step = 0, 0
    for step, (o, d, cam, tgt, jit) in enumerate(S.batches(seed=9 + seed)):
        upd = schedule(cb_step, since)
        an = anneal_at(step)
        for opt in opts.values():
            opt.zero_grad(set_to_none=True)
        j = [S.to_t(jit[i])[:, None] for i in range(3)]
        out = orc.nerfacto_forward(params, cfg, S.to_t(o), S.to_t(d), S.to_t(cam), j, training=True, anneal=an,
                                   proposal_requires_grad=upd)
        ld = orc.nerfacto_losses(out, S.to_t(tgt), cfg)
        sum(ld.values()).backward()
        opts["fields"].step()
        if upd:  # a group is stepped only when it received gradients (engine/optimizers.py:160-172)
            opts["proposal_networks"].step()
            since = 0
        cb_step = step  # step_cb(step): AFTER_TRAIN_ITERATION
        since += 1
        losses.append([float(ld[k].detach()) for k in ("rgb_loss", "interlevel_loss", "distortion_loss")])
        sched.append(upd)
        anneals.append(an)
        if step % 50 == 0:
            print(f"seed {seed} perturb {perturb:g} step {step:4d} loss {sum(losses[-1]):.5f} updated {upd}", flush=True)
    return cfg, params, np.array(losses, np.float64), np.array(sched), np.array(anneals, np.float64)


def evaluate(cfg, params):
    """Eval-mode renders (no jitter, near plane 0, clamp) of every view: per-view PSNR [N_TRAIN + N_HELD_OUT] and the
    images of S.EVAL_CAMERAS."""
    psnrs, images = [], {}
    for cam_id in S.ALL_CAMERAS:
        o, d, gt = S.full_view(cam_id)
        with torch.no_grad():
            ev = orc.nerfacto_forward(params, cfg, S.to_t(o), S.to_t(d), torch.zeros(len(o), dtype=torch.int64), None,
                                      training=False)
        img = ev["rgb"].numpy().astype(np.float32)
        psnrs.append(S.psnr(img, gt))
        if cam_id in S.EVAL_CAMERAS:
            images[cam_id] = img
    return np.array(psnrs), np.stack([images[c] for c in S.EVAL_CAMERAS])


def main(seed):
    cfg, params, losses, sched, anneals = train(seed)
    psnr_views, images = evaluate(cfg, params)
    _, params_t, losses_t, _, _ = train(seed, perturb=1e-6)
    psnr_twin, _ = evaluate(cfg, params_t)
    tr, ho = slice(0, S.N_TRAIN), slice(S.N_TRAIN, None)
    print(f"seed {seed}: oracle PSNR mean over {S.N_TRAIN} training views {psnr_views[tr].mean():.3f} dB (twin {psnr_twin[tr].mean():.3f}), "
          f"over {S.N_HELD_OUT} held-out views {psnr_views[ho].mean():.3f} dB (twin {psnr_twin[ho].mean():.3f}); "
          f"largest single-view difference to the twin {np.abs(psnr_views - psnr_twin).max():.3f} dB")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"psnr_scene_s{seed}.npz"), losses=losses, losses_twin=losses_t,
                        schedule=sched, anneals=anneals, images=images, psnr=np.array([psnr_views[c] for c in S.EVAL_CAMERAS]),
                        psnr_views=psnr_views, psnr_views_twin=psnr_twin, cfg=np.array([MAIN_LOG2, PROP_LOG2, SEED0 + seed]))


if __name__ == "__main__":
    for seed_ in ([int(a) for a in sys.argv[1:]] or list(S.SEEDS)):
        main(seed_)
