#!/usr/bin/env python3
"""bench.py — training rays/sec of the nerfacto hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full nerfacto training iteration on a batch of 4096 synthetic rays PER GPU (BASELINE configs[1]:
L=16 hash T=2^19 F=2, 64-wide MLPs, proposal sampler 256 -> 96 -> 48 samples/ray): proposal sampling, main field,
compositing, MSE + interlevel + distortion losses, backward of all of it, gradient all-reduce (N > 1), Adam over all
19.4 M parameters, and the reference's per-step callbacks (proposal update schedule, weight anneal). Rays are
resident in HBM before the timed region. value = world_size * rays_per_batch / time_per_step, the reference's own
rays/s definition (engine/trainer.py:276-284) with a device sync around the timed region.

Extra objects on the JSON line:
  roofline     — the kernel with the largest share of the step, timed live with HIP events on the launch stream
                 (separate, untimed profiling steps after the timed region), against its algorithmic bytes / flops.
  cpu_baseline — the CPU oracle (oracle/nerfacto_oracle.py, kind "port") running the same training step on a bounded
                 ray sample on this box's host cores (rank 0, N = 1 only).
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak


def build_model(device, seed):
    from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

    torch.manual_seed(seed)
    model = NerfactoModel(NerfactoModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=100)
    return model.to(device).train()


BATCH_SLOTS = 8  # pre-generated ray batches resident in HBM; the timed loop takes a different one every step


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


class Trainer:
    """The reference's Trainer.train_iteration (engine/trainer.py:487-531) for this path, minus logging.

    Eager mode runs the Python body every step. Graph mode captures that same body ONCE per schedule variant
    (proposal networks updated this step / not updated, ray_samplers.py:590) into a hipGraph and replays it: ~60
    kernel launches become one graph launch, which is what a 1-2 ms step needs (MI355X_MICROARCH.md price list:
    eager goes host-bound below ~3 us per kernel). Everything that changes from step to step lives in device memory:
    the ray batch, the jitter draws (graph-safe Philox), the anneal exponent and Adam's bias-corrected step size
    (`hyper`, refreshed by a 20-byte async copy from a ring of pinned host slots before each replay).

    N = 1 with graphs (default): the main-field Adam of iteration k is the first node of iteration k+1's graph, on a branch
    beside select-batch / jitter / the proposal forward (`_deferred_iteration_body`; four captured variants: proposal
    update x pending Adam). Same dependencies as Adam at the end of the iteration, hence the same bits; `finish()` runs the
    last pending update inside the timed region.

    N > 1 (data parallel): the iteration runs as segments (eager launches by default, captured hipGraphs with --dp-graph)
    and the 67 MB main-field all-reduce (RCCL, its own stream) is PIPELINED across steps (nerfstudio_amd/dp_schedule.py). The proposal forward of step k+1 reads only proposal-network parameters, so
        step k:   [proposal fwd k] -> (wait AR_main k-1) [Adam main k-1] -> [main fwd + losses + main bwd k]
                  -> AR_main k (async) -> [proposal bwd k] -> AR_props k -> [Adam props k]      (last two: update steps)
    hides the all-reduce behind the proposal backward of step k AND the proposal forward of step k+1, with exactly the
    sequential semantics (every parameter is updated before its next use). `finish()` drains the pending update."""

    def __init__(self, model, arena, ray_bundle, batch, world=1, use_graph=True, use_runner=True, pool=None,
                 force_dp=False, dp_mode="allreduce"):
        self.model, self.arena, self.rb, self.batch, self.world = model, arena, ray_bundle, batch, world
        self.dp = world > 1 or force_dp  # force_dp: the data-parallel schedule with a one-rank communicator
        # "sharded": reduce-scatter -> Adam on the rank's 1/N arena shard -> all-gather (dp_schedule.py); "allreduce": the
        # replicated optimiser behind an all-reduce (the reference's DDP semantics, and the default)
        self.dp_sharded = self.dp and dp_mode == "sharded"
        self.dp_fork = False  # set below: proposal backward chains beside the main chain in the data-parallel schedule
        self.pool = pool  # BATCH_SLOTS pre-generated batches in HBM (None: one fixed batch)
        self.step = 0
        self.opt_step = 0
        self._true_steps = dict(arena.step_counts)
        dev = ray_bundle.origins.device
        # device-resident step-dependent scalars: Adam (step size, 1/sqrt(bc2)) per optimiser group + the anneal exponent
        self.hyper = torch.zeros(6, device=dev)  # [5] = batch slot of this step
        # The host runs ahead of the GPU, so the pinned source of an async copy must not be rewritten before the copy
        # has executed: a ring of slots, each guarded by the event recorded after its last copy.
        self.hyper_ring = [torch.zeros(6).pin_memory() for _ in range(64)]
        self.hyper_events = [None] * 64
        self.hyper_slot = 0
        from nerfstudio_amd.schedulers import nerfacto_schedulers

        self.schedulers = nerfacto_schedulers()
        self.exchange = None  # nerfstudio_amd.dp_schedule.PipelinedExchange (N > 1 with the runner)
        self.hyper_views = {"fields": self.hyper[0:2], "proposal_networks": self.hyper[2:4]}
        self.loss_buf = torch.zeros((), device=dev)
        model.proposal_sampler.anneal_dev = self.hyper[4:5]
        self.graphs = None
        self.use_graph = use_graph
        self.runner = None
        self.defer = False
        self.defer_scatter = False
        self.opt_parallel = True  # False: the deferred Adam runs on the main stream (per-kernel timing)
        # False: the jitter buffer of the runner is filled by the caller before every iteration (parity tests inject the
        # draws the CPU oracle uses; the default draws them on the device inside the iteration, graph-safe Philox)
        self.draw_jitter = True
        self._pending_main = False  # deferred schedule: the main-field Adam of the previous iteration is still to run
        if use_runner:  # explicit kernel schedule over static buffers (nerfstudio_amd/train_step.py); default
            from nerfstudio_amd.train_step import NerfactoTrainStep

            self.runner = NerfactoTrainStep(model, ray_bundle.origins.shape[0], dev)
            self.runner.set_batch(ray_bundle.origins, ray_bundle.directions, ray_bundle.camera_indices, batch["image"])
            self.runner.anneal_dev = self.hyper[4:5]
            if os.environ.get("NSAMD_SIDE_STREAM", "1") == "0":  # A/B switch: proposal backward on the main stream
                self.runner.side_stream = None
            # N = 1: the main-field Adam of iteration k (470 MB of HBM streaming) runs BESIDE the proposal forward of
            # iteration k+1 (L2-resident gathers and per-ray scans that read only proposal-network parameters) — the
            # single-GPU form of the pipelined schedule above; same dependencies, same bits. Measured on three MI355X boxes
            # (profiles/r02_schedule_ab.txt): 1.3 / 3 / 4.5 % faster than Adam at the end of the iteration when replayed
            # from hipGraphs, neutral with eager launches — so it is the default with graphs. NSAMD_DEFER_MAIN_ADAM=0/1: A/B.
            self.defer = not self.dp and os.environ.get("NSAMD_DEFER_MAIN_ADAM", "1" if use_graph else "0") == "1"
            # NSAMD_DEFER_SCATTER=1 (opt-in, measured and NOT adopted: profiles/r03_negative_results.txt item 8) defers the
            # main TABLE SCATTER of iteration k as well: [scatter k -> Adam main k] becomes one branch of iteration k+1's
            # graph beside [select batch, proposal forward k+1]; the scatter reads copies of iteration k's ray origins /
            # directions / bin edges (0.9 MB, taken beside the main forward) because the next batch overwrites them. Same
            # bits (parameter checksums equal), but 1-2 % SLOWER in the driver window: the route kernel's 768 x 1024-thread
            # workgroups hold every wave slot, the small latency-bound kernels of the proposal forward wait for slots and
            # their branch becomes the long one.
            self.defer_scatter = self.defer and os.environ.get("NSAMD_DEFER_SCATTER", "0") == "1"
            if self.defer:
                self.opt_stream = torch.cuda.Stream(device=dev)
                self._opt_fork, self._opt_join = torch.cuda.Event(), torch.cuda.Event()
                self._sh_fork, self._sh_join = torch.cuda.Event(), torch.cuda.Event()
            if self.dp:
                from nerfstudio_amd.dp_schedule import PipelinedExchange

                # the pending main-field Adam waits for its all-reduce on its own stream, beside the next proposal forward
                # (NSAMD_DP_UPDATE_STREAM=0: on the launch stream after it, the round-2 order)
                # (measured on a one-rank communicator, profiles/r03_dp_rehearsal.txt: 0.992 vs 0.969 ms — off by default)
                upd = torch.cuda.Stream(device=dev) if os.environ.get("NSAMD_DP_UPDATE_STREAM", "0") == "1" else None
                self.exchange = PipelinedExchange(arena, self._run, before_main_update=self._push_hyper,
                                                  sharded=self.dp_sharded, update_stream=upd)
                # eager segments only (a captured segment must end with its streams joined); NSAMD_DP_FORK=0: round-2 order
                self.dp_fork = os.environ.get("NSAMD_DP_FORK", "1") == "1" and self.runner.side_stream is not None
                # the coarse levels of the main table can only ever touch 288 k of their 2.6 M rows: exchange those
                # compactly (2.3 MB instead of 21 MB of the 67 MB main-field all-reduce)
                enc = model.field.mlp_base.encoding
                rows, index = enc.spec.reachable_prefix()
                if index.numel() and index.numel() < rows // 2 and not self.dp_sharded:
                    arena.register_compact(enc.hash_table, rows, index)  # (the reduce-scatter takes the slice as it lies)

    # -- pieces of one iteration ---------------------------------------------------------------------------------
    def _prologue(self, updated):
        from nerfstudio_amd import functional as F

        m, a = self.model, self.arena
        m.set_step(self.step)  # BEFORE_TRAIN_ITERATION callback: proposal weight anneal
        self._push_hyper()

    def _push_hyper(self):
        """Adam step sizes of the NEXT update of each group + the anneal exponent -> device (async, race-free)."""
        from nerfstudio_amd import functional as F

        m, a = self.model, self.arena
        slot = self.hyper_slot
        self.hyper_slot = (slot + 1) % len(self.hyper_ring)
        if self.hyper_events[slot] is not None:
            self.hyper_events[slot].synchronize()  # the copy that last read this slot (64 pushes ago) is done
        h = self.hyper_ring[slot]
        # learning rates of the nerfacto recipe (method_configs.py:110-121): iteration i runs with lr_init * decay(i); a
        # pending (pipelined) main-field update belongs to the previous iteration
        it_fields = self.step - 1 if self._have_pending else self.step
        lr_f = self.schedulers["fields"].get_lr(max(it_fields, 0), a.lr)
        lr_p = self.schedulers["proposal_networks"].get_lr(self.step, a.lr)
        h[0], h[1] = F.adam_hyper(a.step_counts["fields"] + 1, lr_f, a.betas)
        h[2], h[3] = F.adam_hyper(a.step_counts["proposal_networks"] + 1, lr_p, a.betas)
        h[4] = m.proposal_sampler._anneal
        h[5] = float(self.step % BATCH_SLOTS)
        self.hyper.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.hyper_events[slot] = ev

    def _fwd_bwd(self, updated):
        """Single-process path: forward, losses and the main backward (runner: also the proposal backward)."""
        from nerfstudio_amd.cameras.rays import RayBundle

        if self.runner is not None:
            self._select_batch()
            # the main table's gradient is written, not accumulated; the proposal group's gradients are neither produced nor
            # consumed on a step that does not update it (ray_samplers.py:590-599), so its 10 MB need no zero-fill then
            groups = ["fields", "proposal_networks"] if updated else ["fields"]
            self.arena.zero_grad(groups, skip=self.runner.written_params())
            self.runner.forward_backward(updated, self.draw_jitter)  # the backward chains run as parallel branches
            return
        self._select_batch()
        self.arena.zero_grad()
        m = self.model
        m.proposal_sampler.force_updated = updated
        rb = RayBundle(origins=self.rb.origins, directions=self.rb.directions, pixel_area=self.rb.pixel_area,
                       camera_indices=self.rb.camera_indices)
        out = m(rb)
        metrics = m.get_metrics_dict(out, self.batch)
        loss_dict = m.get_loss_dict(out, self.batch, metrics)
        loss = loss_dict["rgb_loss"] + loss_dict["interlevel_loss"] + loss_dict["distortion_loss"]
        loss.backward()
        self.loss_buf.copy_(loss.detach())

    def _deferred_iteration_body(self, updated, pending):
        """One iteration of the deferred schedule (N = 1, runner):
            [Adam main k-1  ||  select batch, proposal forward k] -> main forward, losses, backward chains k
            -> [Adam proposals k]                                                            (update steps)
        Inside a captured hipGraph the two halves of the first line are parallel branches."""
        r, a = self.runner, self.arena
        main = torch.cuda.current_stream()
        beside = pending and self.opt_parallel

        def pending_update():  # what iteration k-1 left behind: [its table scatter ->] its main-field Adam
            if self.defer_scatter:
                r.backward_table(shadow=True)
            a.step(grad_scale=1.0, groups=["fields"], hyper_dev=self.hyper_views)

        if beside:
            self._opt_fork.record(main)
            self.opt_stream.wait_event(self._opt_fork)
            with torch.cuda.stream(self.opt_stream):
                pending_update()
                self._opt_join.record(self.opt_stream)
        elif pending:
            pending_update()
        self._select_batch()
        r.apply_camera_corrections()
        r.forward_proposals(self.draw_jitter, need_enc=updated)
        if beside:
            main.wait_event(self._opt_join)
        if self.defer_scatter:  # the final samples are known: copy what defines them, beside the main forward
            if self.opt_parallel:
                self._sh_fork.record(main)
                self.opt_stream.wait_event(self._sh_fork)
                with torch.cuda.stream(self.opt_stream):
                    r.shadow_points()
                    self._sh_join.record(self.opt_stream)
            else:
                r.shadow_points()
        groups = ["fields", "proposal_networks"] if updated else ["fields"]
        a.zero_grad(groups, skip=r.written_params())
        r.forward_main_and_losses(updated)
        r.defer_table = self.defer_scatter
        try:
            r.backward_all(updated)
        finally:
            r.defer_table = False
        if self.defer_scatter and self.opt_parallel:
            main.wait_event(self._sh_join)
        if updated:
            a.step(grad_scale=1.0, groups=["proposal_networks"], hyper_dev=self.hyper_views)

    def _select_batch(self):
        """This step's rays out of the HBM-resident pool (slot index in device memory: replayable) — the hand-over the
        reference's datamanager does each iteration (base_datamanager.py:506-515)."""
        if self.pool is None:
            return
        from nerfstudio_amd import _native as N

        p = self.pool
        if self.runner is not None:
            r = self.runner
            o, d, c, t = r.origins, r.directions, r.camera_indices, r.target
        else:
            o, d, c, t = self.rb.origins, self.rb.directions, self.rb.camera_indices, self.batch["image"]
        N.check(N.load().nsamd_select_batch(N.ptr(self.hyper[5:6]), BATCH_SLOTS, o.shape[0], N.ptr(p["origins"]),
                                            N.ptr(p["directions"]), N.ptr(p["cameras"]), N.ptr(p["target"]), N.ptr(o),
                                            N.ptr(d), N.ptr(c), N.ptr(t), N.stream()), "select_batch")

    def _optimise(self, updated):
        # the reference steps an optimiser group only when it received gradients (engine/optimizers.py:160-172)
        groups = ["fields", "proposal_networks"] if updated else ["fields"]
        self.arena.step(grad_scale=1.0 / self.world, groups=groups, hyper_dev=self.hyper_views)

    # -- data-parallel segments (N > 1, runner) --------------------------------------------------------------------------
    @property
    def pipelined(self):
        return self.dp and self.runner is not None

    def _seg(self, name):
        """The body of one captured segment (also what the eager path runs)."""
        r, a = self.runner, self.arena
        if name == "pfwd":
            self._select_batch()
            r.forward_proposals(self.draw_jitter)
        elif name in (("main", True), ("main", False)):
            if name[1] and self.dp_fork:
                # update step, eager launches: the proposal chains start on their side streams here, beside the main chain
                # (as in the N = 1 schedule) — and, since the exchange starts the main-field collective right after this
                # segment, beside that too; "pbwd" only joins them
                a.zero_grad(["fields", "proposal_networks"], skip=r.written_params())
                r.forward_main_and_losses(True)
                r.backward_fork(True)
            else:
                a.zero_grad(["fields"], skip=r.written_params())
                r.forward_main_and_losses(name[1])
                r.backward_main()
        elif name == "pbwd":
            if self.dp_fork:
                r.backward_join(True)
            else:
                a.zero_grad(["proposal_networks"], skip=r.written_params())
                r.backward_proposals()
        elif name in ("mopt", "popt"):
            grp = "fields" if name == "mopt" else "proposal_networks"
            if self.dp_sharded:  # this rank's 1/N of the group; the exchange all-gathers the updated parameters
                a.step_shard(grp, grad_scale=1.0 / self.world, hyper_dev=self.hyper_views)
            else:
                a.step(grad_scale=1.0 / self.world, groups=[grp], hyper_dev=self.hyper_views)
        else:
            raise KeyError(name)


    def _run(self, name):
        if os.environ.get("NSAMD_DP_TIMING") == "1":  # diagnostics: host-synchronous per-segment timing
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self._run_inner(name)
            torch.cuda.synchronize()
            self._seg_times = getattr(self, "_seg_times", {})
            self._seg_times.setdefault(str(name), []).append((time.perf_counter() - t0) * 1e3)
            return
        self._run_inner(name)

    def _run_inner(self, name):
        if self.graphs is not None:
            self.graphs[name].replay()
            if name == "mopt":
                self.arena.step_counts["fields"] += 1  # the replayed Adam launch did step the group
            elif name == "popt":
                self.arena.step_counts["proposal_networks"] += 1
        else:
            self._seg(name)

    def finish(self):
        """Drain the data-parallel pipeline (no-op for N = 1)."""
        if self.exchange is not None:
            self.exchange.finish()
        if self._pending_main:  # deferred schedule: the last iteration's [table scatter and] main-field update
            self._push_hyper()
            if self.defer_scatter:
                self.runner.backward_table(shadow=True)
            self.arena.step(grad_scale=1.0, groups=["fields"], hyper_dev=self.hyper_views)
            self._pending_main = False
            self._true_steps = dict(self.arena.step_counts)

    @property
    def _have_pending(self):
        return self._pending_main or (self.exchange is not None and self.exchange.pending)

    def _pipelined_iteration(self, updated):
        self._prologue(updated)
        self.exchange.iteration(updated)

    def _plain_dp_iteration(self, updated):
        """N > 1 through the autograd modules: one blocking all-reduce of the whole arena (not pipelined)."""
        self._prologue(updated)
        self._fwd_bwd(updated)
        self.arena.all_reduce()
        self._optimise(updated)

    # -- graph capture ---------------------------------------------------------------------------------------------
    def warm_variants(self):
        """One eager iteration of each schedule variant (proposal networks updated / not) on a side stream — allocator and
        lazy-attribute warm-up ahead of a capture. They are real training iterations (parameters and Adam state move) that
        do not advance the step counter; an eager run that is to train through the same states as a captured one calls
        this at the same point (tests/test_gpu_bench_parity.py)."""
        torch.cuda.synchronize()
        assert not self._have_pending
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            defer, self.defer = self.defer, False  # (in order, so that every schedule trains through the same states)
            for upd in (True, False):
                self._eager_iteration(upd)
            self.finish()
            self.defer = defer
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def capture(self):
        self.warm_variants()
        graphs = {}
        if self.pipelined:
            from nerfstudio_amd.dp_schedule import SEGMENTS

            self.dp_fork = False  # captured segments keep the proposal backward in its own segment
            for name in SEGMENTS:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._seg(name)
                graphs[name] = g
        elif self.defer:
            for upd in (True, False):
                for pend in (True, False):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):  # the whole iteration is one graph
                        self._deferred_iteration_body(upd, pend)
                    graphs[("all", upd, pend)] = g
        else:
            for upd in (True, False):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):  # the whole iteration is one graph
                    self._fwd_bwd(upd)
                    self._optimise(upd)
                graphs[("all", upd)] = g
        for name in self.arena.step_counts:  # captures executed nothing; undo the host-side counters they bumped
            self.arena.step_counts[name] = self._true_steps[name]
        self.graphs = graphs

    def _eager_iteration(self, updated):
        if self.pipelined:
            self._pipelined_iteration(updated)
        elif self.dp:
            self._plain_dp_iteration(updated)
        elif self.defer:
            self._prologue(updated)
            self._deferred_iteration_body(updated, self._pending_main)
            self._pending_main = True
        else:
            self._prologue(updated)
            self._fwd_bwd(updated)
            self._optimise(updated)
        self._true_steps = dict(self.arena.step_counts)

    def try_capture(self):
        if not self.use_graph or (self.dp and not self.pipelined):
            return False
        try:
            self.capture()
            return True
        except Exception as e:  # noqa: BLE001 - any capture problem degrades to the eager path, never to no result
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            self.graphs = None
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            return False

    # -- one training iteration ------------------------------------------------------------------------------------
    def train_iteration(self):
        ps = self.model.proposal_sampler
        updated = ps.updated_this_step()
        if self.graphs is None or self.pipelined:
            self._eager_iteration(updated)  # (pipelined: the segments replay their graphs)
        else:
            self._prologue(updated)
            if self.defer:
                self.graphs[("all", updated, self._pending_main)].replay()
                stepped = (("fields",) if self._pending_main else ()) + (("proposal_networks",) if updated else ())
                self._pending_main = True
            else:
                self.graphs[("all", updated)].replay()
                stepped = ("fields", "proposal_networks") if updated else ("fields",)
            for name in stepped:
                self.arena.step_counts[name] += 1  # the replayed Adam launches did step these groups
            self._true_steps = dict(self.arena.step_counts)
        self.opt_step += 1
        if updated:
            ps.mark_updated()
        self.model.after_step(self.step)  # AFTER_TRAIN_ITERATION callback
        self.step += 1
        return self.loss_buf

    def last_loss(self):
        if self.runner is not None:
            return sum(self.runner.loss_dict().values())
        return self.loss_buf


# algorithmic work per launch: bytes for the HBM-bound kernels, flops for the MFMA kernels
def algorithmic_model(key):
    import re

    m = re.search(r"L=(\d+),M=(\d+)", key)
    if key.startswith("nsamd_hashgrid_encode_fwd") and m:
        L, M = int(m.group(1)), int(m.group(2))
        return "hbm", M * L * 8 * 8  # 8 corner gathers x 8 B (F=2 fp32) per level and sample
    if key.startswith("nsamd_hashgrid_encode_bwd") and m:
        L, M = int(m.group(1)), int(m.group(2))
        return "hbm", M * L * 8 * 16  # read-modify-write of 8 corners x 8 B
    M_main = RAYS_PER_GPU * 48
    if key in ("nsamd_field_mlp_fwd", "nsamd_field_mlp_fwd_save"):
        return "mfma", M_main * 2 * 11392  # MACs/sample: 32*64 + 64*16 + 63*64 + 64*64 + 64*3  
    if key == "nsamd_field_fused_fwd":
        return "hbm", M_main * 16 * 8 * 8  # hash gathers (the bound of the fused launch; its MLP half is 4.5 GFLOP of MFMA)
    if key in ("nsamd_field_mlp_bwd", "nsamd_field_mlp_bwd_saved"):
        # training = 3x the forward FLOPs, the forward launch takes 1x, so the backward's ALGORITHMIC share is
        # 2x (data gradient + weight gradient). The recompute of the forward inside nsamd_field_mlp_bwd is executed work,
        # not algorithmic work: it is reported separately (`executed_per_launch`), never in `achieved` / `frac`.
        return "mfma", M_main * 2 * 11392 * 2
    m2 = re.search(r"\[M=(\d+)\]", key)
    if key.startswith("nsamd_density_mlp_fwd") and m2:
        return "hbm", int(m2.group(1)) * (10 * 4 + 4 + 8)  # enc row + selector in, density + pre out
    if key.startswith("nsamd_density_mlp_bwd") and m2:
        return "hbm", int(m2.group(1)) * (10 * 4 * 2 + 4 * 3)
    if key == "nsamd_adam_step":
        return "hbm", None  # filled in by the caller (arena size x 28 B)
    return None, None


# flops / bytes a launch actually executes where that differs from the algorithmic figure (reported next to it)
EXECUTED_PER_LAUNCH = {"nsamd_field_mlp_bwd": RAYS_PER_GPU * 48 * 2 * 11392 * 3}  # + the forward recompute

# entry point -> the kernel name rocprofv3 --kernel-trace --stats lists for it (profiles/*_kernel_stats.csv)
ROCPROF_KERNEL = {
    "nsamd_field_mlp_bwd": "nsamd::field_mlp_bwd_kernel",
    "nsamd_field_mlp_bwd_saved": "nsamd::field_mlp_bwd_kernel (saved activations)",
    "nsamd_field_mlp_fwd": "nsamd::field_mlp_fwd_kernel",
    "nsamd_hashgrid_encode_fwd": "nsamd::hash_encode_fwd_kernel",
    "nsamd_hashgrid_encode_bwd_set": "nsamd::scatter_route_fine_kernel + nsamd::scatter_apply_kernel + nsamd::scatter_finish_kernel",
    "nsamd_hashgrid_encode_bwd": "nsamd::scatter_route_* + nsamd::scatter_apply_kernel + nsamd::scatter_finish_kernel",
    "nsamd_hashgrid_encode_bwd_gated": "nsamd::scatter_route_* + nsamd::scatter_apply_kernel + nsamd::scatter_finish_kernel",
    "nsamd_adam_step": "nsamd::adam_kernel",
}


def kernel_sources_hash():
    """sha256 over the kernel sources: stamps profiles/pmc_traffic.json (scripts/collect_pmc.sh) so that a traffic
    figure measured on other kernels is never reported."""
    import glob
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(ROOT, "nerfstudio_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of `kernel_key` from the committed rocprofv3 PMC passes (scripts/collect_pmc.sh ->
    profiles/pmc_traffic.json: FETCH_SIZE, doubled for 16-B-per-lane streaming reads as MI355X_MICROARCH.md prescribes for
    gfx950, + WRITE_SIZE; separate --pmc passes). Counters cannot be read from inside this process, so the value is the
    one measured for this kernel by the PMC passes — and only if they ran on THESE kernel sources (the file carries
    their hash): None (JSON null) when the sources changed since, or the file has no entry for the kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        data = json.load(open(path))
        if data.get("_kernel_sources_sha256_16") != kernel_sources_hash():
            return None
        entry = data.get(kernel_key)
        return int(entry["hbm_bytes"]) if entry else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def measure_roofline(trainer, arena, steps):
    from nerfstudio_amd import _native as N

    graphs, trainer.graphs = trainer.graphs, None  # per-kernel events need eager launches
    runner = getattr(trainer, "runner", None)
    side = getattr(runner, "side_stream", None)
    if runner is not None:
        runner.side_stream = None  # one stream: a kernel's events must not include a concurrent branch's work
    trainer.opt_parallel = False
    N.PROFILE = {}
    for _ in range(steps):
        trainer.train_iteration()
    trainer.finish()
    torch.cuda.synchronize()
    prof = N.profile_summary(N.PROFILE)
    N.PROFILE = None
    trainer.graphs = graphs
    trainer.opt_parallel = True
    if runner is not None:
        runner.side_stream = side
    table = []
    for key, (calls, total_ms, mean_ms) in prof.items():
        bound, work = algorithmic_model(key)
        if key == "nsamd_adam_step":
            work = arena.numel * 28
        table.append({"kernel": key, "calls_per_step": calls / steps, "ms_per_step": total_ms / steps, "mean_ms": mean_ms,
                      "bound": bound, "work": work})
    table.sort(key=lambda r: -r["ms_per_step"])
    ranked = [r for r in table if r["bound"] is not None and r["work"]]

    def entry(top):
        sec = top["mean_ms"] * 1e-3
        if top["bound"] == "hbm":
            ach, peak, unit = top["work"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = top["work"] / sec / 1e12, F32_MFMA_PEAK_TFLOPS, "TFLOP/s"
        base = top["kernel"].split("[")[0]
        roof = {"bound": top["bound"], "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                "traffic": pmc_traffic(top["kernel"]), "kernel": top["kernel"], "avg_launch_ms": round(top["mean_ms"], 4),
                "algorithmic_per_launch": top["work"], "rocprof_kernel": ROCPROF_KERNEL.get(base)}
        if base in EXECUTED_PER_LAUNCH:  # the utilisation view (work the launch executes, incl. recomputation)
            ex = EXECUTED_PER_LAUNCH[base]
            roof["executed_per_launch"] = ex
            roof["executed_frac"] = round(ex / sec / (1e9 if top["bound"] == "hbm" else 1e12) / peak, 4)
        return roof

    roof = entry(ranked[0]) if ranked else None
    # The main-field MLP backward and the main-table scatter are within a few percent of each other per step (0.19 ms
    # both): which one is "the dominant kernel" flips between runs. The runner-up rides along so that both are in every line.
    if roof is not None and len(ranked) > 1:
        roof["runner_up"] = entry(ranked[1])
    return roof, table


def cpu_baseline(n_rays=RAYS_PER_GPU, steps=3, threads=None, workload="bounded"):
    """The CPU oracle (oracle/nerfacto_oracle.py, a restatement of the reference's torch path pinned to it by
    tests/golden) running the same training step — forward, losses, backward, Adam over all 19.4 M parameters — on the
    METRIC'S configuration: one batch of 4096 rays (BASELINE configs[1]); 1 warm-up step + `steps` timed ones, median
    (about 2-4 s per step). profiles/r02_cpu_reference_vs_port.txt holds the authoring-container comparison of this port
    with the reference's own modules on the same 4096 rays (within +-20 %).
    Thread count: torch's CPU ops on this workload peak at ~16 threads on the MI355X host (measured 8/16/32/64/128
    threads: 135/141/111/63/30 rays/s on the 256-ray sample of round 1), so 16 is used rather than all cores."""
    from oracle import nerfacto_oracle as orc

    if threads is None:
        threads = min(16, os.cpu_count() or 16)
    torch.set_num_threads(threads)

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


# ---------------------------------------------------------------------------------------------------------------------
# --workload ngp: BASELINE configs[3] — instant-ngp: occupancy-grid ray marching + early termination (packed samples)
# ---------------------------------------------------------------------------------------------------------------------
NGP_DENSITY = 60.0  # synthetic field: sigma ~ 60 -> alpha ~ 0.19 per step, a ray is opaque (T < 1e-4) after ~45 samples


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


def run_ngp(args, device):
    """One step = NGPModel's training iteration on 4096 synthetic rays: occupancy-grid march (count / prefix / write), density
    on the candidates, packed transmittance scan with early termination + compaction, NerfactoField on the survivors, packed
    weights + compositing, MSE, backward (packed scans, field MLPs, table scatter), fused Adam. Through the module / autograd
    path of nerfstudio_amd.instant_ngp (eager launches; the packed arrays are allocated to size each step, as the reference
    does). The occupancy grid is SYNTHETIC and FIXED (random 5 %-occupied 128^3 x 4 levels) and the random
    field's density is lifted to ~60 so that rays become opaque after ~45 kept samples: the refresh of the grid (every 16th
    step in training) would replace the synthetic grid by the random field's own and is timed separately
    (config.grid_refresh_ms)."""
    from nerfstudio_amd import _native as N
    from nerfstudio_amd import functional as F
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.cameras.rays import RayBundle
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
    arena = ParamArena({"fields": list(model.field.parameters())}, lr=1e-2, eps=1e-15)
    o, d, cam, tgt = synthetic_rays(1000)
    n = RAYS_PER_GPU
    rb = RayBundle(origins=torch.from_numpy(o).to(device), directions=torch.from_numpy(d).to(device),
                   pixel_area=torch.full((n, 1), 1e-6, device=device), camera_indices=torch.from_numpy(cam).to(device))
    batch = {"image": torch.from_numpy(tgt).to(device)}
    samples = []
    runner = None
    if not args.ngp_module_path:
        # the explicit kernel schedule over capacity-sized buffers (nerfstudio_amd/ngp_step.py); --ngp-module-path: the same
        # iteration through the nn.Module / autograd classes (host-bound: profiles/r03_final_bench_ngp.json)
        from nerfstudio_amd.ngp_step import NgpTrainStep

        runner = NgpTrainStep(model, n, device)
        runner.set_batch(rb.origins, rb.directions, rb.camera_indices, batch["image"])
        table_param = model.field.mlp_base.encoding.hash_table

    def step():
        if runner is not None:
            arena.zero_grad(skip=[table_param])  # the scatter writes the table's gradient
            runner.forward()
            loss = runner.loss()
            runner.backward()
            arena.step()
            samples.append(runner.num_kept)
            return loss
        arena.zero_grad()
        out = model(rb)
        loss = model.get_loss_dict(out, batch)["rgb_loss"]
        loss.backward()
        arena.step()
        samples.append(out["num_samples_per_ray"])
        return loss

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(loss)), "training diverged"
    kept = float(np.mean(samples[-args.steps:])) if runner is not None else \
        float(torch.stack(samples[-args.steps:]).float().sum(dim=1).mean())
    # ---- per-kernel table (eager launches through the binding, HIP events on the launch stream) ----
    N.PROFILE = {}
    prof_steps = max(1, args.profile_steps)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = N.profile_summary(N.PROFILE)
    N.PROFILE = None
    table = sorted(((k, c / prof_steps, tot / prof_steps, mean) for k, (c, tot, mean) in prof.items()), key=lambda r: -r[2])
    if args.kernel_table:
        for k, c, ms, mean in table:
            print(f"{k:64s} {c:5.1f}/step {ms:9.4f} ms/step {mean:9.4f} ms/launch", file=sys.stderr)
    # candidates (before the visibility scan): one more march
    cand = F.occgrid_march(rb.origins, rb.directions, grid.binaries, grid._roi, cfg.render_step_size, cfg.near_plane,
                           cfg.far_plane, None, None, cfg.cone_angle, torch.rand(n, device=device), coarse=grid._coarse)
    n_cand = int(cand[0].numel())
    lattice = int(ngp_lattice_steps(o, d, cfg.render_step_size, cfg.cone_angle, cfg.near_plane, cfg.far_plane, cfg.grid_levels).sum())
    # roofline of the dominant PACKED kernel: the occupancy march (count + write launches). Algorithmic bytes per launch pair:
    # one occupancy byte per lattice step and pass, 16 B per emitted sample (ray index, t_start, t_end), 24 B in + 20 B out
    # per ray (origin, direction; count, packed_info row)
    march_ms = sum(mean for k, _, _, mean in table if k.startswith("nsamd_occgrid_march"))
    packed = [(k, ms) for k, _, ms, _ in table if "occgrid" in k or "packed" in k]
    march_bytes = 2 * lattice + 16 * n_cand + 44 * n
    roof = {"bound": "hbm", "achieved": round(march_bytes / (march_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(march_bytes / (march_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
            "kernel": "nsamd_occgrid_march_count + nsamd_occgrid_march_write", "avg_launch_ms": round(march_ms, 4),
            "algorithmic_per_launch": march_bytes, "rocprof_kernel": "nsamd::occgrid_march_kernel<false> + <true>",
            "note": "latency-bound by construction: 1 B of grid per lattice step; lattice steps/s = "
                    f"{2 * lattice / (march_ms * 1e-3):.3e}"}
    top = next(((k, mean) for k, _, _, mean in table if algorithmic_model_ngp(k, kept) is not None), None)
    roof_step = None
    if top is not None:
        bound, work = algorithmic_model_ngp(top[0], kept)
        ach = work / (top[1] * 1e-3) / (1e9 if bound == "hbm" else 1e12)
        peak = HBM_PEAK_GBS if bound == "hbm" else F32_MFMA_PEAK_TFLOPS
        roof_step = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                     "frac": round(ach / peak, 4), "kernel": top[0], "avg_launch_ms": round(top[1], 4),
                     "algorithmic_per_launch": int(work)}
    # grid refresh (excluded from the step, see the docstring): timed once on a copy of the state
    occs0, bin0 = grid.occs.clone(), grid.binaries.clone()
    for timed in (False, True):  # (the first call pays the allocator's first 4 M-point buffers)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        grid.update_every_n_steps(step=512, occ_eval_fn=lambda x: model.field.density_fn(x) * float(cfg.render_step_size))
        torch.cuda.synchronize()
        refresh_ms = (time.perf_counter() - t1) * 1e3
    grid.occs.copy_(occs0)
    grid.binaries.copy_(bin0)
    ms = elapsed / args.steps * 1e3
    out = {
        "metric": "training rays/sec (4096 rays per GPU, instant-ngp packed path)",
        "value": round(RAYS_PER_GPU / (elapsed / args.steps), 1), "unit": "rays/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "instant-ngp 1xMI355X (BASELINE configs[3]): occupancy-grid ray marching (128^3 x 4 levels, random 5 % "
                               "occupied, fixed), packed transmittance scan with early termination + compaction, NerfactoField "
                               "(L=16 hash T=2^19, 64x2 MLP) on the surviving samples, packed compositing, MSE, backward, Adam; "
                               "4096 rays/batch",
                   "rays_per_gpu": RAYS_PER_GPU, "lattice_steps_per_ray": round(lattice / n, 1),
                   "candidate_samples_per_ray": round(n_cand / n, 2), "kept_samples_per_ray": round(kept / n, 2),
                   "field_density": NGP_DENSITY, "render_step_size": cfg.render_step_size, "cone_angle": cfg.cone_angle,
                   "alpha_thre": cfg.alpha_thre, "params": arena.numel, "final_loss": round(float(loss), 6),
                   "grid_refresh_ms": round(refresh_ms, 3),
                   "launch": "eager (module / autograd path)" if runner is None else
                   "explicit kernel schedule over capacity-sized buffers (ngp_step.py): eager launches, two host reads of a sample count per step",
                   "packed_kernels_ms_per_step": {k: round(v, 4) for k, v in packed}},
        "roofline": roof, "roofline_step": roof_step,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_ngp(o, d, cam, tgt, grid.binaries.cpu().numpy().astype(bool), cfg)
    print(json.dumps(out))


def algorithmic_model_ngp(key, kept_samples):
    """Algorithmic work of the field kernels on the packed samples (M = kept samples of the step, SURVEY.md §8d per-sample
    figures)."""
    M = float(kept_samples)
    base = key.split("[")[0]
    if base == "nsamd_hashgrid_encode_fwd" and "L=16" in key:
        import re

        m = re.search(r"M=(\d+)", key)
        return "hbm", float(m.group(1)) * 16 * 8 * 8
    if base in ("nsamd_hashgrid_encode_bwd", "nsamd_hashgrid_encode_bwd_set") and "L=16" in key:
        return "hbm", M * 16 * 8 * 16
    if base == "nsamd_field_mlp_fwd":
        return "mfma", M * 2 * 11392
    if base == "nsamd_field_mlp_bwd":
        return "mfma", M * 2 * 11392 * 2
    return None


def cpu_baseline_ngp(o, d, cam, tgt, binaries, cfg, n_rays=256, steps=2):
    """The packed path's CPU restatement (oracle/packed_oracle.py marcher + visibility, oracle field, packed compositing,
    MSE, backward, torch Adam) on a bounded sample: `n_rays` of the same rays against the same grid (the numpy marcher walks
    rays step by step: ~10 s per 256 rays)."""
    from oracle import nerfacto_oracle as orc
    from oracle import packed_oracle as po

    torch.set_num_threads(min(16, os.cpu_count() or 16))
    ocfg = orc.NerfactoCfg(prop_grids=(), num_images=100, average_init_density=1.0)
    params = orc.init_params(ocfg, seed=0)
    params = {k: v for k, v in params.items() if k.startswith("field.")}
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
                      f"packed compositing, MSE, backward, Adam over the field's parameters; median of {steps} steps after 1 "
                      f"warm-up ({med:.2f} s/step)"}


def dry_run(args, rank, world):
    """CPU-only rehearsal of the multi-GPU launch (the driver's `python -m torch.distributed.run --nproc-per-node N ...
    bench.py --gpus N` line cannot be tried on RCCL before the round ends): same argument / environment handling, a gloo
    process group instead of RCCL, the real model + ParamArena + compact table prefix + PipelinedExchange with the kernel
    segments replaced by rank-dependent synthetic gradients and an SGD update. Checks that every rank ends with identical
    parameters equal to the sequential data-parallel result, then prints the JSON line (value null, "dry_run": true)."""
    from nerfstudio_amd.arena import ParamArena
    from nerfstudio_amd.dp_schedule import PipelinedExchange

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    model = build_model(torch.device("cpu"), seed=rank)  # different init per rank: the broadcast must fix it
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    arena.broadcast_params()
    enc = model.field.mlp_base.encoding
    rows, index = enc.spec.reachable_prefix()
    if args.dp_mode != "sharded":
        arena.register_compact(enc.hash_table, rows, index)
    start = arena.flat.clone()
    lr, steps = 0.5, max(2, args.steps)
    schedule = [k % 3 != 2 for k in range(steps)]
    step = {"k": 0}
    reach = torch.zeros(rows, dtype=torch.bool)
    reach[index] = True
    off_t = next(o for p, o in zip(arena.params, arena.offsets) if p is enc.hash_table)

    def local_grad(r, k):
        """Deterministic per-rank, per-step gradient of the whole arena; zero on the unreachable rows of the prefix."""
        g = torch.full((arena.numel,), float(r + 1) * (k + 1) * 1e-3)
        pref = g[off_t:off_t + 2 * rows].view(rows, 2)
        pref[~reach] = 0.0
        return g

    def run(name):
        k = step["k"]
        if name in (("main", True), ("main", False)):
            a, b = arena.groups["fields"]
            arena.grad[a:b] = local_grad(rank, k)[a:b]
        elif name == "pbwd":
            a, b = arena.groups["proposal_networks"]
            arena.grad[a:b] = local_grad(rank, k)[a:b]
        elif name in ("mopt", "popt"):
            grp = "fields" if name == "mopt" else "proposal_networks"
            a, b = arena.shard_span(grp) if sharded else arena.groups[grp]
            arena.flat[a:b] -= lr * arena.grad[a:b] / world

    sharded = args.dp_mode == "sharded"
    ex = PipelinedExchange(arena, run, sharded=sharded)
    t0 = time.perf_counter()
    for k in range(steps):
        step["k"] = k
        ex.iteration(schedule[k])
    ex.finish()
    elapsed = time.perf_counter() - t0
    expect = start.clone()
    for k in range(steps):
        mean = sum(local_grad(r, k) for r in range(world)) / world
        a, b = arena.groups["fields"]
        expect[a:b] -= lr * mean[a:b]
        if schedule[k]:
            a, b = arena.groups["proposal_networks"]
            expect[a:b] -= lr * mean[a:b]
    err = float((arena.flat - expect).abs().max())
    assert err <= 1e-5, f"rank {rank}: pipelined exchange differs from sequential data-parallel SGD by {err}"
    if world > 1:
        chk = torch.tensor([float(arena.flat.double().sum())], dtype=torch.float64)
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(float(g) == float(gathered[0]) for g in gathered), "ranks ended with different parameters"
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "training rays/sec (4096 rays x 48 samples per GPU)", "value": None, "unit": "rays/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "dry_run": True,
                          "config": {"workload": "launch rehearsal on CPU over gloo: model + arena + compact prefix + pipelined "
                                                 "exchange, synthetic gradients", "params": arena.numel,
                                     "dp_mode": args.dp_mode, "ranks": dist.get_world_size() if world > 1 else 1,
                                     "compact_rows": int(index.numel()), "prefix_rows": int(rows),
                                     "exchange_s_per_step": round(elapsed / steps, 4), "max_abs_error": err}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--autograd", action="store_true",
                    help="drive the step through the nn.Module / autograd API instead of the explicit kernel schedule")
    ap.add_argument("--fused-model-api", action="store_true",
                    help="with the nn.Module driver (implies --autograd): config.fused_train_step — the Model API "
                         "(get_outputs / get_loss_dict / loss.backward()) with the explicit kernel schedule underneath")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL on ROCm); gloo only for functional tests")
    ap.add_argument("--share-gpu", action="store_true", help="functional test: every rank uses cuda:0 (needs gloo)")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--dp-graph", action="store_true", help="N > 1: replay captured hipGraph segments instead of eager launches")
    ap.add_argument("--ngp-module-path", action="store_true",
                    help="--workload ngp through the nn.Module / autograd classes instead of the explicit schedule (ngp_step.py)")
    ap.add_argument("--workload", choices=["bounded", "unbounded", "ngp"], default="bounded",
                    help="bounded = BASELINE configs[1]/[2] (the metric's configuration); unbounded = configs[4] "
                         "(cameras outside the box, most samples in the contracted region); ngp = configs[3] (instant-ngp: "
                         "occupancy-grid marching + early termination, packed samples; N = 1 only)")
    ap.add_argument("--start-step", type=int, default=0,
                    help="not the headline: start the step counter (proposal update schedule, anneal, learning rate) at this "
                         "training step — e.g. 5000 = the steady state of the schedule, proposal networks updated every 6th "
                         "iteration (models/nerfacto.py:208-213) instead of every 2nd as in the first 1000; parameters are "
                         "still at their initial state")
    ap.add_argument("--fixed-batch", action="store_true", help="train on one fixed ray batch instead of rotating the pool")
    ap.add_argument("--force-dp", action="store_true",
                    help="N = 1 only: run the data-parallel schedule (pipelined exchange, compact table prefix, async "
                         "all-reduce on the communication stream) over a ONE-rank RCCL communicator — exercises the N > 1 "
                         "code path on a single-GPU box; the losses must equal the plain N = 1 run")
    ap.add_argument("--dp-mode", choices=["allreduce", "sharded"], default="allreduce",
                    help="N > 1 gradient exchange: all-reduce + replicated Adam (default, the reference's DDP semantics) or "
                         "reduce-scatter -> Adam on the rank's 1/N arena shard -> all-gather (same parameters, 1/N of the "
                         "optimiser's HBM traffic)")
    ap.add_argument("--param-checksum", action="store_true",
                    help="add sha256 digests of the parameter arena and both Adam moments to config (bit-equality checks "
                         "between schedule variants across processes)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: check the launch plumbing (RANK / WORLD_SIZE / MASTER_* env, process group, the pipelined "
                         "exchange with the compact table prefix over gloo) and print the JSON skeleton")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.force_dp:
        assert world == 1, "--force-dp is the single-GPU rehearsal of the data-parallel path"
        os.environ["NSAMD_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=args.dist_backend)

    from nerfstudio_amd import _native
    from nerfstudio_amd import functional as F
    from nerfstudio_amd.arena import ParamArena

    _native.load()  # fail loudly if the HIP extension is missing
    F.DIRECT_GRAD = True  # backward kernels accumulate straight into the arena's gradient views
    if args.workload == "ngp":
        assert world == 1 and not args.force_dp, "--workload ngp is a single-GPU line"
        return run_ngp(args, device)
    model = build_model(device, seed=0)  # same init on every rank (replicated model)
    if args.fused_model_api:
        args.autograd = True
        model.config.fused_train_step = True
    # the reference's optimiser groups (models/nerfacto.py:255-260), AdamOptimizerConfig(lr=1e-2, eps=1e-15) each
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    arena.broadcast_params()
    same = os.environ.get("NSAMD_BENCH_SAME_RAYS") == "1"  # functional check: N ranks, identical rays == the N=1 run
    # each rank its own rays (scripts/train.py:98): BATCH_SLOTS batches per rank, disjoint seeds
    rb, batch, pool = synthetic_batch(device, seed=1000 + (0 if same else 100 * rank), workload=args.workload)
    trainer = Trainer(model, arena, rb, batch, world=world, use_graph=not args.no_graph, use_runner=not args.autograd,
                      pool=None if args.fixed_batch else pool, force_dp=args.force_dp, dp_mode=args.dp_mode)

    if args.start_step:
        trainer.step = args.start_step
        model.proposal_sampler._step = args.start_step
    for _ in range(max(1, args.warmup // 2)):  # eager warm-up: lazy kernel attributes, caches, allocator
        trainer.train_iteration()
    trainer.finish()
    # N > 1 launches the segments eagerly by default: at this kernel granularity the host keeps ahead of the GPU either
    # way (N = 1: 1.00 ms/step eager vs 1.03 ms replayed), and replaying captured segments between eager collectives could
    # only be exercised over gloo with two ranks sharing one GPU, where it is pathologically slow (profiles/
    # r01_dp_schedule_check.log). --dp-graph opts in.
    graphed = trainer.try_capture() if ((world == 1 and not args.force_dp) or args.dp_graph) else False
    for _ in range(args.warmup - max(1, args.warmup // 2)):
        trainer.train_iteration()
    trainer.finish()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_iteration()
    trainer.finish()  # N > 1: the last step's pending main-field update is part of the timed work
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = trainer.last_loss()
    assert bool(torch.isfinite(loss)), "training diverged"
    checksum = None
    if args.param_checksum and rank == 0:  # state right after the timed region (before the profiling iterations)
        import hashlib

        checksum = {name: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:32]
                    for name, t in (("params", arena.flat), ("exp_avg", arena.exp_avg), ("exp_avg_sq", arena.exp_avg_sq))}
    if getattr(trainer, "_seg_times", None) and rank == 0:
        for k, v in trainer._seg_times.items():
            print(f"[dp-timing] {k:16s} n={len(v):4d} median {sorted(v)[len(v) // 2]:9.3f} ms  max {max(v):9.3f} ms", file=sys.stderr)

    roof, table = (None, [])
    if rank == 0:
        roof, table = measure_roofline(trainer, arena, max(1, args.profile_steps))
    elif world > 1:  # keep the collective pattern identical on every rank during the profiling steps
        for _ in range(max(1, args.profile_steps)):
            trainer.train_iteration()
        trainer.finish()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "training rays/sec (4096 rays x 48 samples per GPU)",
            "value": round(world * RAYS_PER_GPU / (elapsed / args.steps), 1),
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("nerfacto 1xMI355X" if world == 1 else f"nerfacto {world}xMI355X data-parallel") +
                                   ": L=16 hash (T=2^19, F=2), 64x2 MLP, 48 samples/ray, 4096 rays/batch per GPU " +
                                   ("(BASELINE configs[1]/[2])" if args.workload == "bounded" else
                                    "(BASELINE configs[4]: unbounded scene, cameras at radius ~3, L-inf contraction)") +
                                   "; full training step incl. proposal nets 256->96, losses, Adam; "
                                   f"{1 if args.fixed_batch else BATCH_SLOTS} ray batches resident in HBM, rotated per step",
                       "rays": args.workload,
                       "rays_per_gpu": RAYS_PER_GPU, "global_rays": world * RAYS_PER_GPU,
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
                                  "autograd modules") if args.autograd else "explicit kernel schedule (train_step.py)"},
            "roofline": roof,
        }
        if args.start_step:
            out["config"]["start_step"] = args.start_step
        if checksum is not None:
            out["config"]["param_checksum"] = checksum
        if world > 1 or args.force_dp:  # what the process group itself reports (not the --gpus argument)
            out["config"]["dp_mode"] = args.dp_mode
            out["config"]["rccl_ranks"] = dist.get_world_size()
            out["config"]["dist_backend"] = dist.get_backend()
        if args.force_dp:
            out["config"]["force_dp"] = "data-parallel schedule over a one-rank RCCL communicator (rehearsal of the N > 1 path)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(workload=args.workload)
        if args.kernel_table:
            for r in table:
                print(f"{r['kernel']:64s} {r['calls_per_step']:5.1f}/step {r['ms_per_step']:9.4f} ms/step "
                      f"{r['mean_ms']:9.4f} ms/launch", file=sys.stderr)
        print(json.dumps(out))
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
