"""The nerfacto training iteration as the product runs it fastest: the explicit kernel schedule (train_step.py) captured in
hipGraphs, the main-field Adam deferred beside the next proposal forward, all step-dependent scalars in device memory, and
— for more than one rank — the pipelined gradient exchange of dp_schedule.py.

Reference seam: `Trainer.train_iteration` (engine/trainer.py:487-531) -> `VanillaPipeline.get_train_loss_dict`
(pipelines/base_pipeline.py:290-303) -> `Optimizers` (engine/optimizers.py:74-193) behind `DistributedDataParallel`
(base_pipeline.py:279-282). Two callers drive this class:
  * `pipeline.HipPipeline` — the `nerfacto-hip` method's pipeline: the reference's own trainer calls
    `get_train_loss_dict(step)`, which hands the datamanager's batch to `set_batch` and runs `train_iteration`;
  * `bench.py` — the same object over a pool of synthetic ray batches resident in HBM.

Eager mode runs the Python body every step. Graph mode captures that same body ONCE per schedule variant (proposal networks
updated this step / not, ray_samplers.py:590) into a hipGraph and replays it: ~60 kernel launches become one graph launch,
which is what a sub-millisecond step needs (MI355X_MICROARCH.md price list: eager goes host-bound below ~3 us per kernel).
Everything that changes from step to step lives in device memory: the ray batch, the jitter draws (graph-safe Philox), the
anneal exponent and Adam's bias-corrected step sizes (`hyper`, refreshed by a 32-byte async copy from a ring of pinned host
slots before each replay).

N = 1 with graphs (default): the main-field Adam of iteration k is the first node of iteration k+1's graph, on a branch
beside select-batch / jitter / the proposal forward (`_deferred_iteration_body`; four captured variants: proposal update x
pending Adam). Same dependencies as Adam at the end of the iteration, hence the same bits; `finish()` runs the last pending
update (a caller that reads the parameters — evaluation, checkpoint — calls it first).

N > 1 (data parallel): the iteration runs as segments (eager launches by default, captured hipGraphs on request) and the
main-field gradient exchange (RCCL, its own stream) is PIPELINED across steps (dp_schedule.PipelinedExchange). The proposal
forward of step k+1 reads only proposal-network parameters, so
    step k:   [proposal fwd k] -> (wait X_main k-1) [Adam main k-1] -> [main fwd + losses + main bwd k]
              -> X_main k (async) -> [proposal bwd k] -> X_props k -> [Adam props k]      (last two: update steps)
hides the exchange behind the proposal backward of step k AND the proposal forward of step k+1, with exactly the sequential
semantics (every parameter is updated before its next use).

Camera optimiser on (models/nerfacto.py:131, the reference's nerfacto default): the [num_cameras, 6] exponential map and its
autograd backward are ~100 tiny torch kernels. N = 1: they are captured with everything else (a whole-iteration capture holds
forward, autograd backward and optimiser alike) — launched eagerly around the replay they made the step host-bound (1.80 ms
against 0.76, profiles/r04_camera_optimizer.txt). N > 1 (eager segments) or NSAMD_CAMERAS_OUTSIDE=1: batch selection and pose
corrections before the segments, the rays' share of `pose_adjustment.grad`, the group's exchange and Adam after them; the
kernels read the corrected rays and leave dL/d(origins, directions) per ray either way.
"""
from __future__ import annotations

import os
import sys
import time
from typing import Callable, Dict, Optional

import torch

from . import _native as N

BATCH_SLOTS = 8  # default number of pre-generated ray batches of a pool (bench.py)

_HYPER = {"fields": 0, "proposal_networks": 2, "camera_opt": 6}  # offsets of (step size, 1/sqrt(bc2)) per optimiser group
_HYPER_ANNEAL, _HYPER_SLOT, _HYPER_FLOATS = 4, 5, 8


class HipTrainer:
    """model: nerfacto.NerfactoModel or the plugin's HipNerfactoModel; arena: arena.ParamArena over its optimiser groups
    ("fields", "proposal_networks"[, "camera_opt"]).

    pool: {"origins" [slots,n,3], "directions", "cameras" [slots,n], "target" [slots,n,3]} resident in HBM — iteration i
    trains on slot i % slots, selected on the device (replayable); None: the caller fills the runner's static buffers
    (`set_batch`) before every iteration.
    lr_source(group, iteration) -> learning rate of that iteration (default: the nerfacto recipe's schedulers over arena.lr).
    drive_callbacks: call the model's BEFORE/AFTER_TRAIN_ITERATION callbacks here (False when a trainer does: HipPipeline).
    runner: a train_step.NerfactoTrainStep stand-in (CPU tests of the schedule's host logic)."""

    def __init__(self, model, arena, ray_bundle, batch, world: int = 1, use_graph: bool = True, use_runner: bool = True,
                 pool=None, force_dp: bool = False, dp_mode: str = "allreduce",
                 lr_source: Optional[Callable[[str, int], float]] = None, drive_callbacks: bool = True, runner=None) -> None:
        self.model, self.arena, self.rb, self.batch, self.world = model, arena, ray_bundle, batch, world
        self.dp = world > 1 or force_dp  # force_dp: the data-parallel schedule with a one-rank communicator
        # "sharded": reduce-scatter -> Adam on the rank's 1/N arena shard -> all-gather (dp_schedule.py); "allreduce": the
        # replicated optimiser behind an all-reduce (the reference's DDP semantics, and the default)
        self.dp_sharded = self.dp and dp_mode == "sharded"
        self.dp_fork = False  # set below: proposal backward chains beside the main chain in the data-parallel schedule
        self.pool = pool
        self.slots = int(pool["origins"].shape[0]) if pool is not None else 1
        self.step = 0
        self.opt_step = 0
        self.drive_callbacks = drive_callbacks
        self._true_steps = dict(arena.step_counts)
        dev = ray_bundle.origins.device
        self.on_gpu = dev.type == "cuda"
        # device-resident step-dependent scalars: Adam (step size, 1/sqrt(bc2)) per optimiser group, the anneal exponent,
        # the batch slot of this step
        self.hyper = torch.zeros(_HYPER_FLOATS, device=dev)
        # The host runs ahead of the GPU, so the pinned source of an async copy must not be rewritten before the copy
        # has executed: a ring of slots, each guarded by the event recorded after its last copy.
        self.hyper_ring = [torch.zeros(_HYPER_FLOATS).pin_memory() if self.on_gpu else torch.zeros(_HYPER_FLOATS) for _ in range(64)]
        # (numpy views of the pinned slots: a scalar store into a tensor costs ~5 us of dispatch, eight of them per iteration
        #  were a tenth of the host's share of a step through the pipeline seam)
        self.hyper_ring_np = [h.numpy() for h in self.hyper_ring]
        self.hyper_events = [None] * 64
        self.hyper_slot = 0
        lr_source_is_default = lr_source is None
        if lr_source is None:
            from .schedulers import ExponentialDecayScheduler, ExponentialDecaySchedulerConfig, nerfacto_schedulers

            sched = nerfacto_schedulers()
            # method_configs.py:117-120: camera_opt = Adam(lr 1e-3, eps 1e-15) + ExponentialDecay(lr_final 1e-4, 5000 steps)
            sched["camera_opt"] = ExponentialDecayScheduler(ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=5000))
            base = {"fields": arena.lr, "proposal_networks": arena.lr, "camera_opt": 1e-3}
            lr_source = lambda group, it: sched[group].get_lr(max(it, 0), base[group])  # noqa: E731
        self.lr_source = lr_source
        self.exchange = None  # dp_schedule.PipelinedExchange (N > 1 with the runner)
        # ---- device-side head of the iteration (nsamd_step_prologue), N = 1 with the runner on the GPU ----
        # A replayed graph had two host-issued operations in front of it every iteration — the 32-byte upload of `hyper` and the
        # offset fill of torch's graph-safe generator (for the jitter's uniform_) — and each eager -> graph hand-over leaves the
        # stream idle for ~8 us (profiles/r05_s9_seam_trace_gaps.txt). With the prologue the first node of the graph fetches the
        # step's scalars itself and draws the step's uniforms with a counter-based generator. NSAMD_STEP_PROLOGUE selects where
        # the scalars come from:
        #   "ring" (default)  a ring of rows in pinned HOST memory the device reads directly: the host writes row i % rows for
        #                     iteration i (exact host arithmetic, at the time it would have issued the upload) and launches; the
        #                     device's own row counter picks it up. Nothing is predicted, so it serves a trainer whose learning
        #                     rates come from outside (pipeline.TrainEngine) as well. An event every 64 rows keeps the host from
        #                     lapping the device.
        #   "table"           a table of rows in DEVICE memory the host computes AHEAD by running its own bookkeeping forward
        #                     (re-uploaded when the rows run out or a row differs from what the host computes for the iteration at
        #                     hand); needs iterations whose scalars are a function of the step (this trainer's own schedulers).
        #   "0"               the per-iteration upload and torch's generator (A/B; other draws, same distribution).
        self.prologue = False          # set below
        self.prologue_table = False    # the scalars come from the predicted table
        self.prologue_ring = False     # the scalars come from the ring in host memory
        self.table_rows = 128
        self.ring_rows = 256
        self.hyper_views = {g: self.hyper[o:o + 2] for g, o in _HYPER.items()}
        self.loss_buf = torch.zeros((), device=dev)
        model.proposal_sampler.anneal_dev = self.hyper[_HYPER_ANNEAL:_HYPER_ANNEAL + 1]
        self.graphs = None
        self.use_graph = use_graph and self.on_gpu
        self.runner = runner
        self.defer = False
        self.defer_scatter = False
        self.opt_parallel = True  # False: the deferred Adam runs on the main stream (per-kernel timing)
        # False: the jitter buffer of the runner is filled by the caller before every iteration (parity tests inject the
        # draws the CPU oracle uses; the default draws them on the device inside the iteration, graph-safe Philox)
        self.draw_jitter = True
        self._pending_main = False  # deferred schedule: the main-field Adam of the previous iteration is still to run
        self.cam_group = "camera_opt" if "camera_opt" in arena.groups else None
        self.cam_inside = False  # the camera optimiser's torch ops and Adam are part of the (captured) iteration body
        if use_runner or runner is not None:  # explicit kernel schedule over static buffers (train_step.py); default
            if self.runner is None:
                from .train_step import NerfactoTrainStep

                self.runner = NerfactoTrainStep(model, ray_bundle.origins.shape[0], dev)
            r = self.runner
            r.grad_lookup = arena.grad_lookup()
            if hasattr(r, "prop_gates"):
                r.gates_precleared = True  # `_zero` clears the proposal levels' gradient flags with the gradients
            r.set_batch(ray_bundle.origins, ray_bundle.directions, ray_bundle.camera_indices, batch["image"])
            r.anneal_dev = self.hyper[_HYPER_ANNEAL:_HYPER_ANNEAL + 1]
            cam_on = getattr(r, "cam_opt", None) is not None
            self.cam_inside = cam_on and self.cam_group is not None and not self.dp and os.environ.get("NSAMD_CAMERAS_OUTSIDE", "0") != "1"
            r.cameras_outside = cam_on and not self.cam_inside  # see the module docstring
            if os.environ.get("NSAMD_SIDE_STREAM", "1") == "0":  # A/B switch: proposal backward on the main stream
                r.side_stream = None
            # N = 1: the main-field Adam of iteration k (470 MB of HBM streaming) runs BESIDE the proposal forward of
            # iteration k+1 (L2-resident gathers and per-ray scans that read only proposal-network parameters) — the
            # single-GPU form of the pipelined schedule above; same dependencies, same bits. Measured on three MI355X boxes
            # (profiles/r02_schedule_ab.txt): 1.3 / 3 / 4.5 % faster than Adam at the end of the iteration when replayed
            # from hipGraphs, neutral with eager launches — so it is the default with graphs. NSAMD_DEFER_MAIN_ADAM=0/1: A/B.
            self.defer = (not self.dp and self.on_gpu and
                          os.environ.get("NSAMD_DEFER_MAIN_ADAM", "1" if self.use_graph else "0") == "1")
            # NSAMD_DEFER_SCATTER=1 (opt-in, measured and NOT adopted: profiles/r03_negative_results.txt item 8) defers the
            # main TABLE SCATTER of iteration k as well, beside [select batch, proposal forward k+1]; same bits, 1-2 % slower.
            self.defer_scatter = self.defer and os.environ.get("NSAMD_DEFER_SCATTER", "0") == "1"
            self.fork_after_bins = os.environ.get("NSAMD_FORK_AFTER_BINS", "0") == "1"
            if self.defer:
                self.opt_stream = torch.cuda.Stream(device=dev)
                self._opt_fork, self._opt_join = torch.cuda.Event(), torch.cuda.Event()
                self._sh_fork, self._sh_join = torch.cuda.Event(), torch.cuda.Event()
                self._batch_ready = torch.cuda.Event()
            self.terms_on_branch = os.environ.get("NSAMD_TERMS_ON_BRANCH", "1") == "1"  # (=0: in line, A/B)
            mode = os.environ.get("NSAMD_STEP_PROLOGUE", "ring")
            mode = "ring" if mode == "1" else mode
            if (mode in ("ring", "table") and self.on_gpu and runner is None
                    and getattr(r, "single_jitter", False) and hasattr(r, "jitter")):
                self.prologue = True
                self.step_counter = torch.zeros(2, device=dev, dtype=torch.int64)  # [row, draw]
                # The draws are keyed by (seed, draw counter). The counter starts at the model's training step, so that a trainer
                # built in the middle of a run (a resumed checkpoint, an engine rebuilt for another batch size) does not replay
                # the jitter and background draws of steps 0, 1, ...; the rank is mixed into the seed, so that data-parallel
                # ranks that share torch's seed still draw different numbers (the reference: one generator per process).
                self.step_counter[1] = int(getattr(model, "step", 0) or 0)
                rank = 0
                if world > 1 and os.environ.get("NSAMD_BENCH_SAME_RAYS") != "1":  # (the functional check: every rank the same rays AND draws)
                    import torch.distributed as dist

                    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
                self.rng_seed = ((int(torch.initial_seed()) + 0x632BE59BD9B4E019 * rank) * 0x9E3779B97F4A7C15
                                 + 0x5851F42D4C957F2D) & 0xFFFFFFFFFFFFFFFF
                # the batch slot must be read INSIDE the iteration body: with the camera parts outside the graph the batch is
                # selected eagerly ahead of the replay, i.e. before the prologue would have written the slot (then: the upload
                # for the scalars, the prologue for the draws only)
                # (the data-parallel segments keep the upload as well — their Adam launches are segments of their own, ordered by
                #  the exchange — and take the DRAWS from the prologue: one generator for every schedule, so that a one-rank
                #  data-parallel run trains through the bits of the single-GPU one)
                inside = not r.cameras_outside and not self.dp
                if mode == "ring" and inside:
                    self.prologue_ring = True
                    self.ring_host = torch.zeros(self.ring_rows, _HYPER_FLOATS).pin_memory()  # (device-visible: hipHostMalloc)
                    self.ring_np = self.ring_host.numpy()
                    self._ring_pos = 0                 # rows written so far == the device's row counter at the next launch
                    self._ring_events = [None] * 4     # recorded every 64 rows
                elif mode == "table" and inside and bool(drive_callbacks) and lr_source_is_default:
                    # (this trainer drives the model's callbacks itself and the learning rates are a function of the iteration)
                    self.prologue_table = True
                    self.hyper_table = torch.zeros(self.table_rows * _HYPER_FLOATS, device=dev)
                    self.table_host = torch.zeros(self.table_rows, _HYPER_FLOATS).pin_memory()
                    self.table_host_np = self.table_host.numpy()
                    self._table_pos, self._table_valid, self._table_event, self._table_base = 0, False, None, 0
                    import numpy as np

                    self._row_scratch = np.zeros(_HYPER_FLOATS, dtype=np.float32)
                    self._row_unread = np.zeros(_HYPER_FLOATS, dtype=bool)  # (entries an iteration without a pending update skips)
                    self._row_unread[_HYPER["fields"]:_HYPER["fields"] + 2] = True
            if self.dp:
                from .dp_schedule import PipelinedExchange

                # the pending main-field Adam waits for its exchange on its own stream, beside the next proposal forward
                # (NSAMD_DP_UPDATE_STREAM=1; measured on a one-rank communicator, profiles/r03_dp_rehearsal.txt: 0.992 vs
                # 0.969 ms — off by default)
                upd = torch.cuda.Stream(device=dev) if (self.on_gpu and os.environ.get("NSAMD_DP_UPDATE_STREAM", "0") == "1") else None
                self.exchange = PipelinedExchange(arena, self._run, before_main_update=self._push_hyper,
                                                  sharded=self.dp_sharded, update_stream=upd)
                # eager segments only (a captured segment must end with its streams joined); NSAMD_DP_FORK=0: round-2 order
                self.dp_fork = os.environ.get("NSAMD_DP_FORK", "1") == "1" and getattr(r, "side_stream", None) is not None
                # the coarse levels of the main table can only ever touch 288 k of their 2.6 M rows: exchange those
                # compactly (2.3 MB instead of 21 MB of the 67 MB main-field all-reduce)
                enc = model.field.mlp_base.encoding
                if hasattr(enc, "spec"):
                    rows, index = enc.spec.reachable_prefix()
                    if index.numel() and index.numel() < rows // 2 and not self.dp_sharded:
                        arena.register_compact(enc.hash_table, rows, index)  # (the reduce-scatter takes the slice as it lies)

    # -- the batch ---------------------------------------------------------------------------------------------------
    def set_batch(self, ray_bundle, batch) -> None:
        """The next iteration's rays and targets (a trainer's `datamanager.next_train`, base_datamanager.py:506-515), copied
        into the static buffers the captured graphs read. Stream-ordered: no host synchronisation."""
        assert self.pool is None, "this trainer rotates its own pool of batches"
        o = ray_bundle.origins.reshape(-1, 3)
        if self.runner is not None:
            assert o.shape[0] == self.runner.n, "the captured schedule is built for a fixed number of rays per batch"
            self.runner.set_batch(o, ray_bundle.directions.reshape(-1, 3), ray_bundle.camera_indices.reshape(-1), batch["image"])
        else:
            self.rb, self.batch = ray_bundle, batch

    # -- pieces of one iteration ---------------------------------------------------------------------------------------
    def _prologue(self, updated):
        if self.drive_callbacks:
            self.model.set_step(self.step)  # BEFORE_TRAIN_ITERATION callback: proposal weight anneal
        self._push_hyper()

    def _hyper_row(self, out, step, counts, have_pending):
        """The step-dependent scalars of iteration `step` into `out` (8 floats, numpy): Adam step sizes of the NEXT update of
        each group (`counts`: the groups' step counters before the iteration) + the anneal exponent + the batch slot."""
        from . import functional as F

        a = self.arena
        # iteration i runs with lr(i); a pending (pipelined) main-field update belongs to the previous iteration
        it_fields = step - 1 if have_pending else step
        out[:] = 0.0
        for group, off in _HYPER.items():
            if group not in a.groups:
                continue
            lr = self.lr_source(group, max(it_fields, 0) if group == "fields" else step)
            out[off], out[off + 1] = F.adam_hyper(counts[group] + 1, lr, a.betas)
        out[_HYPER_ANNEAL] = self.model.proposal_sampler._anneal
        out[_HYPER_SLOT] = float(step % self.slots)

    def _predict_rows(self):
        """Rows 1 .. of the table: the scalars of the iterations AFTER the one at hand, by running the host-side bookkeeping of
        those iterations (the model's step callbacks, the sampler's update rule, the optimiser groups' step counters) ahead on
        its own scalar state, which is put back afterwards. Row 0 (the iteration at hand) is already in place."""
        m, ps = self.model, self.model.proposal_sampler
        saved = (ps._step, ps._steps_since_update, ps._anneal, getattr(m, "step", None))
        counts, pending, step = dict(self.arena.step_counts), self._have_pending, self.step
        rows = self.table_host_np
        try:
            for r in range(self.table_rows):
                if r > 0:
                    m.set_step(step)  # BEFORE_TRAIN_ITERATION: the anneal exponent of that iteration
                    self._hyper_row(rows[r], step, counts, pending)
                updated = ps.updated_this_step()
                # what the iteration does to the counters (train_iteration's `stepped`)
                if self.defer:
                    if pending:
                        counts["fields"] += 1
                    pending = True
                else:
                    counts["fields"] += 1
                if updated:
                    counts["proposal_networks"] += 1
                    ps.mark_updated()
                if self.cam_inside:
                    counts[self.cam_group] += 1
                m.after_step(step)  # AFTER_TRAIN_ITERATION: the sampler's step counter
                step += 1
        finally:
            ps._step, ps._steps_since_update, ps._anneal = saved[:3]
            if saved[3] is not None:
                m.step = saved[3]

    def _push_hyper(self, direct: bool = False):
        """Adam step sizes of the NEXT update of each group + the anneal exponent -> device (async, race-free). `direct`: into
        `hyper` itself whatever the mode (a caller that launches an update outside an iteration body: `finish`)."""
        a = self.arena
        if self.prologue_ring and not direct:
            # the graph's first node reads row `counter % rows` out of pinned host memory: write it, launch, done
            i = self._ring_pos
            if i % 64 == 0:
                k = (i // 64) % 4
                ago = self._ring_events[(k + 2) % 4]  # recorded 128 rows ago: the rows about to be rewritten were read long before
                if ago is not None:
                    ago.synchronize()
                ev = torch.cuda.Event()
                ev.record(N.current_stream())
                self._ring_events[k] = ev
            self._hyper_row(self.ring_np[i % self.ring_rows], self.step, a.step_counts, self._have_pending)
            self._ring_pos = i + 1
            return
        if self.prologue_table and not direct:
            # the graph's first node copies row `counter % rows` of the table into `hyper`: nothing to upload while the row the
            # host computes for THIS iteration is the one the device is about to read
            row = self._row_scratch
            self._hyper_row(row, self.step, a.step_counts, self._have_pending)
            pos = self._table_pos
            unread = self.defer and not self._have_pending

            def serves(r):
                # an iteration of the deferred schedule WITHOUT a pending main-field update (the first one, and the one after
                # every `finish`) launches no main-field Adam: its two scalars are not read, so the row predicted for an
                # iteration with a pending update serves it as well — a `finish` does not cost a new table
                want = self.table_host_np[r]
                return bool(((want == row) | self._row_unread).all() if unread else (want == row).all())

            if self._table_valid:
                if pos < self.table_rows and serves(pos):
                    self._table_pos = pos + 1
                    return
                # a caller that rewound the training state (bench.py repeats its window on the same iterations; a checkpoint
                # restore): the rows of those iterations are still in the table — move the device's row counter, nothing else
                back = self.step - self._table_base
                if 0 <= back < self.table_rows and serves(back):
                    self.step_counter[0:1].fill_(back)
                    self._table_pos = back + 1
                    return
            if self._table_event is not None:
                self._table_event.synchronize()  # the previous upload has read the pinned table
            self.table_host_np[0] = row
            self._predict_rows()
            self.hyper_table.copy_(self.table_host.reshape(-1), non_blocking=True)
            self.step_counter[0:1].zero_()
            self._table_event = torch.cuda.Event()
            self._table_event.record(N.current_stream())
            self._table_pos, self._table_valid, self._table_base = 1, True, self.step
            return
        slot = self.hyper_slot
        self.hyper_slot = (slot + 1) % len(self.hyper_ring)
        if self.hyper_events[slot] is not None:
            self.hyper_events[slot].synchronize()  # the copy that last read this slot (64 pushes ago) is done
        h, hn = self.hyper_ring[slot], self.hyper_ring_np[slot]
        self._hyper_row(hn, self.step, a.step_counts, self._have_pending)
        self.hyper.copy_(h, non_blocking=True)
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(N.current_stream())
            self.hyper_events[slot] = ev

    def _step_prologue(self):
        """First launch of an iteration body (captured with it): the step's scalars out of the table, the step's draws."""
        if not self.prologue:
            return
        from . import _native as N

        r = self.runner
        draw = self.draw_jitter
        j = r.jitter if draw else None
        bg = r.bg_rays if (draw and r.bg_rays is not None) else None
        if self.prologue_ring:
            rows_ptr, rows = self.ring_host.data_ptr(), self.ring_rows
        elif self.prologue_table:
            rows_ptr, rows = N.ptr(self.hyper_table), self.table_rows
        else:
            rows_ptr, rows = None, 0
        N.check(N.load().nsamd_step_prologue(
            N.ptr(self.step_counter), rows_ptr, rows, N.ptr(self.hyper), N.ptr(j), j.numel() if j is not None else 0,
            N.ptr(bg), bg.numel() if bg is not None else 0, self.rng_seed, N.stream()), "step_prologue")

    def _fwd_bwd(self, updated):
        """Single-process path: forward, losses and the main backward (runner: also the proposal backward)."""
        from .cameras.rays import RayBundle

        if self.runner is not None:
            self._step_prologue()
            self._select_batch()
            self.runner.apply_camera_corrections()
            # the main table's gradient is written, not accumulated; the proposal group's gradients are neither produced nor
            # consumed on a step that does not update it (ray_samplers.py:590-599), so its 10 MB need no zero-fill then
            self._zero(updated)
            self.runner.forward_proposals(self.draw_jitter and not self.prologue, need_enc=updated)
            self.runner.forward_main_and_losses(updated)
            self.runner.backward_all(updated)  # the backward chains run as parallel branches
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
        loss = sum(loss_dict.values())
        loss.backward()
        self.loss_buf.copy_(loss.detach())

    def _zero(self, updated, groups=None):
        """Zero-fills ahead of an iteration's forward; `groups`: only these gradient slices (a later segment of the same
        iteration: the proposal levels' gradient flags are raised by then and stay as they are)."""
        whole = groups is None
        if groups is None:
            groups = ["fields", "proposal_networks"] if updated else ["fields"]
            if self.cam_inside:
                groups = groups + [self.cam_group]
        self.arena.zero_grad(groups, skip=self.runner.written_params())
        if whole and updated:
            self._clear_gates()

    def _clear_gates(self):
        # (instead of one 4-byte memset node per level ahead of its weights backward; BEFORE the losses launch, which raises
        # the flags when it also runs the levels' weights backward)
        if getattr(self.runner, "gates_precleared", False):
            self.runner.prop_gates.zero_()

    def _deferred_iteration_body(self, updated, pending):
        """One iteration of the deferred schedule (N = 1, runner):
            [Adam main k-1  ||  select batch, proposal forward k] -> main forward, losses, backward chains k
            -> [Adam proposals k]                                                            (update steps)
        Inside a captured hipGraph the two halves of the first line are parallel branches. With the camera optimiser on,
        batch selection and the pose corrections have already run (eagerly, `_cameras_before`)."""
        r, a = self.runner, self.arena
        main = N.current_stream()
        beside = pending and self.opt_parallel
        self._step_prologue()  # the step's scalars and draws: first node, every branch below depends on it
        draw = self.draw_jitter and not self.prologue

        def pending_update():  # what iteration k-1 left behind: [its table scatter ->] its main-field Adam
            if self.defer_scatter:
                r.backward_table(shadow=True)
            a.step(grad_scale=1.0, groups=["fields"], hyper_dev=self.hyper_views)
            if beside:
                # ... and, off the critical path, this iteration's zero-fills: the Adam above was the last reader of the
                # field gradients, the proposal / camera groups were consumed at the end of their last update iteration, and
                # nothing before the join below writes a gradient
                self._zero(updated)

        # The ray terms of this iteration's main-field forward (train_step.ray_terms_launch) need the updated head weights and
        # the selected batch, nothing else: they go on the Adam branch, behind an event the main branch records once the batch
        # is in place — off the critical path instead of a launch (and a dependent-launch gap) in front of the hash forward.
        terms_beside = beside and getattr(r, "ray_terms_on", False) and self.terms_on_branch

        def fork():
            self._opt_fork.record(main)
            self.opt_stream.wait_event(self._opt_fork)
            with N.on_stream(self.opt_stream):
                pending_update()
                if not terms_beside:
                    self._opt_join.record(self.opt_stream)

        def terms_behind_batch():  # (runs inside forward_proposals, right behind the launch that selects the batch)
            self._batch_ready.record(main)
            self.opt_stream.wait_event(self._batch_ready)
            with N.on_stream(self.opt_stream):
                r.ray_terms_launch()
                self._opt_join.record(self.opt_stream)

        # NSAMD_FORK_AFTER_BINS=1: the branch starts BEHIND the launch that selects the batch and writes the initial bins (8 MB of
        # stores that read 45 us beside the HBM-saturating Adam and 14 us alone) instead of in front of it
        late_fork = beside and self.fork_after_bins and not r.cameras_outside and getattr(r, "fuse_select", False) \
            and getattr(r, "cam_opt", None) is None
        terms_beside = terms_beside and not late_fork
        if beside and not late_fork:
            fork()
        elif pending and not beside:
            pending_update()
        if not r.cameras_outside:
            self._select_batch()
            r.apply_camera_corrections()
        if late_fork:
            r.after_bins = fork
        elif terms_beside:
            r.after_bins = terms_behind_batch
        r.forward_proposals(draw, need_enc=updated)
        if beside:
            main.wait_event(self._opt_join)
        if self.defer_scatter:  # the final samples are known: copy what defines them, beside the main forward
            if self.opt_parallel:
                self._sh_fork.record(main)
                self.opt_stream.wait_event(self._sh_fork)
                with N.on_stream(self.opt_stream):
                    r.shadow_points()
                    self._sh_join.record(self.opt_stream)
            else:
                r.shadow_points()
        if not beside:
            self._zero(updated)
        r.forward_main_and_losses(updated)
        r.defer_table = self.defer_scatter
        try:
            r.backward_all(updated)
        finally:
            r.defer_table = False
        if self.defer_scatter and self.opt_parallel:
            main.wait_event(self._sh_join)
        late = (["proposal_networks"] if updated else []) + ([self.cam_group] if self.cam_inside else [])
        if late:
            a.step(grad_scale=1.0, groups=late, hyper_dev=self.hyper_views)

    def _select_batch(self):
        """This step's rays out of the HBM-resident pool (slot index in device memory: replayable) — the hand-over the
        reference's datamanager does each iteration (base_datamanager.py:506-515)."""
        if self.pool is None:
            return
        from . import _native as N

        p = self.pool
        if self.runner is not None:
            r = self.runner
            co = getattr(r, "cam_opt", None) is not None  # the kernels read the pose-corrected copies
            if not co and getattr(r, "fuse_select", False):
                # the runner's next `forward_proposals` selects the batch in the launch that writes the initial bins
                r.pending_select = (N.ptr(self.hyper[_HYPER_SLOT:_HYPER_SLOT + 1]), self.slots, p)
                return
            o, d = (r.raw_origins, r.raw_directions) if co else (r.origins, r.directions)
            c, t = r.camera_indices, r.target
        else:
            o, d, c, t = self.rb.origins, self.rb.directions, self.rb.camera_indices, self.batch["image"]
        N.check(N.load().nsamd_select_batch(N.ptr(self.hyper[_HYPER_SLOT:_HYPER_SLOT + 1]), self.slots, o.shape[0],
                                            N.ptr(p["origins"]), N.ptr(p["directions"]), N.ptr(p["cameras"]), N.ptr(p["target"]),
                                            N.ptr(o), N.ptr(d), N.ptr(c), N.ptr(t), N.stream()), "select_batch")

    def _optimise(self, updated):
        # the reference steps an optimiser group only when it received gradients (engine/optimizers.py:160-172)
        groups = (["fields", "proposal_networks"] if updated else ["fields"]) + ([self.cam_group] if self.cam_inside else [])
        self.arena.step(grad_scale=1.0 / self.world, groups=groups, hyper_dev=self.hyper_views)

    # -- camera optimiser: the host-side halves around the captured part ---------------------------------------------------
    @property
    def _cams_outside(self):
        return self.runner is not None and getattr(self.runner, "cameras_outside", False)

    def _cameras_before(self):
        """Batch selection + pose corrections (cameras/camera_optimizers.py:148-153), eagerly, ahead of the replay."""
        self._select_batch()
        self.runner.apply_camera_corrections()

    def _cameras_after(self, updated):
        """dL/d(origins, directions) per ray -> pose_adjustment.grad (+ the L2 regulariser), the group's exchange and Adam."""
        a, g = self.arena, self.cam_group
        if g is None:
            return
        a.zero_grad([g])
        self.runner.backward_cameras(updated, force=True)
        if self.dp:
            h = a.all_reduce_group(g)
            if h is not None:
                h.wait()
        a.step(grad_scale=1.0 / self.world, groups=[g], hyper_dev=self.hyper_views)

    # -- data-parallel segments (N > 1, runner) --------------------------------------------------------------------------
    @property
    def pipelined(self):
        return self.dp and self.runner is not None

    def _seg(self, name):
        """The body of one captured segment (also what the eager path runs)."""
        r, a = self.runner, self.arena
        if name == "pfwd":
            self._step_prologue()  # (the step's draws; the scalars were uploaded by `_prologue`)
            if not self._cams_outside:
                self._select_batch()
                r.apply_camera_corrections()
            r.forward_proposals(self.draw_jitter and not self.prologue)
        elif name in (("main", True), ("main", False)):
            if name[1] and self.dp_fork:
                # update step, eager launches: the proposal chains start on their side streams here, beside the main chain
                # (as in the N = 1 schedule) — and, since the exchange starts the main-field collective right after this
                # segment, beside that too; "pbwd" only joins them
                self._zero(True)
                r.forward_main_and_losses(True)
                r.backward_fork(True)
            else:
                self._zero(False)
                if name[1]:
                    self._clear_gates()
                r.forward_main_and_losses(name[1])
                r.backward_main(reserve=name[1])
        elif name == "pbwd":
            if self.dp_fork:
                r.backward_join(True)
            else:
                self._zero(True, groups=["proposal_networks"])
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
        """Apply every pending update: afterwards the parameters reflect every iteration taken (callers: the end of a timed
        region, evaluation, checkpointing)."""
        if self.exchange is not None:
            self.exchange.finish()
        if self._pending_main:  # deferred schedule: the last iteration's [table scatter and] main-field update
            self._push_hyper(direct=True)  # (the table's rows stay valid: see `_push_hyper`)
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
        if self._cams_outside:
            self._cameras_before()
        self.exchange.iteration(updated)
        if self._cams_outside:
            self._cameras_after(updated)

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
        side.wait_stream(N.current_stream())
        with N.on_stream(side):
            defer, self.defer = self.defer, False  # (in order, so that every schedule trains through the same states)
            for upd in (True, False):
                self._eager_iteration(upd)
            self.finish()
            self.defer = defer
        N.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def capture(self, warm: bool = True):
        """warm=False: the caller has already run eager iterations of this schedule (kernel attributes, lazily built
        workspaces and the allocator are warm) and must not train twice on one batch (pipeline.HipPipeline)."""
        if warm:
            self.warm_variants()
        else:
            self.finish()
            torch.cuda.synchronize()
        graphs = {}
        if self.pipelined:
            from .dp_schedule import SEGMENTS

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
                    self._plain_body(upd)
                graphs[("all", upd)] = g
        for name in self.arena.step_counts:  # captures executed nothing; undo the host-side counters they bumped
            self.arena.step_counts[name] = self._true_steps[name]
        self.graphs = graphs

    def _plain_body(self, updated):
        """[select batch] -> forward -> losses -> backward -> Adam, in order (what a captured ("all", updated) graph holds)."""
        r = self.runner
        if r is not None and r.cameras_outside:
            self._step_prologue()
            self._zero(updated)
            r.forward_proposals(self.draw_jitter and not self.prologue, need_enc=updated)
            r.forward_main_and_losses(updated)
            r.backward_all(updated)
        else:
            self._fwd_bwd(updated)
        self._optimise(updated)

    def _eager_iteration(self, updated):
        if self.pipelined:
            self._pipelined_iteration(updated)
        elif self.dp:
            self._plain_dp_iteration(updated)
        else:
            self._prologue(updated)
            if self._cams_outside:
                self._cameras_before()
            if self.defer:
                self._deferred_iteration_body(updated, self._pending_main)
                self._pending_main = True
            else:
                self._plain_body(updated)
            if self._cams_outside:
                self._cameras_after(updated)
        self._true_steps = dict(self.arena.step_counts)

    def try_capture(self, warm: bool = True):
        if not self.use_graph or (self.dp and not self.pipelined):
            return False
        try:
            self.capture(warm)
            return True
        except Exception as e:  # noqa: BLE001 - any capture problem degrades to the eager path, never to no result
            print(f"[nerfstudio_amd.trainer] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
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
            if self._cams_outside:
                self._cameras_before()
            if self.defer:
                self.graphs[("all", updated, self._pending_main)].replay()
                stepped = (("fields",) if self._pending_main else ()) + (("proposal_networks",) if updated else ())
                self._pending_main = True
            else:
                self.graphs[("all", updated)].replay()
                stepped = ("fields", "proposal_networks") if updated else ("fields",)
            if self.cam_inside:
                stepped = stepped + (self.cam_group,)
            for name in stepped:
                self.arena.step_counts[name] += 1  # the replayed Adam launches did step these groups
            if self._cams_outside:
                self._cameras_after(updated)
            self._true_steps = dict(self.arena.step_counts)
        self.opt_step += 1
        if updated:
            ps.mark_updated()
        if self.drive_callbacks:
            self.model.after_step(self.step)  # AFTER_TRAIN_ITERATION callback
        self.step += 1
        return self.loss_buf

    def last_loss(self):
        if self.runner is not None:
            return sum(self.runner.loss_dict().values())
        return self.loss_buf

    def loss_dict(self) -> Dict[str, torch.Tensor]:
        return self.runner.loss_dict()
