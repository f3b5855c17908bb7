"""Data-parallel iteration schedule for the nerfacto hot path (reference seam: DistributedDataParallel around the model,
pipelines/base_pipeline.py:281 + Optimizers, engine/optimizers.py:74-193).

The gradient arena is laid out by optimiser group (arena.ParamArena), so each group's gradients are one contiguous slice.
The 67 MB main-field slice dominates the exchange; the proposal networks' forward of the NEXT step reads only
proposal-network parameters, so the main-field all-reduce (RCCL, its own stream) can stay in flight across the step
boundary:

    step k:   [proposal fwd k] -> wait AR_main(k-1), [Adam main k-1] -> [main fwd + losses + main bwd k]
              -> AR_main(k) async -> (update steps) [proposal bwd k] -> AR_props(k) -> [Adam props k]

Every parameter is updated before its next use, i.e. exactly the sequential semantics; only the order in which
independent work is issued changes. `finish()` drains the pending update. The class only sequences: the segments
themselves (kernel launches or replays of captured hipGraphs) are supplied by the caller.

`sharded=True` (SURVEY.md §8e "direct RS+AG", ZeRO-1): the all-reduce of a group becomes a REDUCE-SCATTER into the rank's
1/N shard of the gradient slice, the caller's "mopt" / "popt" segments run Adam on that shard only
(arena.ParamArena.step_shard: 1/N of the optimiser's 544 MB of HBM traffic per step and rank), and an ALL-GATHER of the
updated parameter shards follows the update:

    step k:   [proposal fwd k] -> wait RS_main(k-1), [Adam main shard k-1], AG_main -> [main fwd + losses + main bwd k]
              -> RS_main(k) async -> (update steps) [proposal bwd k] -> RS_props(k) -> [Adam props shard k] -> AG_props

Adam is elementwise, so every rank ends each step with the parameters replicated Adam gives (bit for bit when the
reduce-scatter sums in the all-reduce's order: tests/test_distributed_cpu.py, gloo, world 2 and 4).
"""
from __future__ import annotations

from typing import Callable, Optional

from . import _native as N

SEGMENTS = ("pfwd", ("main", True), ("main", False), "pbwd", "mopt", "popt")


class PipelinedExchange:
    """Sequences one training iteration of the replicated model on every rank.

    `run(name)` executes a segment: "pfwd" proposal forward, ("main", updated) zero the main-group gradients + main
    forward + losses + main backward, "pbwd" zero the proposal-group gradients + proposal backward, "mopt" / "popt" the
    Adam update of the "fields" / "proposal_networks" group (gradient scale 1 / world). `before_main_update()` is
    called right before a pending main update is applied (the caller refreshes step-dependent optimiser scalars)."""

    def __init__(self, arena, run: Callable[[object], None], main_group: str = "fields",
                 proposal_group: str = "proposal_networks", before_main_update: Optional[Callable[[], None]] = None,
                 sharded: bool = False, update_stream=None) -> None:
        self.arena, self.run = arena, run
        self.main_group, self.proposal_group = main_group, proposal_group
        self.before_main_update = before_main_update
        self.sharded = sharded
        # Optional second device stream for the pending main-field update: it alone waits for the exchange, so the
        # proposal forward of the next step starts at once on the caller's stream and the 470 MB Adam pass runs BESIDE
        # it as soon as the gradients have arrived (the N = 1 schedule's "deferred Adam" — bench.py — carried over to
        # N > 1). None (CPU tests, gloo): the update follows the proposal forward on the caller's stream.
        self.update_stream = update_stream
        self._handle = None
        self.pending = False  # a main-field all-reduce is in flight and its Adam update has not been applied

    def _reduce(self, group: str):
        """Start the gradient exchange of one optimiser group; -> handle (None: nothing in flight)."""
        if self.sharded:
            return self.arena.reduce_scatter_group(group, async_op=True)
        return self.arena.all_reduce_group(group, async_op=True)

    def _update(self, group: str, segment: str) -> None:
        """The group's optimiser segment (on the rank's shard when sharded, then the parameter all-gather)."""
        self.run(segment)
        if self.sharded:
            self.arena.all_gather_group(group)

    def _finish_main(self) -> None:
        if self.pending:
            if self._handle is not None:
                self._handle.wait()
            self._update(self.main_group, "mopt")
            self._handle, self.pending = None, False

    def iteration(self, updated: bool) -> None:
        if self.update_stream is not None and self.pending:
            import torch

            cur = N.current_stream()
            self.update_stream.wait_stream(cur)  # the step-dependent optimiser scalars were pushed on `cur`
            with N.on_stream(self.update_stream):
                self._finish_main()  # wait() parks THIS stream until the exchange is done, then the update runs on it
            self.run("pfwd")      # meanwhile, on the caller's stream: reads only proposal-network parameters
            cur.wait_stream(self.update_stream)
        else:
            self.run("pfwd")      # overlaps the all-reduce of the previous step's main-field gradients
            self._finish_main()
        self.run(("main", updated))
        self._handle = self._reduce(self.main_group)
        self.pending = True
        if updated:
            self.run("pbwd")      # ... and so does this
            h = self._reduce(self.proposal_group)
            if h is not None:
                h.wait()
            self._update(self.proposal_group, "popt")

    def finish(self) -> None:
        """Drain the pipeline: afterwards every parameter reflects every step taken."""
        if self.pending:
            if self.before_main_update is not None:
                self.before_main_update()
            self._finish_main()


def rehearse_on_cpu(build_model, steps, warmup, dp_mode, rank, world):
    """CPU-only rehearsal of the multi-GPU launch (the driver's `python -m torch.distributed.run --nproc-per-node N ...
    bench.py --gpus N` line cannot be tried on RCCL before the round ends): same argument / environment handling, a gloo
    process group instead of RCCL, the real model + ParamArena + compact table prefix + PipelinedExchange with the kernel
    segments replaced by rank-dependent synthetic gradients and an SGD update. Checks that every rank ends with identical
    parameters equal to the sequential data-parallel result, then prints the JSON line (value null, "dry_run": true)."""
    import json
    import os
    import time

    import torch
    import torch.distributed as dist

    from .arena import ParamArena

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    model = build_model(torch.device("cpu"), seed=rank)  # different init per rank: the broadcast must fix it
    arena = ParamArena(model.get_param_groups_ordered(), lr=1e-2, eps=1e-15)
    arena.broadcast_params()
    enc = model.field.mlp_base.encoding
    rows, index = enc.spec.reachable_prefix()
    if dp_mode != "sharded":
        arena.register_compact(enc.hash_table, rows, index)
    start = arena.flat.clone()
    requested, lr, steps = steps, 0.5, max(2, steps)
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

    sharded = dp_mode == "sharded"
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
                          "n_gpus": world, "steps": requested, "warmup": warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "dry_run": True,
                          "config": {"workload": "launch rehearsal on CPU over gloo: model + arena + compact prefix + pipelined "
                                                 "exchange, synthetic gradients", "params": arena.numel,
                                     "dp_mode": dp_mode, "ranks": dist.get_world_size() if world > 1 else 1,
                                     "compact_rows": int(index.numel()), "prefix_rows": int(rows),
                                     "exchange_s_per_step": round(elapsed / steps, 4), "max_abs_error": err}}))
    if world > 1:
        dist.destroy_process_group()
