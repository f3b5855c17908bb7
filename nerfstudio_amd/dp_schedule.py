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

            cur = torch.cuda.current_stream()
            self.update_stream.wait_stream(cur)  # the step-dependent optimiser scalars were pushed on `cur`
            with torch.cuda.stream(self.update_stream):
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
