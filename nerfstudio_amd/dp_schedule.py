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
                 proposal_group: str = "proposal_networks", before_main_update: Optional[Callable[[], None]] = None) -> None:
        self.arena, self.run = arena, run
        self.main_group, self.proposal_group = main_group, proposal_group
        self.before_main_update = before_main_update
        self._handle = None
        self.pending = False  # a main-field all-reduce is in flight and its Adam update has not been applied

    def _finish_main(self) -> None:
        if self.pending:
            if self._handle is not None:
                self._handle.wait()
            self.run("mopt")
            self._handle, self.pending = None, False

    def iteration(self, updated: bool) -> None:
        a = self.arena
        self.run("pfwd")          # overlaps the all-reduce of the previous step's main-field gradients
        self._finish_main()
        self.run(("main", updated))
        self._handle = a.all_reduce_group(self.main_group, async_op=True)
        self.pending = True
        if updated:
            self.run("pbwd")      # ... and so does this
            h = a.all_reduce_group(self.proposal_group, async_op=True)
            if h is not None:
                h.wait()
            self.run("popt")

    def finish(self) -> None:
        """Drain the pipeline: afterwards every parameter reflects every step taken."""
        if self.pending:
            if self.before_main_update is not None:
                self.before_main_update()
            self._finish_main()
