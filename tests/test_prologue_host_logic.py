"""Host logic of trainer.HipTrainer's step prologue (CPU tier, no kernels): the rows the device's first node will read must be,
iteration after iteration, the scalars the host computes for that iteration — for the ring in host memory and for the table
predicted ahead, with the deferred main-field Adam and without, across `finish()` calls and rewinds of the training state
(bench.py's repeated windows). The device side is a stand-in that does what `nsamd_step_prologue` does with the row counter;
the kernel itself and the real trainer are covered on the GPU (tests/test_gpu_fused_launches.py)."""
import numpy as np
import pytest
import torch

from nerfstudio_amd import trainer as T


class _Event:
    def record(self, stream=None):
        pass

    def synchronize(self):
        pass


class _Sampler:
    """ProposalNetworkSampler's update rule (model_components/ray_samplers.py:560-575, 590-599)."""

    def __init__(self):
        self._step, self._steps_since_update, self._anneal = 0, 0, 0.0

    def updated_this_step(self):
        every = int(np.clip(1 + 4 * self._step / 5000, 1, 5))
        return self._steps_since_update > every or self._step < 10

    def mark_updated(self):
        self._steps_since_update = 0


class _Model:
    def __init__(self):
        self.proposal_sampler, self.step = _Sampler(), 0

    def set_step(self, step):  # BEFORE_TRAIN_ITERATION: the anneal exponent
        self.step = step
        self.proposal_sampler._anneal = float(np.clip(step / 1000.0, 0, 1))

    def after_step(self, step):  # AFTER_TRAIN_ITERATION
        ps = self.proposal_sampler
        ps._step = step
        ps._steps_since_update += 1


class _Arena:
    def __init__(self):
        self.groups = {"fields": (0, 1), "proposal_networks": (1, 2)}
        self.step_counts = {"fields": 0, "proposal_networks": 0}
        self.betas, self.lr = (0.9, 0.999), 1e-2


def _trainer(mode, defer, monkeypatch):
    from nerfstudio_amd.schedulers import nerfacto_schedulers

    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(T.N, "current_stream", lambda: None)  # (the events are recorded on torch's current stream)
    t = T.HipTrainer.__new__(T.HipTrainer)
    t.model, t.arena, t.step, t.slots, t.defer = _Model(), _Arena(), 0, 8, defer
    t._pending_main, t.exchange, t.cam_inside, t.cam_group, t.drive_callbacks = False, None, False, None, True
    sched = nerfacto_schedulers()
    t.lr_source = lambda group, it: sched[group].get_lr(max(it, 0), 1e-2)
    t.prologue, t.prologue_table, t.prologue_ring = True, mode == "table", mode == "ring"
    t.table_rows, t.ring_rows = 128, 256
    t.hyper = np.zeros(8, np.float32)
    t.device_row = [0]  # nsamd_step_prologue's counter[0]
    if mode == "ring":
        t.ring_np, t._ring_pos, t._ring_events = np.zeros((256, 8), np.float32), 0, [None] * 4
    else:
        t.table_host_np, t.device_table = np.zeros((128, 8), np.float32), np.zeros((128, 8), np.float32)
        t._table_pos, t._table_valid, t._table_event, t._table_base = 0, False, None, 0
        t._row_scratch, t._row_unread = np.zeros(8, np.float32), np.zeros(8, bool)
        t._row_unread[0:2] = True
        t.refills = 0

        class _DeviceTable:
            def copy_(self, src, non_blocking=False):
                t.device_table[:] = t.table_host_np
                t.refills += 1

        class _Counter:
            def __getitem__(self, key):
                class _View:
                    def zero_(self):
                        t.device_row[0] = 0

                    def fill_(self, value):
                        t.device_row[0] = int(value)

                return _View()

        t.hyper_table, t.step_counter = _DeviceTable(), _Counter()
        t.table_host = type("H", (), {"reshape": lambda self, *a: None})()
    return t


def _iteration(t):
    """train_iteration's host side around a stand-in for the replayed graph's first node."""
    ps = t.model.proposal_sampler
    updated = ps.updated_this_step()
    t.model.set_step(t.step)
    want = np.zeros(8, np.float32)
    pending = t._have_pending
    t._hyper_row(want, t.step, t.arena.step_counts, pending)
    t._push_hyper()
    rows = t.ring_np if t.prologue_ring else t.device_table
    t.hyper[:] = rows[t.device_row[0] % len(rows)]
    t.device_row[0] += 1
    read = slice(2, 8) if (t.defer and not pending) else slice(0, 8)  # (no pending update: the main-field scalars are not read)
    assert np.array_equal(t.hyper[read], want[read]), (t.step, t.hyper, want)
    if t.defer:
        if pending:
            t.arena.step_counts["fields"] += 1
        t._pending_main = True
    else:
        t.arena.step_counts["fields"] += 1
    if updated:
        t.arena.step_counts["proposal_networks"] += 1
        ps.mark_updated()
    t.model.after_step(t.step)
    t.step += 1


def _finish(t):
    if t._pending_main:
        t.arena.step_counts["fields"] += 1
        t._pending_main = False


@pytest.mark.parametrize("defer", [True, False])
@pytest.mark.parametrize("mode", ["ring", "table"])
def test_every_iteration_reads_the_scalars_the_host_computes(monkeypatch, mode, defer):
    t = _trainer(mode, defer, monkeypatch)
    for _ in range(30):
        _iteration(t)
    _finish(t)
    ps = t.model.proposal_sampler
    snap = (t.step, dict(t.arena.step_counts), (ps._step, ps._steps_since_update, ps._anneal), t.model.step)
    before = getattr(t, "refills", 0)
    for _ in range(7):  # bench.py: the same window repeated from the restored training state
        t.step = snap[0]
        t.arena.step_counts.update(snap[1])
        ps._step, ps._steps_since_update, ps._anneal = snap[2]
        t.model.step = snap[3]
        for _ in range(20):
            _iteration(t)
        _finish(t)
    if mode == "table":  # a rewind or a finish() inside the table's rows costs no new table
        assert t.refills - before <= 1, (before, t.refills)
    for _ in range(400):  # past the end of the ring and of several tables
        _iteration(t)
    if mode == "table":
        assert t.refills <= 2 + (30 + 20 + 400) // 128 + 1
