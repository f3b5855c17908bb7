"""CPU, world_size 2, gloo: the data-parallel exchange of the path — replicated parameters in one flat arena,
per-rank gradients, ONE all-reduce of the gradient arena, mean folded into the optimiser scale (the reference gets
the same numbers from DistributedDataParallel, pipelines/base_pipeline.py:279-282)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfstudio_amd.arena import ParamArena

        torch.manual_seed(123 + rank)  # different init per rank on purpose: broadcast must fix it
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        shared = torch.nn.Parameter(torch.randn(11, 2))
        params = list(net.parameters()) + [shared, shared]  # a parameter registered twice (proposal-net hash table)
        arena = ParamArena(params)
        assert len(arena.params) == 5
        arena.broadcast_params(src=0)
        ref0 = [p.detach().clone() for p in arena.params]
        # per-rank batch ("each rank draws its own rays", scripts/train.py:98)
        g = torch.Generator().manual_seed(1000 + rank)
        x = torch.randn(16, 7, generator=g)
        arena.zero_grad()
        loss = net(x).pow(2).mean() + (shared * (rank + 1)).sum()
        loss.backward()
        for p, off in zip(arena.params, arena.offsets):  # gradients landed in the arena views
            assert p.grad.data_ptr() == arena.grad.data_ptr() + 4 * off
        local = arena.grad.clone()
        scale = arena.all_reduce()
        assert scale == 1.0 / world
        summed = arena.grad.clone()
        # grouped arena: asynchronous all-reduce of ONE group's slice leaves the other group's gradients local
        g = ParamArena({"fields": [torch.nn.Parameter(torch.full((70,), float(rank + 1)))],
                        "proposal_networks": [torch.nn.Parameter(torch.full((5, 3), 10.0 * (rank + 1)))]})
        assert list(g.groups) == ["fields", "proposal_networks"] and g.groups["fields"] == (0, 128) and g.groups["proposal_networks"] == (128, 192)
        for p_ in g.params:
            p_.grad.copy_(p_.data)
        h = g.all_reduce_span(*g.groups["fields"], async_op=True)
        h.wait()
        assert float(g.grad[:70].sum()) == 70 * 3.0 and float(g.grad[128:143].sum()) == 15 * 10.0 * (rank + 1)
        q.put((rank, [r.numpy() for r in ref0], local.numpy(), summed.numpy(), arena.numel))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_arena_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (_, p0, l0, s0, n0), (_, p1, l1, s1, n1) = res
    assert n0 == n1 and n0 % 64 == 0
    for a, b in zip(p0, p1):  # replicated parameters after the broadcast
        assert (a == b).all()
    assert not (l0 == l1).all()  # different rays -> different local gradients
    import numpy as np

    np.testing.assert_allclose(s0, l0 + l1, rtol=1e-6, atol=1e-7)  # the arena holds the SUM on every rank
    np.testing.assert_array_equal(s0, s1)


def test_arena_single_process_layout():
    from nerfstudio_amd.arena import ParamArena

    a = torch.nn.Parameter(torch.arange(10.0))
    b = torch.nn.Parameter(torch.ones(3, 70))
    arena = ParamArena([a, b])
    assert arena.offsets == [0, 64] and arena.numel == 64 + 256
    assert a.data_ptr() == arena.flat.data_ptr() and b.data_ptr() == arena.flat.data_ptr() + 4 * 64
    assert torch.equal(arena.flat[:10], torch.arange(10.0)) and float(arena.flat[10:64].abs().sum()) == 0
    (a.sum() * 2 + b.sum()).backward()
    assert float(arena.grad[:10].sum()) == 20 and float(arena.grad[64:64 + 210].sum()) == 210
    arena.zero_grad()
    assert float(arena.grad.abs().sum()) == 0 and a.grad.data_ptr() == arena.grad.data_ptr()
    assert arena.all_reduce() == 1.0  # no process group: no-op


# ---------------------------------------------------------------------------------------------------------------------
# dp_schedule.PipelinedExchange: the main-group all-reduce stays in flight across the step boundary
# ---------------------------------------------------------------------------------------------------------------------
def _toy_batches(rank, steps):
    g = torch.Generator().manual_seed(77 + rank)
    return [(torch.randn(8, 6, generator=g), torch.randn(8, generator=g)) for _ in range(steps)]


def _toy_reference(world, steps, lr):
    """Sequential semantics: mean gradient over the ranks' batches, main group every step, proposal group on update
    steps (even steps), every parameter updated before its next use."""
    wf, wp = torch.full((6,), 0.3), torch.full((6,), -0.2)
    data = [_toy_batches(r, steps) for r in range(world)]
    for k in range(steps):
        updated = k % 2 == 0
        gf, gp = torch.zeros(6), torch.zeros(6)
        for r in range(world):
            x, y = data[r][k]
            s = torch.tanh(x @ wp)                       # "proposal forward"
            err = (x * s[:, None]) @ wf - y              # "main forward" on the proposal output
            gf += 2 * ((x * s[:, None]) * err[:, None]).mean(0)
            if updated:
                ds = 2 * err * (x @ wf) / len(y)         # dL/ds
                gp += ((1 - s * s) * ds) @ x
        wf = wf - lr * gf / world
        if updated:
            wp = wp - lr * gp / world
    return wf, wp


def _schedule_worker(rank, world, port, q, steps, lr):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfstudio_amd.arena import ParamArena
        from nerfstudio_amd.dp_schedule import PipelinedExchange

        wf, wp = torch.nn.Parameter(torch.full((6,), 0.3)), torch.nn.Parameter(torch.full((6,), -0.2))
        arena = ParamArena({"fields": [wf], "proposal_networks": [wp]})
        data = _toy_batches(rank, steps)
        state = {"k": 0}
        order = []

        def run(name):
            order.append(name)
            x, y = data[state["k"]]
            with torch.no_grad():
                if name == "pfwd":
                    state["s"] = torch.tanh(x @ wp)
                elif name in (("main", True), ("main", False)):
                    arena.zero_grad(["fields"])
                    s = state["s"]
                    err = (x * s[:, None]) @ wf - y
                    wf.grad += 2 * ((x * s[:, None]) * err[:, None]).mean(0)
                    state["ds"] = 2 * err * (x @ wf) / len(y)
                elif name == "pbwd":
                    arena.zero_grad(["proposal_networks"])
                    s = state["s"]
                    wp.grad += ((1 - s * s) * state["ds"]) @ x
                elif name == "mopt":
                    a, b = arena.groups["fields"]
                    arena.flat[a:b] -= lr * arena.grad[a:b] / world
                elif name == "popt":
                    a, b = arena.groups["proposal_networks"]
                    arena.flat[a:b] -= lr * arena.grad[a:b] / world

        ex = PipelinedExchange(arena, run)
        for k in range(steps):
            state["k"] = k
            ex.iteration(updated=(k % 2 == 0))
        assert ex.pending  # the last main update is still in flight ...
        ex.finish()        # ... until the pipeline is drained
        assert not ex.pending
        q.put((rank, wf.detach().clone().numpy(), wp.detach().clone().numpy(), order[:9]))
    finally:
        dist.destroy_process_group()


def test_pipelined_exchange_equals_sequential_data_parallel():
    world, steps, lr = 2, 7, 0.05
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_schedule_worker, args=(r, world, port, q, steps, lr)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wf_ref, wp_ref = _toy_reference(world, steps, lr)
    for rank, wf, wp, order in out:
        assert torch.allclose(torch.from_numpy(wf), wf_ref, atol=1e-6), (rank, wf, wf_ref)
        assert torch.allclose(torch.from_numpy(wp), wp_ref, atol=1e-6), (rank, wp, wp_ref)
    # step 0: pfwd, main, pbwd, popt (main update pending); step 1 starts with pfwd BEFORE the pending main update
    assert out[0][3] == ["pfwd", ("main", True), "pbwd", "popt", "pfwd", "mopt", ("main", False), "pfwd", "mopt"]


# ---------------------------------------------------------------------------------------------------------------------
# ParamArena.all_reduce_group with a registered compact table prefix == plain all-reduce of the whole slice
# ---------------------------------------------------------------------------------------------------------------------
def _compact_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfstudio_amd import functional as F
        from nerfstudio_amd.arena import ParamArena

        spec = F.HashGridSpec(num_levels=6, min_res=2, max_res=40, log2_hashmap_size=8)
        rows, idx = spec.reachable_prefix()
        assert 0 < idx.numel() < rows < spec.num_levels * spec.table_size
        table = torch.nn.Parameter(torch.zeros(spec.num_levels * spec.table_size, 2))
        other = torch.nn.Parameter(torch.zeros(37))
        prop = torch.nn.Parameter(torch.zeros(5, 3))
        arena = ParamArena({"fields": [other, table], "proposal_networks": [prop]})  # table NOT first: two dense spans
        g = torch.Generator().manual_seed(5 + rank)
        with torch.no_grad():
            table.grad[idx] = torch.randn(idx.numel(), 2, generator=g)          # only reachable rows of the prefix
            table.grad[rows:] = torch.randn(table.shape[0] - rows, 2, generator=g)  # fine levels: dense
            other.grad.copy_(torch.randn(37, generator=g))
            prop.grad.fill_(float(rank + 1))
        expect = arena.grad.clone()
        dist.all_reduce(expect)  # what a plain all-reduce of everything gives
        arena.register_compact(table, rows, idx)
        for async_op in (True, False):
            saved = arena.grad.clone()
            h = arena.all_reduce_group("fields", async_op=async_op)
            h.wait()
            a, b = arena.groups["fields"]
            assert torch.equal(arena.grad[a:b], expect[a:b]), async_op
            pa, pb = arena.groups["proposal_networks"]
            assert torch.equal(arena.grad[pa:pb], saved[pa:pb])  # the other group is untouched
            arena.grad.copy_(saved)
        q.put((rank, int(idx.numel()), rows))
    finally:
        dist.destroy_process_group()


def test_compact_table_prefix_exchange_equals_full_all_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_compact_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1:] == out[1][1:]
