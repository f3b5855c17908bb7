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
        from nerfstudio_amd.arena import _GROUP_ALIGN as GA  # groups start and end on shardable boundaries

        assert list(g.groups) == ["fields", "proposal_networks"] and g.groups["fields"] == (0, GA) and g.groups["proposal_networks"] == (GA, 2 * GA)
        assert all(GA % (64 * w) == 0 for w in range(1, 9))
        for p_ in g.params:
            p_.grad.copy_(p_.data)
        h = g.all_reduce_span(*g.groups["fields"], async_op=True)
        h.wait()
        assert float(g.grad[:70].sum()) == 70 * 3.0 and float(g.grad[GA:GA + 15].sum()) == 15 * 10.0 * (rank + 1)
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
    from nerfstudio_amd.arena import _GROUP_ALIGN

    # tensors on 256-B boundaries; the (single) group padded to the shardable boundary with zeros
    assert arena.offsets == [0, 64] and arena.numel == _GROUP_ALIGN and arena.groups == {"all": (0, _GROUP_ALIGN)}
    assert float(arena.flat[64 + 210:].abs().sum()) == 0
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


def _toy_reference(world, steps, lr, schedule):
    """Sequential semantics: mean gradient over the ranks' batches, main group every step, proposal group on the steps
    `schedule` marks as update steps, every parameter updated before its next use."""
    wf, wp = torch.full((6,), 0.3), torch.full((6,), -0.2)
    data = [_toy_batches(r, steps) for r in range(world)]
    for k in range(steps):
        updated = schedule[k]
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


def _schedule_worker(rank, world, port, q, steps, lr, schedule):
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
            ex.iteration(updated=schedule[k])
        assert ex.pending  # the last main update is still in flight ...
        ex.finish()        # ... until the pipeline is drained
        assert not ex.pending
        q.put((rank, wf.detach().clone().numpy(), wp.detach().clone().numpy(), order))
    finally:
        dist.destroy_process_group()


def _nerfacto_schedule(steps, warmup=6, every=3):
    """ProposalNetworkSampler's update rule (ray_samplers.py:590 with nerfacto's schedule, models/nerfacto.py:208-213),
    scaled down: every step while the sampler's own step < 3, then whenever more than update_sched(step) steps passed
    since the last update — an UNEVEN pattern (gaps of 1, 2, 2, 3, 3, ...)."""
    import numpy as np

    out, since, cb = [], 0, 0
    for k in range(steps):
        upd = since > float(np.clip(np.interp(cb, [0, warmup], [0, every]), 1, every)) or cb < 3
        out.append(bool(upd))
        if upd:
            since = 0
        cb = k
        since += 1
    return out


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,steps,pattern", [(2, 7, "alternating"), (4, 12, "nerfacto")])
def test_pipelined_exchange_equals_sequential_data_parallel(world, steps, pattern):
    """2 ranks with alternating update steps and 4 ranks with the uneven update pattern of the nerfacto schedule: the
    pipelined exchange (main all-reduce in flight across the step boundary, proposal slice only on update steps) ends
    with exactly the parameters of sequential data-parallel SGD, on every rank."""
    lr = 0.05
    schedule = [k % 2 == 0 for k in range(steps)] if pattern == "alternating" else _nerfacto_schedule(steps)
    if pattern == "nerfacto":
        assert schedule[:4] == [True] * 4 and not all(schedule) and sum(schedule[4:]) >= 2
        gaps = [j - i for i, j in zip([k for k, u in enumerate(schedule) if u][:-1], [k for k, u in enumerate(schedule) if u][1:])]
        assert len(set(gaps)) > 1, gaps  # uneven
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_schedule_worker, args=(r, world, port, q, steps, lr, schedule)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wf_ref, wp_ref = _toy_reference(world, steps, lr, schedule)
    for rank, wf, wp, order in out:
        assert torch.allclose(torch.from_numpy(wf), wf_ref, atol=1e-6), (rank, wf, wf_ref)
        assert torch.allclose(torch.from_numpy(wp), wp_ref, atol=1e-6), (rank, wp, wp_ref)
        assert order == out[0][3]  # every rank issues the same segments in the same order (the collectives pair up)
    order = out[0][3]
    if pattern == "alternating":
        # step 0: pfwd, main, pbwd, popt (main update pending); step 1 starts with pfwd BEFORE the pending main update
        assert order[:9] == ["pfwd", ("main", True), "pbwd", "popt", "pfwd", "mopt", ("main", False), "pfwd", "mopt"]
    assert order.count("popt") == order.count("pbwd") == sum(schedule) and order.count("mopt") == steps
    assert order[-1] == "mopt"  # finish() drains the pending main-field update


# ---------------------------------------------------------------------------------------------------------------------
# Sharded optimiser (reduce-scatter -> Adam on the rank's 1/N shard -> all-gather) == replicated Adam behind an all-reduce
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_adam(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-15, grad_scale=1.0, hyper_dev=None):
    """torch.optim.Adam's update (torch/optim/adam.py, single tensor, no amsgrad / weight decay) on arena slices — the
    CPU stand-in for csrc/misc.hip's kernel in these gloo tests (elementwise, like the kernel)."""
    import math

    b1, b2 = betas
    g = grads * grad_scale
    exp_avg.lerp_(g, 1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    params.addcdiv_(exp_avg, denom, value=-lr / bc1)


def _sharded_worker(rank, world, port, q, steps, schedule):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfstudio_amd import functional as F
        from nerfstudio_amd.arena import ParamArena
        from nerfstudio_amd.dp_schedule import PipelinedExchange

        F.adam_step = _cpu_adam
        out = {}
        for mode in ("allreduce", "sharded"):
            torch.manual_seed(5)
            table = torch.nn.Parameter(torch.randn(90000, 2) * 1e-2)   # "hash table": most of the group
            w = torch.nn.Parameter(torch.randn(64, 32) * 0.1)
            pt = torch.nn.Parameter(torch.randn(40000, 2) * 1e-2)
            pw = torch.nn.Parameter(torch.randn(16, 10) * 0.1)
            arena = ParamArena({"fields": [table, w], "proposal_networks": [pt, pw]}, lr=1e-2, eps=1e-15)
            state = {"k": 0}

            def run(name, arena=arena, mode=mode):
                k = state["k"]
                if name in (("main", True), ("main", False), "pbwd"):
                    grp = "proposal_networks" if name == "pbwd" else "fields"
                    a, b = arena.groups[grp]
                    g = torch.Generator().manual_seed(1000 * rank + 10 * k + (name == "pbwd"))
                    arena.grad[a:b] = torch.randn(b - a, generator=g) * (1e-3 if grp == "fields" else 1e-4)
                    used = sum((p.numel() + 63) // 64 * 64 for p in arena.group_params[grp])
                    arena.grad[a + used:b] = 0.0  # the group's alignment padding never carries gradient
                elif name in ("mopt", "popt"):
                    grp = "fields" if name == "mopt" else "proposal_networks"
                    if mode == "sharded":
                        arena.step_shard(grp, grad_scale=1.0 / world)
                    else:
                        arena.step(grad_scale=1.0 / world, groups=[grp])

            ex = PipelinedExchange(arena, run, sharded=(mode == "sharded"))
            for k in range(steps):
                state["k"] = k
                ex.iteration(updated=schedule[k])
            ex.finish()
            out[mode] = (arena.flat.clone(), arena.exp_avg.clone(), arena.exp_avg_sq.clone(),
                         {n: arena.shard_span(n) for n in arena.groups}, dict(arena.step_counts))
        q.put((rank, {m: tuple(t.numpy() if torch.is_tensor(t) else t for t in v) for m, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,steps,pattern", [(2, 7, "alternating"), (4, 12, "nerfacto")])
def test_sharded_adam_equals_replicated_adam_bit_for_bit(world, steps, pattern):
    """arena mode "sharded" (reduce-scatter -> Adam on the rank's 1/N shard -> all-gather; dp_schedule.PipelinedExchange
    with sharded=True) against the replicated Adam behind the all-reduce, same per-rank gradients, world 2 with alternating
    update steps and world 4 with the uneven nerfacto update pattern: the PARAMETERS are equal bit for bit on every rank
    and between the ranks, the rank's shard of both moments equals the replicated moments, moments outside the shard were
    never touched, and both groups' step counters agree."""
    import numpy as np

    schedule = [k % 2 == 0 for k in range(steps)] if pattern == "alternating" else _nerfacto_schedule(steps)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, steps, schedule)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = res[0]["allreduce"][0]
    assert np.abs(ref).max() > 0 and not np.array_equal(ref, 0 * ref)
    for rank in range(world):
        flat_a, m_a, v_a, _, steps_a = res[rank]["allreduce"]
        flat_s, m_s, v_s, spans, steps_s = res[rank]["sharded"]
        np.testing.assert_array_equal(flat_a, ref)            # replicated: every rank the same
        np.testing.assert_array_equal(flat_s.view(np.int32), ref.view(np.int32))  # sharded: the same BITS
        assert steps_a == steps_s and steps_s["fields"] == steps and steps_s["proposal_networks"] == sum(schedule)
        for name, (sa, sb) in spans.items():
            np.testing.assert_array_equal(m_s[sa:sb], m_a[sa:sb])
            np.testing.assert_array_equal(v_s[sa:sb], v_a[sa:sb])
        own = np.zeros(flat_s.shape, dtype=bool)
        for sa, sb in spans.values():
            own[sa:sb] = True
        assert not m_s[~own].any() and not v_s[~own].any(), "moments outside the rank's shards must stay untouched"
        assert m_a[~own].any()  # (the replicated optimiser does hold them)


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


# ---------------------------------------------------------------------------------------------------------------------
# the driver's multi-GPU launch line, rehearsed on CPU (bench.py --dry-run: gloo instead of RCCL, no kernels)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(600)
@pytest.mark.parametrize("dp_mode", ["allreduce", "sharded"])
def test_bench_launch_line_dry_run_under_torchrun(dp_mode):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 2 --steps K --warmup W` — exactly the driver's line plus --dry-run: argument and RANK / WORLD_SIZE / MASTER_*
    handling, process-group setup, the real model / arena / compact prefix / pipelined exchange over gloo, ONE JSON line
    from rank 0 with the contract's keys."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--dry-run", "--dp-mode", dp_mode]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=540, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["scaling"] == "weak"
    assert out["config"]["max_abs_error"] <= 1e-5 and 0 < out["config"]["compact_rows"] < out["config"]["prefix_rows"]
    assert out["config"]["dp_mode"] == dp_mode and out["config"]["ranks"] == 2


def _ddp_guard_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfstudio_amd.fused_step import ddp_reason
        from nerfstudio_amd.instant_ngp import InstantNGPModelConfig, NGPModel
        from nerfstudio_amd.nerfacto import NerfactoModel, NerfactoModelConfig

        aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        args = [{"hidden_dim": 16, "log2_hashmap_size": 7, "num_levels": 5, "max_res": r, "use_linear": False} for r in (32, 64)]
        m = NerfactoModel(NerfactoModelConfig(log2_hashmap_size=8, proposal_net_args_list=args, fused_train_step=True), aabb, 3).train()
        g = NGPModel(InstantNGPModelConfig(grid_resolution=8, grid_levels=1, log2_hashmap_size=8, fused_train_step=True), aabb, 3).train()
        msgs = []
        for model in (m, g):
            try:
                model._fused_step()
                msgs.append("no error")
            except NotImplementedError as e:
                msgs.append(str(e))
        q.put((rank, ddp_reason(), msgs))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_fused_steps_refuse_more_than_one_rank():
    """The fused steps write gradients straight into param.grad — DistributedDataParallel's reducer (which the reference's
    pipeline wraps the model in when world_size > 1, pipelines/base_pipeline.py:279-282) would never see them and the ranks
    would diverge silently. With a process group of two ranks both models say so instead."""
    from nerfstudio_amd.fused_step import ddp_reason

    assert ddp_reason() is None  # no process group: a single process may use them
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_guard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, reason, msgs in got:
        assert reason and "DistributedDataParallel" in reason
        assert all("DistributedDataParallel" in msg and "module path" in msg for msg in msgs), msgs
