#!/usr/bin/env python3
"""GPU diagnostic (not part of the product): where does the hash-grid gradient scatter spend its time?
Times nsamd_hashgrid_encode_fwd/bwd one level at a time for several grid resolutions and point orderings, plus the
CPU-oracle step at several thread counts (to pick a fair cpu_baseline thread count)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nerfstudio_amd import _native as N

lib = N.load()
dev = torch.device("cuda")
torch.manual_seed(0)
n_rays, S = 4096, 48
M = n_rays * S

def ray_points():
    o = torch.randn(n_rays, 3) * 0.5
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3), dim=-1)
    t = torch.sort(torch.rand(n_rays, S) * 4.0, dim=-1)[0]
    x = o[:, None] + d[:, None] * t[..., None]
    mag = x.abs().amax(-1, keepdim=True)
    x = torch.where(mag < 1, x, (2 - 1 / mag) * (x / mag))
    return ((x + 2) / 4).clamp(0.001, 0.999).reshape(-1, 3)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

pts_ray = ray_points().to(dev).contiguous()
pts_rand = torch.rand(M, 3, device=dev)
for log2T in (19,):
    T = 1 << log2T
    table = torch.randn(T, 2, device=dev)
    dtable = torch.zeros(T, 2, device=dev)
    for name, pts in (("ray-ordered", pts_ray), ("uniform-random", pts_rand)):
        for res in (16, 30, 58, 111, 212, 406, 776, 1482, 2047):
            g = N.make_grid(1, log2T, [float(res)])
            enc = torch.empty(2, M, device=dev)
            denc = torch.randn(2, M, device=dev)
            P = N.make_points(positions=pts)
            f = lambda: N.check(lib.nsamd_hashgrid_encode_fwd(P, M, 0, N.Aabb(), N.ptr(table), g, N.ptr(enc), 1, M, None, N.stream()), "f")
            b = lambda: N.check(lib.nsamd_hashgrid_encode_bwd(P, M, 0, N.Aabb(), N.ptr(table), g, N.ptr(denc), 1, M, N.ptr(dtable), None, None, 0, N.stream()), "b")
            tf, tb = timeit(f), timeit(b)
            print(f"T=2^{log2T} {name:15s} res={res:5d}  fwd {tf*1e3:8.1f} us   bwd {tb*1e3:8.1f} us   "
                  f"({M*16/tb/1e6:7.2f} G atomics/s)", flush=True)

# pure atomic microbenchmarks through torch (index_add_ uses atomics): random vs contended
idx_rand = torch.randint(0, 1 << 20, (M * 16,), device=dev)
idx_hot = torch.randint(0, 4096, (M * 16,), device=dev)
val = torch.randn(M * 16, device=dev)
buf = torch.zeros(1 << 20, device=dev)
for name, idx in (("random 4MB", idx_rand), ("4096 hot addresses", idx_hot)):
    t = timeit(lambda: buf.index_add_(0, idx, val))
    print(f"torch index_add_ {name:20s}: {t*1e3:8.1f} us  ({M*16/t/1e6:7.2f} G atomics/s)", flush=True)

if "--cpu" in sys.argv:
    sys.argv = [sys.argv[0]]
    import bench
    for th in (8, 16, 32, 64):
        torch.set_num_threads(th)
        r = bench.cpu_baseline(n_rays=256, steps=2)
        print("cpu threads", th, r["value"], "rays/s", flush=True)
