#!/usr/bin/env python3
"""GPU diagnostic: binned scatter (hash_bwd_bin + hash_bwd_apply) time vs grid resolution, 16 identical levels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfstudio_amd import _native as N
from nerfstudio_amd import functional as F
lib = N.load(); dev = torch.device("cuda"); torch.manual_seed(0)
n_rays, S = 4096, 48; M = n_rays * S
o = torch.randn(n_rays, 3) * 0.5
d = torch.nn.functional.normalize(torch.randn(n_rays, 3), dim=-1)
t = torch.sort(torch.rand(n_rays, S) * 4.0, dim=-1)[0]
x = o[:, None] + d[:, None] * t[..., None]
mag = x.abs().amax(-1, keepdim=True)
x = torch.where(mag < 1, x, (2 - 1 / mag) * (x / mag))
pts_ray = ((x + 2) / 4).clamp(0.001, 0.999).reshape(-1, 3).to(dev).contiguous()
pts_rand = torch.rand(M, 3, device=dev)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
L, log2T = 16, 19
spec = F.HashGridSpec(L, 16, 2048, log2T)
table = torch.randn(L << log2T, 2, device=dev); dtable = torch.zeros_like(table)
ws, ws_n = F._scatter_workspace(spec, dev, M)
denc = torch.randn(2 * L, M, device=dev)
for name, pts in (("ray-ordered", pts_ray), ("uniform-random", pts_rand)):
    for res in (16, 58, 212, 776, 2047):
        g = N.make_grid(L, log2T, [float(res)] * L)
        P = N.make_points(positions=pts)
        b = lambda: N.check(lib.nsamd_hashgrid_encode_bwd(P, M, 0, N.Aabb(), N.ptr(table), g, N.ptr(denc), 1, M, N.ptr(dtable), None, N.ptr(ws), ws_n, N.stream()), "b")
        N.PROFILE = None
        tb = timeit(b)
        print(f"{name:15s} res={res:5d} x16 levels: bin+apply {tb*1e3:8.1f} us", flush=True)
