#!/usr/bin/env python3
"""GPU diagnostic: cost of the table-gradient scatter of the PROPOSAL grids (ray mode, real piecewise sample layout) as a
function of the fraction of samples with a non-zero gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfstudio_amd import _native as N
from nerfstudio_amd import functional as F
lib = N.load(); dev = torch.device("cuda"); torch.manual_seed(0)
n = 4096
o = (torch.randn(n, 3) * 0.5).to(dev); d = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1).to(dev)
nears = torch.full((n,), 0.05, device=dev); fars = torch.full((n,), 1000.0, device=dev)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
CONFIGS = (("prop0", 256, F.HashGridSpec(5, 16, 128, 17)), ("prop1", 96, F.HashGridSpec(5, 16, 256, 17)),
           ("main", 48, F.HashGridSpec(16, 16, 2048, 19)))
if "--main" in sys.argv:
    CONFIGS = CONFIGS[2:]
for name, S, spec in CONFIGS:
    s_bins, t_bins = F.piecewise_bins(nears, fars, S, torch.rand(n, device=dev))
    M = n * S
    table = torch.randn(spec.num_levels * spec.table_size, 2, device=dev); dtable = torch.zeros_like(table)
    SET = "--accumulate" not in sys.argv  # default: the write-only entry point the training step uses
    ws, ws_n = F._scatter_workspace(spec, dev, M, write_only=SET)
    fn = lib.nsamd_hashgrid_encode_bwd_set if SET else lib.nsamd_hashgrid_encode_bwd
    P = N.make_points(None, o, d, t_bins, S)
    for frac in ((1.0,) if "--dense" in sys.argv else (0.0, 0.1, 0.5, 1.0)):
        denc = torch.randn(spec.out_dim, M, device=dev)
        keep = (torch.rand(M, device=dev) < frac).float()
        denc = (denc * keep).contiguous()
        b = lambda: N.check(fn(P, M, N.XFORM_CONTRACT, N.Aabb(), N.ptr(table), spec.native(), N.ptr(denc), 1, M,
                                                         N.ptr(dtable), None, N.ptr(ws), ws_n, N.stream()), "b")
        print(f"{name} S={S:3d} nonzero-gradient fraction {frac:4.1f}: scatter {timeit(b)*1e3:8.1f} us", flush=True)
