#!/usr/bin/env python3
"""Clock stamps of the fused proposal resampling kernel (pdf_resample_kernel<true>, one wavefront per ray) on the benchmark's
first proposal level (4096 rays, 256 -> 96 samples): instrumented build via NSAMD_LIB (make -C nerfstudio_amd/csrc probe)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["NSAMD_LIB"] = os.path.join(ROOT, "nerfstudio_amd", "libnsamd_probe.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nerfstudio_amd import _native as N  # noqa: E402

lib = N.load()
n, S0, S1 = 4096, 256, 96
dev = "cuda"
torch.manual_seed(0)
t_bins = torch.sort(torch.rand(n, S0 + 1, device=dev) * 5 + 0.05, dim=-1).values
s_bins = torch.sort(torch.rand(n, S0 + 1, device=dev), dim=-1).values
dens = torch.rand(n, S0, device=dev) * 3
u = torch.linspace(0.0, 1.0 - 1.0 / (S1 + 1), S1 + 1, device=dev)
jit = torch.rand(n, device=dev)
nears, fars = torch.full((n,), 0.05, device=dev), torch.full((n,), 1000.0, device=dev)
anneal = torch.tensor([0.7], device=dev)
w, dm = torch.empty(n, S0, device=dev), torch.empty(n, device=dev)
so, to = torch.empty(n, S1 + 1, device=dev), torch.empty(n, S1 + 1, device=dev)
st = N.stream()


def run():
    N.check(lib.nsamd_proposal_resample(N.ptr(t_bins), N.ptr(s_bins), N.ptr(dens), S0, N.ptr(u), N.ptr(jit), N.ptr(nears),
                                        N.ptr(fars), 1.0, N.ptr(anneal), 0.01, 1e-5, 1.0 / (2 * (S1 + 1)), 0, n, S1, N.ptr(w),
                                        N.ptr(dm), N.ptr(so), N.ptr(to), st), "resample")


for _ in range(5):
    run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record()
torch.cuda.synchronize()
print(f"proposal_resample 256 -> 96, 4096 rays: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch")
buf = torch.zeros(n, 64, dtype=torch.int64, device=dev)
lib.cdll.nsamd_probe_set_clocks_sampler.argtypes = [C.c_void_p]
assert lib.cdll.nsamd_probe_set_clocks_sampler(buf.data_ptr()) == 0
run()
torch.cuda.synchronize()
assert lib.cdll.nsamd_probe_set_clocks_sampler(None) == 0
t = buf.cpu()
t = t[t[:, 0] > 0]
labels = [(1, "all global loads issued, previous edges -> LDS"), (2, "weights: 4 x (scan, 2 exp, store)"), (3, "median depth"),
          (4, "anneal pow + padding + total"), (5, "pdf / cdf scans"), (6, "inverse-CDF search + stores")]
prev = 0
print(f"{t.shape[0]} waves; wave lifetime mean {(t[:, 6] - t[:, 0]).double().mean().item():.0f} clocks")
for slot, label in labels:
    d = (t[:, slot] - t[:, prev]).double()
    print(f"   {prev} -> {slot}  {label:48s} mean {d.mean().item():8.0f}  min {d.min().item():7.0f}  max {d.max().item():8.0f}")
    prev = slot
