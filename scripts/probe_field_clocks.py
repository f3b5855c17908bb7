#!/usr/bin/env python3
"""Where do the main-field MLP kernels spend their time? Builds csrc/field_mlp.hip ALONE with -DNSAMD_PROBE_CLOCKS
(lane 0 of every wave stamps the shader clock at phase boundaries) into nerfstudio_amd/libnsamd_probe_field.so — a probe
library, never the product — runs forward and backward on the benchmark's shape (4096 rays x 48 samples) and prints the mean
cycles between stamps over all waves. GPU box only:  python scripts/probe_field_clocks.py [--no-build]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfstudio_amd import _native as N  # noqa: E402

SO = os.path.join(ROOT, "nerfstudio_amd", "libnsamd_probe_field.so")
if "--no-build" not in sys.argv or not os.path.exists(SO):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-munsafe-fp-atomics", "-fPIC", "-DNSAMD_PROBE_CLOCKS", "-shared",
                           os.path.join(ROOT, "nerfstudio_amd", "csrc", "field_mlp.hip"),
                           os.path.join(ROOT, "nerfstudio_amd", "csrc", "scatter.hip"), "-o", SO])
lib = C.CDLL(SO)
vp, i64 = C.c_void_p, C.c_int64
lib.nsamd_field_mlp_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, N.FieldMlp, vp, vp, vp]
lib.nsamd_field_mlp_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, N.FieldMlp, vp, vp, vp, N.FieldMlpGrads, vp, i64, vp]
lib.nsamd_probe_set_clocks_field.argtypes = [vp]
lib.nsamd_field_mlp_bwd_scatter.argtypes = [N.Points, C.c_int, N.Aabb, N.Grid, vp, vp, vp, vp, vp, i64, i64, N.FieldMlp, vp, vp, vp,
                                            N.FieldMlpGrads, vp, i64, vp, vp, i64, vp]
lib.nsamd_field_mlp_bwd_scatter_workspace.argtypes = [N.Grid, i64, C.POINTER(C.c_int64)]
lib.nsamd_field_mlp_bwd_scatter_workspace.restype = C.c_int64
ROUTE = "--route" in sys.argv  # the backward that emits the scatter's pass-1 records (nsamd_field_mlp_bwd_scatter)

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
rays, S = 4096, 48
M = rays * S


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


enc = rnd(32, M, scale=0.1)
sel = torch.ones(M, device=dev)
dirs = torch.nn.functional.normalize(rnd(rays, 3), dim=-1)
cams = torch.randint(0, 100, (rays,), generator=g).to(dev)
shapes = [(64, 32), (64,), (16, 64), (16,), (64, 63), (64,), (64, 64), (64,), (3, 64), (3,)]
params = [rnd(*s, scale=0.2) for s in shapes]
emb = rnd(100, 32, scale=0.1)
fm = N.FieldMlp(*(p.data_ptr() for p in params), emb.data_ptr(), 100, 1.0)
grads_t = [torch.zeros_like(p) for p in params] + [torch.zeros_like(emb)]
grads = N.FieldMlpGrads(*(t.data_ptr() for t in grads_t))
dens, rgb = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
ddens, drgb, denc = rnd(M, scale=1e-3), rnd(M, 3, scale=1e-3), torch.empty(32, M, device=dev)
ws = torch.empty(64 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream


def fwd():
    assert lib.nsamd_field_mlp_fwd(enc.data_ptr(), sel.data_ptr(), dirs.data_ptr(), cams.data_ptr(), None, S, M, fm,
                                   dens.data_ptr(), rgb.data_ptr(), st) == 0


def bwd():
    assert lib.nsamd_field_mlp_bwd(enc.data_ptr(), sel.data_ptr(), dirs.data_ptr(), cams.data_ptr(), None, S, M, fm,
                                   ddens.data_ptr(), drgb.data_ptr(), denc.data_ptr(), grads, ws.data_ptr(), ws.numel(), st) == 0


if ROUTE:
    from nerfstudio_amd import functional as F

    spec = F.HashGridSpec(num_levels=16, min_res=16, max_res=2048, log2_hashmap_size=19)
    origins = rnd(rays, 3, scale=0.5)
    tb = torch.sort(torch.rand(rays, S + 1, generator=g) * 2.0 + 0.05, dim=-1)[0].to(dev)
    pts = N.make_points(None, origins, dirs, tb, S)
    state = C.c_int64(0)
    words = lib.nsamd_field_mlp_bwd_scatter_workspace(spec.native(), M, C.byref(state))
    sws = torch.empty(words, device=dev)
    sws[:state.value].zero_()
    dtable = torch.empty(16 << 19, 2, device=dev)
    box = N.make_aabb(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))

    def bwd():  # noqa: F811
        assert lib.nsamd_field_mlp_bwd_scatter(pts, N.XFORM_CONTRACT, box, spec.native(), enc.data_ptr(), sel.data_ptr(),
                                               dirs.data_ptr(), cams.data_ptr(), None, S, M, fm, ddens.data_ptr(), drgb.data_ptr(),
                                               None, grads, ws.data_ptr(), ws.numel(), dtable.data_ptr(), sws.data_ptr(), words, st) == 0


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def stamps(fn, waves):
    buf = torch.zeros(waves, 64, dtype=torch.int64, device=dev)
    assert lib.nsamd_probe_set_clocks_field(buf.data_ptr()) == 0
    fn()
    torch.cuda.synchronize()
    assert lib.nsamd_probe_set_clocks_field(None) == 0
    return buf.cpu()


def report(name, t, labels):
    t = t[(t[:, 0] > 0)]
    start = t[:, 0].min()
    print(f"-- {name}: {t.shape[0]} waves; kernel span {(t[:, 63].max() - start).item()} clocks "
          f"(first wave start -> last wave end); wave lifetime mean {(t[:, 63] - t[:, 0]).double().mean().item():.0f}")
    prev = 0
    for slot, label in labels:
        ok = (t[:, slot] > 0) & (t[:, prev] > 0)
        if ok.sum() == 0:
            continue
        d = (t[ok, slot] - t[ok, prev]).double()
        print(f"   slot {prev:2d} -> {slot:2d}  {label:34s} mean {d.mean().item():9.0f}  min {d.min().item():8.0f}  max {d.max().item():8.0f}   (n={int(ok.sum())})")
        prev = slot


print(f"fwd {timed(fwd):.1f} us   bwd {timed(bwd):.1f} us   (no clock buffer set)")
W = int(os.environ.get("NSAMD_FIELD_FWD_WAVES", "16"))
f = stamps(fwd, {4: 768 * 4, 8: 512 * 8, 16: 256 * 16}[W])
labels = [(1, "stage weights + barrier")]
for it in range(4 if W == 4 else 3):
    labels += [(2 + 4 * it, f"tile {it}: inputs issued"), (3 + 4 * it, f"tile {it}: base layers"),
               (4 + 4 * it, f"tile {it}: SH + head layers"), (5 + 4 * it, f"tile {it}: outputs stored")]
labels += [(63, "end")]
report("field_mlp_fwd", f, labels)
b = stamps(bwd, 256 * 8)
labels = [(1, "stage weights + barrier")]
for it in range(6):
    k = 10 * it
    labels += [(2 + k, f"it {it}: top"), (3 + k, f"it {it}: inputs + forward"), (4 + k, f"it {it}: head2 phase"),
               (5 + k, f"it {it}: head1 phase"), (6 + k, f"it {it}: head0 phase"), (7 + k, f"it {it}: app + base1 phase")]
    labels += [
               (9 + k, f"it {it}: base0 dgrad + record emission")] if ROUTE else []
    labels += [(8 + k, f"it {it}: base0 phase + store")]
labels += [(62, "loop end"), (63, "emit partials")]
report("field_mlp_bwd", b, labels)
print(f"fwd {timed(fwd):.1f} us   bwd {timed(bwd):.1f} us   (after)")
