"""ctypes binding of libnsamd.so — the C-ABI boundary declared in include/nsamd.h.

This is the ONLY place the product touches native code. It never falls back: if the shared library is missing or a
kernel reports an error a RuntimeError is raised (the reference's tcnn seam silently falls back to torch,
field_components/encodings.py:350-352 — a `implementation="hip"` request must not).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnsamd.so")

MAX_LEVELS = 32
XFORM_NONE, XFORM_CONTRACT, XFORM_AABB = 0, 1, 2
BG_NONE, BG_LAST_SAMPLE, BG_CONSTANT = 0, 1, 2

vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int32
f32 = C.c_float


class Grid(C.Structure):
    _fields_ = [("num_levels", i32), ("log2_table_size", i32), ("scalings", f32 * MAX_LEVELS)]


class Points(C.Structure):
    _fields_ = [("positions", vp), ("origins", vp), ("directions", vp), ("t_bins", vp), ("samples_per_ray", i32)]


class Aabb(C.Structure):
    _fields_ = [("lo", f32 * 3), ("hi", f32 * 3)]


class DensityMlp(C.Structure):
    _fields_ = [("W0", vp), ("b0", vp), ("W1", vp), ("b1", vp), ("in_dim", i32), ("hidden", i32),
                ("average_init_density", f32)]


class FieldMlp(C.Structure):
    _fields_ = [("base_W0", vp), ("base_b0", vp), ("base_W1", vp), ("base_b1", vp), ("head_W0", vp), ("head_b0", vp),
                ("head_W1", vp), ("head_b1", vp), ("head_W2", vp), ("head_b2", vp), ("appearance", vp),
                ("num_images", i32), ("average_init_density", f32), ("ray_terms", vp), ("ray_inputs", vp)]


class OccGrid(C.Structure):
    _fields_ = [("binaries", vp), ("levels", i32), ("resolution", i32), ("aabb", f32 * 6), ("coarse", vp)]


class RayGrads(C.Structure):
    _fields_ = [("d_origins", vp * 4), ("d_directions", vp * 4), ("count", i32)]


class FieldMlpGrads(C.Structure):
    _fields_ = [("base_W0", vp), ("base_b0", vp), ("base_W1", vp), ("base_b1", vp), ("head_W0", vp), ("head_b0", vp),
                ("head_W1", vp), ("head_b1", vp), ("head_W2", vp), ("head_b2", vp), ("appearance", vp)]


class ProposalLevelBwd(C.Structure):
    """nsamd_proposal_level_bwd: one proposal level's gated backward chain (nsamd_proposal_levels_bwd)."""
    _fields_ = [("num_rays", i64), ("samples_per_ray", i32),
                ("t_bins", vp), ("density", vp), ("dweights", vp), ("ddensity", vp), ("gate", vp), ("ray_mask", vp),
                ("enc", vp), ("selector", vp), ("pre", vp), ("mlp", DensityMlp), ("denc", vp),
                ("dW0", vp), ("db0", vp), ("dW1", vp), ("db1", vp), ("mlp_workspace", vp), ("mlp_workspace_floats", i64),
                ("origins", vp), ("directions", vp), ("transform", C.c_int), ("aabb", Aabb), ("table", vp), ("grid", Grid),
                ("dtable", vp), ("scatter_workspace", vp), ("scatter_workspace_floats", i64)]


# name -> argtypes (restype is int unless listed in _RESTYPES). Mirrors include/nsamd.h one to one; the CPU test
# tests/test_abi.py checks that every symbol declared in the header is exported by the library and listed here.
_SIGNATURES = {
    "nsamd_hashgrid_encode_fwd": [Points, i64, C.c_int, Aabb, vp, Grid, vp, i64, i64, vp, vp],
    "nsamd_hashgrid_encode_bwd": [Points, i64, C.c_int, Aabb, vp, Grid, vp, i64, i64, vp, vp, vp, i64, vp],
    "nsamd_hashgrid_encode_bwd_set": [Points, i64, C.c_int, Aabb, vp, Grid, vp, i64, i64, vp, vp, vp, i64, vp],
    "nsamd_hashgrid_encode_bwd_rays": [Points, i64, C.c_int, Aabb, vp, Grid, vp, i64, i64, vp, vp, C.c_int, vp],
    "nsamd_hashgrid_encode_bwd_gated": [Points, i64, C.c_int, Aabb, vp, Grid, vp, i64, i64, vp, vp, i64, vp, vp, vp],
    "nsamd_hashgrid_encode_bwd_rays_gated": [Points, i64, C.c_int, Aabb, vp, Grid, vp, i64, i64, vp, vp, C.c_int, vp, vp, vp],
    "nsamd_hashgrid_encode_bwd_workspace": [Grid, i64, C.c_int],
    "nsamd_hashgrid_encode_bwd_workspace_state": [Grid, i64],
    "nsamd_hashgrid_scatter_events": [vp, vp, vp],
    "nsamd_sh4_encode": [vp, i64, vp, vp],
    "nsamd_nerf_encode": [Points, i64, vp, i32, i32, vp, vp],
    "nsamd_contract_linf": [vp, i64, vp, vp],
    "nsamd_density_mlp_fwd": [vp, vp, i64, DensityMlp, vp, vp, vp],
    "nsamd_density_field_fwd": [Points, i64, C.c_int, Aabb, vp, Grid, DensityMlp, vp, vp, vp, vp, vp],
    "nsamd_density_mlp_bwd": [vp, vp, vp, vp, i64, DensityMlp, vp, vp, vp, vp, vp, vp, i64, vp],
    "nsamd_density_mlp_bwd_gated": [vp, vp, vp, vp, i64, DensityMlp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp],
    "nsamd_field_mlp_fwd": [vp, vp, vp, vp, vp, i64, i64, FieldMlp, vp, vp, vp],
    "nsamd_field_mlp_bwd": [vp, vp, vp, vp, vp, i64, i64, FieldMlp, vp, vp, vp, FieldMlpGrads, vp, i64, vp],
    "nsamd_field_ray_terms": [vp, vp, vp, i64, FieldMlp, vp, vp, vp],
    "nsamd_field_mlp_bwd_scatter": [Points, C.c_int, Aabb, Grid, vp, vp, vp, vp, vp, i64, i64, FieldMlp, vp, vp, vp, FieldMlpGrads, vp,
                                    i64, vp, vp, i64, vp],
    "nsamd_field_mlp_bwd_scatter_phase": [Points, C.c_int, Aabb, Grid, vp, vp, vp, vp, vp, i64, i64, FieldMlp, vp, vp, vp, FieldMlpGrads,
                                          vp, i64, vp, vp, i64, C.c_int, vp],
    "nsamd_field_mlp_bwd_scatter_workspace": [Grid, i64, C.POINTER(C.c_int64)],
    "nsamd_field_mlp_bwd_reserve_cus": [C.c_int],
    "nsamd_proposal_levels_bwd": [C.POINTER(ProposalLevelBwd), i32, i32, vp],
    "nsamd_field_mlp_bwd_phase": [vp, vp, vp, vp, vp, i64, i64, FieldMlp, vp, vp, vp, FieldMlpGrads, vp, i64, C.c_int, vp],
    "nsamd_linear_fwd": [vp, vp, vp, i64, i32, i32, C.c_int, vp, vp],
    "nsamd_linear_bwd": [vp, vp, vp, vp, i64, i32, i32, C.c_int, vp, vp, vp, vp],
    "nsamd_piecewise_bins": [vp, vp, vp, vp, i32, i64, i32, C.c_int, vp, vp, vp],
    "nsamd_weights_fwd": [vp, vp, i64, i32, vp, vp],
    "nsamd_weights_bwd": [vp, vp, vp, i64, i32, vp, vp],
    "nsamd_weights_bwd_gate": [vp, vp, vp, i64, i32, vp, vp, vp, i32, vp],
    "nsamd_pdf_resample": [vp, vp, i32, vp, vp, vp, vp, f32, vp, f32, f32, f32, C.c_int, i32, i32, i64, i32, vp, vp, vp, vp],
    "nsamd_proposal_resample": [vp, vp, vp, i32, vp, vp, vp, vp, f32, vp, f32, f32, f32, C.c_int, i64, i32, vp, vp, vp, vp, vp],
    "nsamd_composite_fwd": [vp, vp, vp, i64, i32, C.c_int, C.POINTER(f32), C.c_int, vp, vp, vp, vp, vp, vp, vp],
    "nsamd_render_train": [vp, vp, vp, i64, i32, C.c_int, C.POINTER(f32), vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "nsamd_render_train_bwd": [vp, vp, vp, vp, i64, i32, C.c_int, C.POINTER(f32), vp, vp, vp, vp, vp, vp],
    "nsamd_composite_bwd": [vp, vp, vp, i64, i32, C.c_int, C.POINTER(f32), vp, vp, vp, vp, vp, vp, vp, vp],
    "nsamd_distance_gradient_scale": [vp, i64, i32, vp, vp, vp],
    "nsamd_mse_loss": [vp, vp, i64, f32, vp, vp, vp],
    "nsamd_interlevel_loss": [vp, vp, i32, vp, vp, i32, i64, f32, vp, vp, vp],
    "nsamd_distortion_loss": [vp, vp, i32, i64, f32, vp, vp, vp],
    "nsamd_proposal_losses": [vp, vp, i32, i32, vp, vp, vp, i64, f32, f32, vp, vp, vp, vp, vp],
    "nsamd_train_loss_values": [vp, vp, i32, vp, i64, i32, f32, f32, vp, vp],
    "nsamd_occgrid_march_count": [vp, vp, vp, vp, i64, f32, f32, OccGrid, f32, f32, vp, vp, vp],
    "nsamd_occgrid_march_write": [vp, vp, vp, vp, i64, f32, f32, OccGrid, f32, f32, vp, vp, vp, vp, vp, vp],
    "nsamd_occgrid_march_count_stash": [vp, vp, vp, vp, i64, f32, f32, OccGrid, f32, f32, vp, vp, vp, i32, vp],
    "nsamd_occgrid_march_write_stashed": [vp, vp, vp, vp, i64, f32, f32, OccGrid, f32, f32, vp, vp, vp, i32, vp, vp, vp, vp],
    "nsamd_occgrid_coarse_words": [i32, i32],
    "nsamd_occgrid_cell_positions": [vp, i64, OccGrid, vp, vp, vp],
    "nsamd_occgrid_update": [vp, i64, vp, vp, i64, f32, vp, vp],
    "nsamd_occgrid_binarise": [vp, i32, i32, f32, vp, vp, vp, vp, vp],
    "nsamd_packed_info": [vp, i64, vp, vp, vp],
    "nsamd_packed_weights_fwd": [vp, vp, vp, vp, i64, vp, vp, vp],
    "nsamd_packed_weights_bwd": [vp, vp, vp, vp, vp, i64, vp, vp],
    "nsamd_packed_visibility": [vp, vp, vp, vp, i64, f32, f32, vp, vp, vp],
    "nsamd_packed_compact": [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp],
    "nsamd_packed_composite_fwd": [vp, vp, vp, vp, vp, i64, C.c_int, C.POINTER(f32), C.c_int, vp, vp, vp, vp],
    "nsamd_packed_composite_bwd": [vp, vp, vp, i64, C.c_int, C.POINTER(f32), vp, vp, vp, vp, vp],
    "nsamd_packed_positions": [vp, vp, vp, vp, vp, i64, vp, vp],
    "nsamd_raygen_pinhole": [vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp],
    "nsamd_raygen_pinhole_grid": [vp, f32, f32, f32, f32, i32, i64, i64, i64, vp, vp, vp, vp],
    "nsamd_rows_gather": [vp, vp, i64, i32, vp, vp],
    "nsamd_rows_scatter": [vp, vp, i64, i32, vp, vp],
    "nsamd_select_batch": [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "nsamd_select_bins": [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, C.c_int, vp, vp, vp],
    "nsamd_step_prologue": [vp, vp, i32, vp, vp, i64, vp, i64, C.c_uint64, vp],
    "nsamd_camera_apply": [vp, i32, i32, vp, vp, vp, i64, vp, vp, vp],
    "nsamd_camera_backward": [vp, i32, i32, vp, vp, i64, RayGrads, f32, f32, vp, vp, vp],
    "nsamd_adam_step": [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, i32, f32, vp, vp],
    "nsamd_version": [],
    "nsamd_status_string": [C.c_int],
    "nsamd_device_info": [C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32],
    "nsamd_probe_mfma16": [vp, vp, vp, vp],
    "nsamd_probe_mfma_bf16": [vp, vp, vp, vp],
}
_RESTYPES = {"nsamd_version": C.c_char_p, "nsamd_status_string": C.c_char_p,
             "nsamd_hashgrid_encode_bwd_workspace": C.c_int64, "nsamd_hashgrid_encode_bwd_workspace_state": C.c_int64,
             "nsamd_field_mlp_bwd_scatter_workspace": C.c_int64,
             "nsamd_occgrid_coarse_words": C.c_int64}

_lib = None
ERR_UNSUPPORTED = -2  # nsamd_status NSAMD_ERR_UNSUPPORTED

# Optional live kernel timing (bench.py's `roofline` leg): when PROFILE is a dict, every launch through the binding is
# bracketed by a pair of HIP events recorded on torch's current stream — the stream the kernels are enqueued on.
PROFILE: Optional[dict] = None


class _Entry:
    """One C-ABI entry point; transparently records (start, end) events per call when profiling is enabled."""

    __slots__ = ("fn", "name")

    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *args):
        prof = PROFILE
        if prof is None:
            return self.fn(*args)
        key = self.name
        if self.name == "nsamd_density_field_fwd":
            key = f"{self.name}[M={args[1]}]"
        elif self.name.startswith("nsamd_hashgrid_encode"):
            key = f"{self.name}[L={args[5].num_levels},M={args[1]}]"
        elif self.name.startswith("nsamd_density_mlp"):
            key = f"{self.name}[M={args[2] if self.name.endswith('fwd') else args[4]}]"
        elif self.name == "nsamd_field_mlp_bwd_scatter_phase":
            key = f"{self.name}[{ {1: 'gradients+records', 2: 'dw_reduce', 4: 'apply'}.get(args[21], args[21]) }]"
        elif self.name == "nsamd_adam_step":
            key = f"{self.name}[n={args[4]}]"
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = self.fn(*args)
        e1.record()
        prof.setdefault(key, []).append((e0, e1))
        return r


class _Lib:
    pass


def load():
    """Load libnsamd.so (once). Raises RuntimeError when it has not been built — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("NSAMD_LIB", LIB_PATH)  # (a probe / instrumented build of the same sources, scripts/probe_*)
    if not os.path.exists(path):
        raise RuntimeError(
            f"nerfstudio_amd: native library not found at {path}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C nerfstudio_amd/csrc`). "
            "There is no CPU/torch fallback for implementation='hip'."
        )
    cdll = C.CDLL(path)
    lib = _Lib()
    lib.cdll = cdll
    older_ok = "NSAMD_LIB" in os.environ and os.environ.get("NSAMD_LIB_OLDER_ABI") == "1"
    for name, argtypes in _SIGNATURES.items():
        try:
            fn = getattr(cdll, name)  # AttributeError here = header / library mismatch
        except AttributeError:
            if not older_ok:
                raise
            # same-box A/B against an OLDER build (NSAMD_LIB=... NSAMD_LIB_OLDER_ABI=1): an entry point it lacks fails when called
            def missing(*_a, _n=name):
                raise RuntimeError(f"nsamd: {_n} is not exported by {path} (an older build loaded for A/B)")

            setattr(lib, name, missing)
            continue
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
        setattr(lib, name, _Entry(fn, name) if fn.restype is C.c_int and name not in ("nsamd_device_info",) else fn)
    _lib = lib
    return lib


def profile_summary(prof: dict) -> dict:
    """{key: (calls, total_ms, mean_ms)} from recorded event pairs (call after a device synchronise)."""
    out = {}
    for key, pairs in prof.items():
        ms = [a.elapsed_time(b) for a, b in pairs]
        if os.environ.get("NSAMD_ROOFLINE_SAMPLES") == "1" and "field_mlp_bwd" in key:  # diagnostics: every launch's time
            import sys

            print(f"[samples] {key}: " + " ".join(f"{m:.4f}" for m in ms), file=sys.stderr)
        out[key] = (len(ms), float(sum(ms)), float(sum(ms) / max(1, len(ms))))
    return out


def status_string(status: int) -> str:
    return load().nsamd_status_string(status).decode()


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"nsamd: {what} failed with status {status}: {status_string(status)}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous; callers keep it alive."""
    if t is None:
        return None
    assert t.is_contiguous(), "nsamd kernels take dense row-major tensors"
    return t.data_ptr()


# torch.cuda.current_stream() / `with torch.cuda.stream(s)` with no device argument resolve "the current device" through
# torch._utils._get_available_device_type -> torch.cuda.is_available() -> a device-count query of the driver: ~20 us per call on
# this ROCm build, 14 - 16 of them per eagerly launched iteration = a third of the host's share of an eager / data-parallel step
# (profiles/r06_s30_eager_host_cprofile.txt: 3170 x _cuda_getDeviceCount in 225 iterations). The helpers below name the device.
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (private torch entry points: the public route is the fallback)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """The raw hipStream_t of torch's current stream (kernels must run on it, SURVEY.md §8b threading)."""
    if _RAW_STREAM is None or _GET_DEVICE is None:
        return torch.cuda.current_stream().cuda_stream
    return _RAW_STREAM(_GET_DEVICE())


def current_stream() -> "torch.cuda.Stream":
    """torch.cuda.current_stream() of the current device, without the device-count query."""
    if _GET_DEVICE is None:
        return torch.cuda.current_stream()
    return torch.cuda.current_stream(_GET_DEVICE())


class on_stream:
    """`with on_stream(s):` — torch.cuda.stream(s) for a stream of the CURRENT device, without the two device-count queries."""

    __slots__ = ("s", "prev")

    def __init__(self, s) -> None:
        self.s = s

    def __enter__(self):
        self.prev = current_stream()
        torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


def require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "nerfstudio_amd: implementation='hip' runs on an MI355X only; got a CPU tensor "
                "(there is no CPU fallback — use the reference's implementation='torch' for CPU runs)."
            )


def make_grid(num_levels: int, log2_table_size: int, scalings) -> Grid:
    if num_levels > MAX_LEVELS:
        raise ValueError(f"at most {MAX_LEVELS} hash levels are supported, got {num_levels}")
    g = Grid()
    g.num_levels = int(num_levels)
    g.log2_table_size = int(log2_table_size)
    for i, s in enumerate(scalings):
        g.scalings[i] = float(s)
    return g


def make_points(positions=None, origins=None, directions=None, t_bins=None, samples_per_ray: int = 0) -> Points:
    p = Points()
    p.positions = ptr(positions)
    p.origins = ptr(origins)
    p.directions = ptr(directions)
    p.t_bins = ptr(t_bins)
    p.samples_per_ray = int(samples_per_ray)
    return p


def make_aabb(aabb: Optional[torch.Tensor]) -> Aabb:
    a = Aabb()
    if aabb is not None:
        vals = aabb.detach().cpu().reshape(2, 3).tolist()
        for i in range(3):
            a.lo[i] = vals[0][i]
            a.hi[i] = vals[1][i]
    return a


def device_info() -> dict:
    lib = load()
    cus, wf, lds = i32(), i32(), i32()
    name = C.create_string_buffer(64)
    check(lib.nsamd_device_info(C.byref(cus), C.byref(wf), C.byref(lds), name, 64), "nsamd_device_info")
    return {"num_cus": cus.value, "wavefront_size": wf.value, "lds_bytes_per_cu": lds.value, "arch": name.value.decode()}
