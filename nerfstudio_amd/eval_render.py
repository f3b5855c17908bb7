"""Full-image eval render of the nerfacto path as a device-side chunk loop.

Reference: `Model.get_outputs_for_camera_ray_bundle` (models/base_model.py:178-205) slices the camera's ray bundle into
chunks of `eval_num_rays_per_chunk` rays in Python, runs `forward` on each — ~60 module calls and as many tensor
allocations per chunk — and `torch.cat`s every output. Here one chunk is ONE explicit kernel schedule over static buffers
(the forward half of train_step.NerfactoTrainStep in eval mode), captured once per chunk size in a hipGraph; a frame is

    for each chunk:  copy the chunk's rays into the static input buffers -> replay -> copy the outputs into their
                     rows of the preallocated image buffers

i.e. no per-chunk Python graph of modules, no allocation, no `torch.cat`. Eval semantics as the reference's
(models/nerfacto.py:298-348 with `self.training == False`): near plane reset to 0 (scene_colliders.py:186-191), no jitter —
bin centres in the initial sampler, the fixed 1/(2 nb) offset in the PDF resampling (ray_samplers.py:104-111, 323-327) —,
the mean (or zero) appearance embedding for every sample (fields/nerfacto_field.py:253-261), nan_to_num on the samples and
a clamp of the composited colour (renderers.py:225-231), expected depth clipped to the CHUNK's min / max sample midpoint
(the reference clips per forward call, i.e. per chunk, renderers.py:380-383). A last, shorter chunk is padded with copies
of its last ray (which leaves that min / max untouched) and only its valid rows are copied out.
Same kernels, same order as the module path: the outputs are equal bit for bit (tests/test_gpu_kernels.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import _native as N
from . import functional as F  # noqa: F401
from .utils import profiler


class EvalRenderer:
    def __init__(self, model, chunk: Optional[int] = None, use_graph: bool = True) -> None:
        from .train_step import NerfactoTrainStep

        self.model = model
        self.chunk = int(chunk or model.config.eval_num_rays_per_chunk)
        dev = next(model.parameters()).device
        N.require_cuda(next(model.parameters()))
        self.step = NerfactoTrainStep(model, self.chunk, dev, compute_depths=True, forward_only=True)
        self.step.nears.zero_()  # NearFarCollider at inference (reset_near_plane)
        fld = model.field
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        self.app_const = torch.zeros(emb.shape[1], device=dev) if emb is not None else None
        self.use_graph = use_graph
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._captured_addresses = ()
        if self.step.bg_mode == 3:  # "random": eval composites without a background (renderers.py:112-115 adds none)
            self.bg_mode, self.bg_vals = N.BG_NONE, None
        else:
            self.bg_mode, self.bg_vals = self.step.bg_mode, self.step.bg_vals

    # ---- one chunk ------------------------------------------------------------------------------------------------------
    def _refresh_constants(self) -> None:
        """Per-frame values that live in device memory (so that a captured graph sees the current ones): the anneal exponent
        and the appearance row of eval mode."""
        m, s = self.model, self.step
        s.anneal_dev.fill_(float(m.proposal_sampler._anneal))
        fld = m.field
        if self.app_const is not None:
            emb = fld.embedding_appearance.embedding.weight
            with torch.no_grad():
                if fld.use_average_appearance_embedding:
                    torch.mean(emb, dim=0, out=self.app_const)
                else:
                    self.app_const.zero_()

    def _launch_chunk(self) -> None:
        """The kernel schedule of one chunk (all arguments are static buffers: capturable)."""
        s, lib, st, n = self.step, N.load(), N.stream(), self.chunk
        ck = N.check
        S0 = s.counts[0]
        ck(lib.nsamd_piecewise_bins(N.ptr(s.nears), N.ptr(s.fars), N.ptr(s.edges), None, 0, n, S0, s.spacing,
                                    N.ptr(s.s_bins[0]), N.ptr(s.t_bins[0]), st), "piecewise_bins")
        for lvl in range(s.n_prop):
            net = s.props[lvl]
            S, m = s.counts[lvl], n * s.counts[lvl]
            W0, b0, W1, b1 = net.mlp_base[1].param_tensors()
            dm = N.DensityMlp(N.ptr(W0), N.ptr(b0), N.ptr(W1), N.ptr(b1), W0.shape[1], W0.shape[0],
                              float(net.average_init_density))
            fused = lib.nsamd_density_field_fwd(s._points(lvl), m, net._transform, net._box, N.ptr(net.encoding.hash_table),
                                                net.encoding.spec.native(), dm, None, None, N.ptr(s.p_dens[lvl]), None, st)
            if fused == N.ERR_UNSUPPORTED:
                s.ensure_proposal_features(lvl)
                ck(lib.nsamd_hashgrid_encode_fwd(s._points(lvl), m, net._transform, net._box, N.ptr(net.encoding.hash_table),
                                                 net.encoding.spec.native(), N.ptr(s.p_enc[lvl]), 1, m, N.ptr(s.p_sel[lvl]), st),
                   "hashgrid_encode_fwd")
                ck(lib.nsamd_density_mlp_fwd(N.ptr(s.p_enc[lvl]), N.ptr(s.p_sel[lvl]), m, dm, N.ptr(s.p_dens[lvl]), None, st),
                   "density_mlp_fwd")
            else:
                ck(fused, "density_field_fwd")
            S2 = s.counts[lvl + 1]
            ck(lib.nsamd_proposal_resample(N.ptr(s.t_bins[lvl]), N.ptr(s.s_bins[lvl]), N.ptr(s.p_dens[lvl]), S,
                                           N.ptr(s.u_base[lvl + 1]), None, N.ptr(s.nears), N.ptr(s.fars), 1.0,
                                           N.ptr(s.anneal_dev), 0.01, 1e-5, 1.0 / (2 * (S2 + 1)), s.spacing, n, S2,
                                           N.ptr(s.weights[lvl]), N.ptr(s.depth_med[lvl]), N.ptr(s.s_bins[lvl + 1]),
                                           N.ptr(s.t_bins[lvl + 1]), st), "proposal_resample")
        L = s.n_prop
        S, mm = s.counts[L], s.m_main
        fld = self.model.field
        enc = fld.mlp_base.encoding
        params = [*fld.mlp_base.mlp.param_tensors(), *fld.mlp_head.param_tensors()]
        emb = fld.embedding_appearance.embedding.weight if fld.embedding_appearance is not None else None
        fm = N.FieldMlp(*(N.ptr(p) for p in params), N.ptr(emb), emb.shape[0] if emb is not None else 0,
                        float(fld.average_init_density))
        if s.ray_terms_on:  # head layer 0's per-ray share once per ray (include/nsamd.h, nsamd_field_mlp.ray_terms)
            ck(lib.nsamd_field_ray_terms(N.ptr(s.directions), None, N.ptr(self.app_const), n, fm, N.ptr(s.ray_terms), None, st),
               "field_ray_terms")
            fm.ray_terms = N.ptr(s.ray_terms)
        ck(lib.nsamd_hashgrid_encode_fwd(s._points(L), mm, fld._transform, fld._box, N.ptr(enc.hash_table), enc.spec.native(),
                                         N.ptr(s.f_enc), 1, mm, N.ptr(s.f_sel), st), "hashgrid_encode_fwd")
        ck(lib.nsamd_field_mlp_fwd(N.ptr(s.f_enc), N.ptr(s.f_sel), N.ptr(s.directions), None, N.ptr(self.app_const), S, mm, fm,
                                   N.ptr(s.f_dens), N.ptr(s.f_rgb), st), "field_mlp_fwd")
        ck(lib.nsamd_weights_fwd(N.ptr(s.t_bins[L]), N.ptr(s.f_dens), n, S, N.ptr(s.weights[L]), st), "weights_fwd")
        ck(lib.nsamd_composite_fwd(N.ptr(s.f_rgb), N.ptr(s.weights[L]), N.ptr(s.t_bins[L]), n, S, self.bg_mode, self.bg_vals, 1,
                                   N.ptr(s.rgb), N.ptr(s.acc), N.ptr(s.depth_exp), N.ptr(s.depth_med[L]), None,
                                   N.ptr(s.minmax_ws), st), "composite_fwd")

    def _addresses(self):
        """Everything the captured launches read through raw pointers that this object does not own: the model's parameters
        and buffers (a ParamArena built later re-homes them, `.to()` moves them, an embedding may be swapped)."""
        m = self.model
        return tuple(t.data_ptr() for t in list(m.parameters()) + list(m.buffers()))

    def _run_chunk(self) -> None:
        if not self.use_graph:
            self._launch_chunk()
            return
        if self.graph is not None and self._addresses() != self._captured_addresses:
            self.graph = None  # the graph holds stale pointers: capture again
        if self.graph is None:
            # the warm-up launch below may allocate (a proposal network the fused density kernel does not take): outside the capture
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # lazy kernel attributes / first-use work outside the capture
                self._launch_chunk()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch_chunk()
            self.graph = g
            self._captured_addresses = self._addresses()
        self.graph.replay()

    # ---- a frame ----------------------------------------------------------------------------------------------------------
    @profiler.time_function
    @torch.no_grad()
    def render(self, camera_ray_bundle) -> Dict[str, Tensor]:
        """-> the reference's output dict for a camera (`rgb`, `accumulation`, `depth`, `expected_depth`, `prop_depth_i`),
        each `[*image_shape, C]`."""
        s, n = self.step, self.chunk
        image_shape = camera_ray_bundle.origins.shape[:-1]
        o = camera_ray_bundle.origins.reshape(-1, 3)
        d = camera_ray_bundle.directions.reshape(-1, 3)
        total = o.shape[0]
        dev = o.device
        out = {"rgb": torch.empty((total, 3), device=dev), "accumulation": torch.empty((total, 1), device=dev),
               "depth": torch.empty((total, 1), device=dev), "expected_depth": torch.empty((total, 1), device=dev)}
        for i in range(s.n_prop):
            out[f"prop_depth_{i}"] = torch.empty((total, 1), device=dev)
        self._refresh_constants()
        for a in range(0, total, n):
            b = min(a + n, total)
            k = b - a
            s.origins[:k].copy_(o[a:b])
            s.directions[:k].copy_(d[a:b])
            if k < n:  # pad with copies of the last ray: the chunk's depth clip range is unchanged
                s.origins[k:].copy_(o[b - 1:b].expand(n - k, 3))
                s.directions[k:].copy_(d[b - 1:b].expand(n - k, 3))
            self._run_chunk()
            out["rgb"][a:b].copy_(s.rgb[:k])
            out["accumulation"][a:b, 0].copy_(s.acc[:k])
            out["expected_depth"][a:b, 0].copy_(s.depth_exp[:k])
            out["depth"][a:b, 0].copy_(s.depth_med[-1][:k])
            for i in range(s.n_prop):
                out[f"prop_depth_{i}"][a:b, 0].copy_(s.depth_med[i][:k])
        return {k_: v.view(*image_shape, -1) for k_, v in out.items()}

    @profiler.time_function
    @torch.no_grad()
    def render_camera(self, c2w: Tensor, fx: float, fy: float, cx: float, cy: float, height: int, width: int) -> Dict[str, Tensor]:
        """`render` for ONE pinhole camera WITHOUT a ray bundle: each chunk's rays are generated straight into the schedule's
        static input buffers (nsamd_raygen_pinhole_grid: pixel first + i of the implicit row-major grid, the arithmetic of
        Cameras.generate_rays(camera_indices=0, keep_shape=True), cameras/cameras.py:321-503 perspective branch) — what
        Model.get_outputs_for_camera (models/base_model.py:166-175) builds as an [H, W] bundle with ~40 torch launches over
        full-image tensors and then slices. Same bits as `render` on that bundle. c2w: [3, 4] on the model's device."""
        s, n = self.step, self.chunk
        total = int(height) * int(width)
        dev = s.origins.device
        c2w = c2w.reshape(3, 4).to(device=dev, dtype=torch.float32).contiguous()
        out = {"rgb": torch.empty((total, 3), device=dev), "accumulation": torch.empty((total, 1), device=dev),
               "depth": torch.empty((total, 1), device=dev), "expected_depth": torch.empty((total, 1), device=dev)}
        for i in range(s.n_prop):
            out[f"prop_depth_{i}"] = torch.empty((total, 1), device=dev)
        self._refresh_constants()
        lib = N.load()
        for a in range(0, total, n):
            k = min(a + n, total) - a
            N.check(lib.nsamd_raygen_pinhole_grid(N.ptr(c2w), float(fx), float(fy), float(cx), float(cy), int(width), a, k, n,
                                                  N.ptr(s.origins), N.ptr(s.directions), None, N.stream()), "raygen_pinhole_grid")
            self._run_chunk()
            b = a + k
            out["rgb"][a:b].copy_(s.rgb[:k])
            out["accumulation"][a:b, 0].copy_(s.acc[:k])
            out["expected_depth"][a:b, 0].copy_(s.depth_exp[:k])
            out["depth"][a:b, 0].copy_(s.depth_med[-1][:k])
            for i in range(s.n_prop):
                out[f"prop_depth_{i}"][a:b, 0].copy_(s.depth_med[i][:k])
        return {k_: v.view(int(height), int(width), -1) for k_, v in out.items()}


def pinhole_camera_args(camera):
    """(c2w [3,4], fx, fy, cx, cy, H, W) of a single undistorted perspective `Cameras` object (cameras/cameras.py), or None when the
    camera needs the general ray generator (fisheye / equirectangular / orthographic types, distortion parameters, several
    cameras, per-camera metadata the field consumes)."""
    try:
        c2w = camera.camera_to_worlds
        if c2w.dim() == 3:
            if c2w.shape[0] != 1:
                return None
            c2w = c2w[0]
        ctype = getattr(camera, "camera_type", None)
        if ctype is not None and int(torch.as_tensor(ctype).reshape(-1)[0]) != 1:  # CameraType.PERSPECTIVE
            return None
        dist = getattr(camera, "distortion_params", None)
        if dist is not None and bool(torch.any(torch.as_tensor(dist) != 0)):
            return None
        one = lambda t: float(torch.as_tensor(t).reshape(-1)[0])  # noqa: E731
        h, w = int(one(camera.height)), int(one(camera.width))
        if h <= 0 or w <= 0 or torch.as_tensor(camera.fx).numel() != 1:
            return None
        return c2w[:3, :4], one(camera.fx), one(camera.fy), one(camera.cx), one(camera.cy), h, w
    except (AttributeError, TypeError, ValueError, IndexError):
        return None


def supported(model) -> Optional[str]:
    """None, or why this model's eval render has to stay on the module path."""
    cfg = model.config
    if getattr(cfg, "predict_normals", False):
        return "predict_normals"
    if not all(hasattr(model, a) for a in ("proposal_networks", "field", "proposal_sampler")):
        return "not a nerfacto model"
    return None
