"""Renderers (reference: nerfstudio/model_components/renderers.py — RGBRenderer :60-232, AccumulationRenderer
:289-317, DepthRenderer :320-385). One HIP kernel, one wavefront per ray (csrc/render.hip)."""
from typing import Literal, Optional, Union

import torch
from torch import Tensor, nn

from .. import functional as F
from ..cameras.rays import RaySamples, t_bins_of

BackgroundColor = Union[Literal["random", "last_sample", "black", "white"], Tensor]


def _packed(ray_indices, num_rays) -> bool:
    """Packed samples from the VolumetricSampler (renderers.py:93, 310, 369): `[n, ...]` samples + the ray of each."""
    return ray_indices is not None and num_rays is not None


def _packed_info(ray_indices: Tensor, num_rays: int) -> Tensor:
    """nerfacc.pack_info: `[num_rays, 2]` (start, count); samples of a ray are contiguous, rays in increasing order."""
    cached = getattr(ray_indices, "_nsamd_packed_info", None)  # VolumetricSampler leaves the marcher's own (start, count)
    if cached is not None and cached.shape[0] == num_rays:      # rows here: no bincount + prefix + host sync per renderer
        return cached
    counts = torch.bincount(ray_indices, minlength=num_rays).to(torch.int32)
    return F.packed_info_from_counts(counts)[0]


class RGBRenderer(nn.Module):
    """Standard volumetric rendering of colour."""

    def __init__(self, background_color: BackgroundColor = "random") -> None:
        super().__init__()
        self.background_color: BackgroundColor = background_color

    @classmethod
    def combine_rgb(cls, rgb: Tensor, weights: Tensor, background_color: BackgroundColor = "random",
                    ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None) -> Tensor:
        """Composite samples along the ray (renderers.py:72-119): sum_i w_i rgb_i, plus background * (1 - sum_i w_i)
        for "last_sample" / "white" / "black" / an RGB tensor; "random" adds nothing (as if the background were black —
        the random colour is blended in blend_background_for_loss_computation). No nan_to_num / clamp (that is
        forward()'s eval branch). rgb `[*bs,S,3]`, weights `[*bs,S,1]` -> `[*bs,3]`."""
        if _packed(ray_indices, num_rays):
            if isinstance(background_color, str) and background_color == "last_sample":
                raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")
            return F.packed_composite(rgb, weights[..., 0], ray_indices, _packed_info(ray_indices, num_rays), None, None,
                                      background_color)[0]
        shape, s = rgb.shape[:-2], rgb.shape[-2]
        out, _, _ = F.composite(rgb.reshape(-1, s, 3), weights.reshape(-1, s), None, background_color, expected_depth=False)
        return out.view(*shape, 3)

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, background_color: Optional[BackgroundColor] = None) -> Tensor:
        """rgb `[*bs,S,3]`, weights `[*bs,S,1]` -> `[*bs,3]` (renderers.py:201-232); packed: rgb `[n,3]`, weights `[n,1]`."""
        if background_color is None:
            background_color = self.background_color
        if _packed(ray_indices, num_rays):
            if self.training:
                return self.combine_rgb(rgb, weights, background_color, ray_indices, num_rays)
            with torch.no_grad():  # eval: nan_to_num on the samples, clamp the result (renderers.py:225-231)
                return F.packed_composite(rgb, weights[..., 0], ray_indices, _packed_info(ray_indices, num_rays), None, None,
                                          background_color, eval_mode=True)[0]
        shape = rgb.shape[:-2]
        s = rgb.shape[-2]
        rgb2, w2 = rgb.reshape(-1, s, 3), weights.reshape(-1, s)
        if self.training:
            return self.combine_rgb(rgb, weights, background_color=background_color)
        else:
            # eval: nan_to_num on the samples, clamp the result (renderers.py:225-231); no gradient needed
            dummy_t = torch.zeros((w2.shape[0], s + 1), device=w2.device)
            out = F.composite_eval(rgb2, w2, dummy_t, background_color)[0]
        return out.view(*shape, 3)

    @classmethod
    def get_background_color(cls, background_color, shape, device):
        named = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0)}
        if isinstance(background_color, str):
            background_color = torch.tensor(named[background_color])
        return background_color.expand(shape).to(device)

    def blend_background(self, image: Tensor, background_color: Optional[BackgroundColor] = None) -> Tensor:
        """RGBA ground truth -> RGB over the background (renderers.py:150-173); RGB passes through."""
        if image.size(-1) < 4:
            return image
        rgb, opacity = image[..., :3], image[..., 3:]
        if background_color is None:
            background_color = self.background_color
            if background_color in {"last_sample", "random"}:
                background_color = "black"
        bg = self.get_background_color(background_color, rgb.shape, rgb.device)
        return rgb * opacity + bg * (1 - opacity)

    def blend_background_for_loss_computation(self, pred_image: Tensor, pred_accumulation: Tensor, gt_image: Tensor):
        """renderers.py:175-199."""
        background_color = self.background_color
        if background_color == "last_sample":
            background_color = "black"
        elif background_color == "random":
            background_color = torch.rand_like(pred_image)
            pred_image = pred_image + background_color * (1.0 - pred_accumulation)
        gt_image = self.blend_background(gt_image, background_color=background_color)
        return pred_image, gt_image


class AccumulationRenderer(nn.Module):
    """Accumulated weight along a ray."""

    @classmethod
    def forward(cls, weights: Tensor, ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None) -> Tensor:
        if _packed(ray_indices, num_rays):  # accumulate_along_rays(weights, None) (renderers.py:310-314)
            zero = torch.zeros((weights.shape[0], 3), device=weights.device)
            return F.packed_composite(zero, weights[..., 0], ray_indices, _packed_info(ray_indices, num_rays))[1][:, None]
        shape = weights.shape[:-2]
        s = weights.shape[-2]
        w2 = weights.reshape(-1, s)
        if w2.requires_grad:
            rgb0 = torch.zeros((w2.shape[0], s, 3), device=w2.device)
            acc = F.composite(rgb0, w2, None, "random", expected_depth=False)[1]
        else:
            acc = F.accumulation(w2)
        return acc.view(*shape, 1)


class DepthRenderer(nn.Module):
    """Depth along a ray: "median" = first sample where the running weight reaches 0.5; "expected" = weighted mean of
    the sample midpoints, clipped to the batch-wide midpoint range (renderers.py:320-385)."""

    def __init__(self, method: Literal["median", "expected"] = "median") -> None:
        super().__init__()
        if method not in ("median", "expected"):
            raise NotImplementedError(f"Method {method} not implemented")
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None) -> Tensor:
        if _packed(ray_indices, num_rays):
            if self.method != "expected":
                raise NotImplementedError("packed samples support DepthRenderer('expected') (renderers.py:365-383)")
            starts, ends = ray_samples.frustums.starts[..., 0].contiguous(), ray_samples.frustums.ends[..., 0].contiguous()
            zero = torch.zeros((weights.shape[0], 3), device=weights.device)
            depth = F.packed_composite(zero, weights[..., 0], ray_indices, _packed_info(ray_indices, num_rays), starts, ends)[2]
            steps = (starts + ends) / 2
            return torch.clip(depth, steps.min(), steps.max())[:, None]
        shape = weights.shape[:-2]
        s = weights.shape[-2]
        w2 = weights.reshape(-1, s)
        t_bins = t_bins_of(ray_samples).reshape(-1, s + 1)
        if self.method == "median":
            return F.depth_median(w2, t_bins).view(*shape, 1)
        rgb0 = torch.zeros((w2.shape[0], s, 3), device=w2.device)
        depth = F.composite(rgb0, w2, t_bins, "random", expected_depth=True)[2]
        return depth.view(*shape, 1)


class NormalsRenderer(nn.Module):
    """Weighted sum of the per-sample normals along a ray (renderers.py:429-449), normalised with the reference's
    `safe_normalize` (utils/math.py:214-227: v / (|v| + 1e-10))."""

    @classmethod
    def forward(cls, normals: Tensor, weights: Tensor, normalize: bool = True) -> Tensor:
        n = torch.sum(weights * normals, dim=-2)
        if normalize:
            n = n / (torch.norm(n, dim=-1, keepdim=True) + 1e-10)
        return n
