"""Ray generator (reference: nerfstudio/model_components/ray_generators.py:26-56)."""
import torch
from torch import Tensor, nn

from .. import functional as F
from ..cameras.rays import RayBundle


class RayGenerator(nn.Module):
    """(camera, row, col) pixel indices -> RayBundle, as one HIP kernel (csrc/misc.hip) for pinhole cameras.

    `cameras` is anything exposing nerfstudio's `Cameras` tensors: `camera_to_worlds [C,3,4]`, `fx, fy, cx, cy [C]`
    or `[C,1]` (cameras/cameras.py:88-170). Non-perspective lenses and lens distortion stay on the reference's
    torch path and are rejected here.
    """

    def __init__(self, cameras) -> None:
        super().__init__()
        self.cameras = cameras
        ctype = getattr(cameras, "camera_type", None)
        if ctype is not None and torch.is_tensor(ctype) and bool((ctype != 1).any()):  # CameraType.PERSPECTIVE == 1
            raise ValueError("the hip RayGenerator handles perspective cameras only")
        dist = getattr(cameras, "distortion_params", None)
        if dist is not None and bool((dist != 0).any()):
            raise ValueError("the hip RayGenerator does not undistort; pass undistorted pinhole cameras")
        self.register_buffer("c2w", torch.as_tensor(cameras.camera_to_worlds).float().reshape(-1, 3, 4).clone(), persistent=False)
        for name in ("fx", "fy", "cx", "cy"):
            self.register_buffer(name, torch.as_tensor(getattr(cameras, name)).float().reshape(-1).clone(), persistent=False)

    def forward(self, ray_indices: Tensor) -> RayBundle:
        """ray_indices `[num_rays,3]` = (camera, row, col) -> RayBundle (pixel centres at +0.5)."""
        o, d, pa, dn = F.raygen_pinhole(ray_indices, self.c2w, self.fx, self.fy, self.cx, self.cy)
        return RayBundle(
            origins=o,
            directions=d,
            pixel_area=pa,
            camera_indices=ray_indices[:, 0:1].to(torch.int64),
            metadata={"directions_norm": dn},
        )
