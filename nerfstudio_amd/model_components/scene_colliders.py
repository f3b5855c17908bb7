"""Scene colliders: near / far values for the rays of a bundle (reference: nerfstudio/model_components/scene_colliders.py:28-46,
169-191). No kernel is involved — two [N,1] fills on any bundle with `origins`, `nears`, `fars` (this package's RayBundle or the
reference's); the classes keep the reference's names, constructor arguments and eval-mode behaviour."""
import torch
from torch import nn


class SceneCollider(nn.Module):
    """forward() leaves a bundle that already carries nears and fars alone, else asks the subclass for them."""

    def __init__(self, **kwargs) -> None:
        self.kwargs = kwargs
        super().__init__()

    def set_nears_and_fars(self, ray_bundle):
        raise NotImplementedError

    def forward(self, ray_bundle):
        done = ray_bundle.nears is not None and ray_bundle.fars is not None
        return ray_bundle if done else self.set_nears_and_fars(ray_bundle)


class NearFarCollider(SceneCollider):
    """Constant planes for every ray; in eval mode the near plane drops to 0 unless reset_near_plane is off."""

    def __init__(self, near_plane: float, far_plane: float, reset_near_plane: bool = True, **kwargs) -> None:
        self.near_plane, self.far_plane, self.reset_near_plane = near_plane, far_plane, reset_near_plane
        super().__init__(**kwargs)

    def set_nears_and_fars(self, ray_bundle):
        near = self.near_plane if (self.training or not self.reset_near_plane) else 0
        template = ray_bundle.origins[..., 0:1]
        ray_bundle.nears = torch.ones_like(template) * near
        ray_bundle.fars = torch.ones_like(template) * self.far_plane
        return ray_bundle
