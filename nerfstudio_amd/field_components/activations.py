"""trunc_exp (reference: nerfstudio/field_components/activations.py:28-54).

On the fused fields the activation and its clamped backward run inside the MLP kernels
(csrc/density_mlp.hip, csrc/field_mlp.hip). This stand-alone autograd function is elementwise tensor plumbing for
callers that apply it themselves; it is not on the hot path."""
import torch


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply
