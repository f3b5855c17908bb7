"""CPU check of the algebra behind the ray terms (include/nsamd.h, nsamd_field_mlp.ray_terms; csrc/field_mlp.hip RAYC kernels):
head layer 0 of the nerfacto field takes [SH16(dir) | geo15 | appearance32] (fields/nerfacto_field.py:283-310); 48 of those 63
inputs are the same for every sample of a ray. The kernels compute that share of the pre-activation once per ray and form the
per-ray columns of the layer's weight gradient and the appearance rows' gradient from the per-tile sums of
dL/d(pre-activation). Here: the same factorisation in float64 torch against autograd on the dense layer."""
import numpy as np
import torch


def test_ray_terms_factorisation_equals_the_dense_layer():
    rs = np.random.RandomState(0)
    R, S, ncam = 7, 48, 3
    t = lambda *s: torch.from_numpy(rs.standard_normal(s))  # noqa: E731  (float64)
    W0 = t(64, 63).requires_grad_(True)
    b0 = t(64).requires_grad_(True)
    emb = t(ncam, 32).requires_grad_(True)
    sh = t(R, 16)                       # SH of the ray's view direction
    cams = torch.from_numpy(rs.randint(0, ncam, (R,)))
    geo = t(R, S, 15)                   # per-sample geo features
    up = t(R, S, 64)                    # dL/d(pre-activation) of head layer 0, per sample

    # dense layer, autograd
    x = torch.cat([sh[:, None, :].expand(R, S, 16), geo, emb[cams][:, None, :].expand(R, S, 32)], dim=-1)
    pre = x @ W0.t() + b0
    (pre * up).sum().backward()

    # ray terms: the per-ray share once per ray ...
    with torch.no_grad():
        cols_ray = list(range(16)) + list(range(31, 63))
        x_ray = torch.cat([sh, emb[cams]], dim=-1)                                  # ray_inputs [R, 48]
        terms = b0 + x_ray @ W0[:, cols_ray].t()                                    # ray_terms [R, 64]
        pre2 = terms[:, None, :] + geo @ W0[:, 16:31].t()
        np.testing.assert_allclose(pre2.numpy(), pre.detach().numpy(), rtol=1e-12, atol=1e-12)
        # ... and the gradients from the tiles' sums of dL/d(pre-activation): 16-sample tiles, three per ray
        S_tile = up.reshape(R, S // 16, 16, 64).sum(dim=2)                           # [R, 3, 64]
        dW_ray = torch.einsum("rtn,rc->nc", S_tile, x_ray)                           # per-ray columns
        dW_geo = torch.einsum("rsn,rsc->nc", up, geo)
        rows = torch.einsum("rtn,na->rta", S_tile, W0[:, 31:63])                     # the tiles' appearance rows (what the kernel stores)
        d_emb = torch.zeros_like(emb)
        d_emb.index_add_(0, cams, rows.sum(dim=1))
        np.testing.assert_allclose(dW_ray.numpy(), W0.grad[:, cols_ray].numpy(), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(dW_geo.numpy(), W0.grad[:, 16:31].numpy(), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(S_tile.sum(dim=(0, 1)).numpy(), b0.grad.numpy(), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(d_emb.numpy(), emb.grad.numpy(), rtol=1e-10, atol=1e-10)
