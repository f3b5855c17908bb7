"""Procedural scene for the PSNR stand-in (Blender Lego is not in the container). An analytic emissive
volume (two Gaussian density blobs, position-dependent colour) seen by pinhole cameras on a sphere; ground-truth pixels
by fine quadrature of the volume-rendering integral. 300 steps on 120 views fit the training views to ~32-35 dB and
HELD-OUT views to ~28-30 dB (with the 12 views of round 1 the colour was carried by the view direction and a new view
stayed at 7 dB), so both statements of the reference's acceptance test can be checked: novel-view PSNR above 20 dB
(tests/test_nerfacto_integration.py:62-72) and agreement of the two implementations' PSNRs after identical training
(north_star: within 0.1 dB). Test infrastructure: shared by the fixture generator (CPU oracle training runs,
tests/golden/make_psnr_fixture.py) and the GPU test that must reproduce its PSNR."""
import functools

import numpy as np
import torch

H = W = 24
N_TRAIN, N_HELD_OUT, STEPS, RAYS_PER_STEP = 120, 20, 300, 512
SEEDS = (0, 1, 2, 3, 4, 5, 6, 7)  # independent runs: different initialisation (41 + s) and ray batches (9 + s)
FOCAL = 28.0


def cameras(seed=4):
    """[N_TRAIN + N_HELD_OUT, 3, 4] camera-to-world (nerfstudio convention: -z forward, +y up), the last ones held out."""
    rs = np.random.RandomState(seed)
    c2w = []
    for i in range(N_TRAIN + N_HELD_OUT):
        v = rs.standard_normal(3)
        pos = 1.25 * v / np.linalg.norm(v)
        fwd = -pos / np.linalg.norm(pos)
        up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.9 else np.array([0.0, 1.0, 0.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w.append(np.stack([right, up, -fwd, pos], axis=1))
    return np.stack(c2w).astype(np.float32)


def rays_of(c2w, cam_idx, ys, xs):
    """Pinhole rays through pixel centres (cameras.py:598-634, 781-787): origins, unit directions."""
    c = c2w[cam_idx]
    x = (xs + 0.5 - W / 2.0) / FOCAL
    y = -(ys + 0.5 - H / 2.0) / FOCAL
    d_cam = np.stack([x, y, -np.ones_like(x)], axis=-1)
    d = np.einsum("nij,nj->ni", c[:, :, :3], d_cam)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    return c[:, :, 3].astype(np.float32), d.astype(np.float32)


def field(x):
    """density [.., 1], colour [.., 3] of the analytic volume."""
    s = 40.0 * np.exp(-np.sum(x * x, -1) / (2 * 0.25**2))
    s = s + 25.0 * np.exp(-np.sum((x - np.array([0.4, 0.2, -0.1])) ** 2, -1) / (2 * 0.15**2))
    col = 0.5 + 0.5 * np.sin(3.0 * x + np.array([0.0, 2.0, 4.0]))
    return s, col


def ground_truth(o, d, t_max=4.0, n=768):
    """Volume-rendering integral by midpoint quadrature; the remainder takes the colour at t_max (what the model's
    "last_sample" background can represent)."""
    t = (np.arange(n) + 0.5) * (t_max / n)
    x = o[:, None, :] + d[:, None, :] * t[None, :, None]
    s, col = field(x.astype(np.float64))
    dt = t_max / n
    alpha = 1.0 - np.exp(-s * dt)
    trans = np.exp(-np.concatenate([np.zeros_like(s[:, :1]), np.cumsum(s * dt, -1)[:, :-1]], -1))
    w = alpha * trans
    rgb = np.sum(w[..., None] * col, 1) + (1.0 - w.sum(-1, keepdims=True)) * col[:, -1]
    return rgb.astype(np.float32)


def batches(seed=9, steps=None):
    """STEPS training batches: (origins, directions, camera indices, target rgb, jitter [3, n]) — seeded numpy streams,
    identical on every machine."""
    rs = np.random.RandomState(seed)
    c2w = cameras()
    out = []
    for _ in range(STEPS if steps is None else steps):
        cam = rs.randint(0, N_TRAIN, RAYS_PER_STEP)
        ys, xs = rs.randint(0, H, RAYS_PER_STEP), rs.randint(0, W, RAYS_PER_STEP)
        o, d = rays_of(c2w, cam, ys.astype(np.float64), xs.astype(np.float64))
        jit = rs.uniform(0, 1, (3, RAYS_PER_STEP)).astype(np.float32)
        out.append((o, d, cam.astype(np.int64), ground_truth(o, d), jit))
    return out


EVAL_CAMERAS = (0, 3, N_TRAIN, N_TRAIN + 1)  # views whose rendered images are kept in the fixtures (2 training, 2 held out)
ALL_CAMERAS = tuple(range(N_TRAIN + N_HELD_OUT))  # per-view PSNRs are kept for all of them


@functools.lru_cache(maxsize=None)  # (the same 140 views for every seed and run: ~4 s of quadrature each time)
def full_view(cam_id):
    """All H x W rays of one camera and their ground-truth colours."""
    c2w = cameras()
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cam = np.full(H * W, cam_id)
    o, d = rays_of(c2w, cam, ys.reshape(-1).astype(np.float64), xs.reshape(-1).astype(np.float64))
    return o, d, ground_truth(o, d)


def psnr(pred, gt):
    mse = float(np.mean((np.asarray(pred, np.float64) - np.asarray(gt, np.float64)) ** 2))
    return -10.0 * np.log10(mse)


def to_t(a):
    return torch.from_numpy(np.ascontiguousarray(a))
