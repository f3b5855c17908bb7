"""CPU tests of the arithmetic the HIP kernels share (nerfstudio_amd/csrc/common.h, scatter.h), compiled for the host by
tests/hostcheck/helpers.cc: spatial hash, cell location and blend weights, scene contraction and its backward, position
normalisation + selector, spherical harmonics, the piecewise spacing functions, Frustums.get_positions, nan_to_num, and
the 64-bit fixed-point accumulation of the table scatter. Pinned to the reference's known answers (tests/golden/kat.npz,
hashgrid.npz — written by the reference itself) and to the oracle. The library built here is test infrastructure: the
product never loads it (the kernels run the same functions on the device)."""
import ctypes as C
import math
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import nerfacto_oracle as O

F32P = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
I64P = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def hc(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostcheck") / "libhostcheck.so")
    src = os.path.join(ROOT, "tests", "hostcheck", "helpers.cc")
    # -ffp-contract=off as the kernels are built (csrc/Makefile): no FMA contraction of a*b+c
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", out, src], check=True)
    lib = C.CDLL(out)
    lib.hc_hash_corners.argtypes = [F32P, C.c_int64, C.c_float, C.c_int, I64P]
    lib.hc_cell_weights.argtypes = [F32P, C.c_int64, C.c_float, F32P]
    lib.hc_hash_fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int]
    lib.hc_hash_fn.restype = C.c_uint32
    lib.hc_contract.argtypes = [F32P, C.c_int64]
    lib.hc_contract_bwd.argtypes = [F32P, F32P, C.c_int64]
    lib.hc_normalise.argtypes = [F32P, C.c_int64, C.c_int, F32P, F32P, F32P]
    lib.hc_sh4.argtypes = [F32P, C.c_int64, F32P]
    lib.hc_spacing.argtypes = [F32P, C.c_int64, F32P, F32P]
    lib.hc_spacing_to_euclidean.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    lib.hc_spacing_to_euclidean.restype = C.c_float
    lib.hc_positions.argtypes = [F32P, F32P, F32P, C.c_int64, C.c_int64, F32P]
    opt = C.c_void_p  # (nullable float arrays: passed as raw addresses)
    lib.hc_positions_burst.argtypes = [opt, opt, opt, opt, C.c_int64, C.c_int64, C.c_int, F32P]
    lib.hc_nan_to_num.argtypes = [F32P, C.c_int64, C.c_float]
    lib.hc_fixed_scale.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hc_to_fixed.argtypes = [F32P, C.c_int64, C.c_int, I64P]
    lib.hc_fixed_sum.argtypes = [F32P, I64P, C.c_int64, C.c_int]
    lib.hc_fixed_sum.restype = C.c_float
    lib.hc_cam_exp_map.argtypes = [C.c_int, F32P, C.c_int64, F32P]
    lib.hc_cam_exp_map_bwd.argtypes = [C.c_int, F32P, F32P, C.c_int64, C.c_float, C.c_float, C.c_int, F32P]
    return lib


def test_hash_fn_known_answers(hc):
    """HashEncoding(num_levels=2, log2_hashmap_size=5).hash_fn of the reference (kat.npz)."""
    kat = load_golden("kat")
    # row l of the input is hashed as level l: the reference adds the level's table offset l * 2^5 (encodings.py:412-413)
    for level, ((ix, iy, iz), want) in enumerate(zip(kat["hash_in"], kat["hash_out"])):
        assert hc.hc_hash_fn(int(ix), int(iy), int(iz), 5) + level * 32 == int(want)
    assert [hc.hc_hash_fn(3, 7, 11, 5), hc.hc_hash_fn(1, 2, 3, 5) + 32] == [19, 60]
    # negative coordinates: int64 `% table_size` of the reference == uint32 wrap-around for power-of-two tables
    for ix, iy, iz in [(-1, 5, 9), (-7, -3, 2), (12, -40, -1)]:
        ref = ((ix * 1) ^ (iy * 2654435761) ^ (iz * 805459861)) % 2**19
        assert hc.hc_hash_fn(ix, iy, iz, 19) == ref


def test_hash_forward_from_the_kernel_helpers_reproduces_the_reference_fixture(hc):
    """hashgrid.npz holds HashEncoding.pytorch_fwd of the reference on 240 points: rebuilding it from locate_cell /
    corner_index / the blend weights (the functions every hash kernel is made of) must give the same features, and the
    corner indices must equal the oracle's bit for bit."""
    g = load_golden("hashgrid")
    x, table, scalings = g["x"], g["table"], g["scalings"]
    levels = len(scalings)
    T = table.shape[0] // levels
    log2_T = int(math.log2(T))
    out = np.zeros((x.shape[0], 2 * levels), np.float32)
    for lvl, scale in enumerate(scalings):
        idx = np.zeros((x.shape[0], 8), np.int64)
        w = np.zeros((x.shape[0], 3), np.float32)
        hc.hc_hash_corners(x, x.shape[0], float(scale), log2_T, idx)
        hc.hc_cell_weights(x, x.shape[0], float(scale), w)
        sx = x * np.float32(scale)
        lo, hi = np.floor(sx).astype(np.int64), np.ceil(sx).astype(np.int64)
        for k in range(8):
            cx = np.where(k & 1, hi[:, 0], lo[:, 0])
            cy = np.where(k & 2, hi[:, 1], lo[:, 1])
            cz = np.where(k & 4, hi[:, 2], lo[:, 2])
            np.testing.assert_array_equal(idx[:, k] + lvl * T, O.hash_corner_index(cx, cy, cz, lvl, T))
        acc = np.zeros((x.shape[0], 2), np.float64)
        for k in range(8):
            wk = np.ones(x.shape[0], np.float64)
            for a, bit in enumerate((1, 2, 4)):
                wk = wk * np.where(k & bit, w[:, a], 1.0 - w[:, a])
            acc += wk[:, None] * table[idx[:, k] + lvl * T]
        out[:, 2 * lvl:2 * lvl + 2] = acc
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=2e-6)


def test_contraction_known_answers_and_backward(hc):
    kat = load_golden("kat")
    x = np.ascontiguousarray(kat["contract_in"])
    hc.hc_contract(x, x.shape[0])
    np.testing.assert_array_equal(x, kat["contract_out"])
    rs = np.random.RandomState(3)
    pts = (rs.standard_normal((4000, 3)) * 2.5).astype(np.float32)
    pts[:50] *= 1e-3  # deep inside the unit cube: identity branch
    mine = pts.copy()
    hc.hc_contract(mine, mine.shape[0])
    np.testing.assert_allclose(mine, O.contract_linf(torch.from_numpy(pts)).numpy(), rtol=0, atol=2e-7)
    # backward against autograd through the oracle's restatement of SceneContraction (spatial_distortions.py:66-69)
    g = rs.standard_normal(pts.shape).astype(np.float32)
    t = torch.from_numpy(pts).requires_grad_(True)
    O.contract_linf(t).backward(torch.from_numpy(g))
    mine_g = g.copy()
    hc.hc_contract_bwd(pts, mine_g, pts.shape[0])
    np.testing.assert_allclose(mine_g, t.grad.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("contraction", [True, False])
def test_position_normalisation_and_selector(hc, contraction):
    rs = np.random.RandomState(4)
    pts = (rs.standard_normal((3000, 3)) * (3.0 if contraction else 0.8)).astype(np.float32)
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    want, sel = O.normalise_positions(torch.from_numpy(pts), contraction, aabb)
    mine, mine_sel = pts.copy(), np.zeros(pts.shape[0], np.float32)
    lo, hi = aabb[0].numpy().copy(), aabb[1].numpy().copy()
    hc.hc_normalise(mine, pts.shape[0], 1 if contraction else 2, lo, hi, mine_sel)
    np.testing.assert_array_equal(mine_sel.astype(bool), sel.numpy())
    np.testing.assert_allclose(mine, want.numpy(), rtol=0, atol=2e-7)
    if not contraction:
        assert 0 < int(mine_sel.sum()) < pts.shape[0]  # both sides of the selector are exercised


def test_spherical_harmonics_known_answer_and_oracle(hc):
    kat = load_golden("kat")
    out = np.zeros((1, 16), np.float32)
    hc.hc_sh4(np.ascontiguousarray(kat["sh_in"]), 1, out)
    np.testing.assert_allclose(out, kat["sh_out"], rtol=0, atol=1e-7)
    rs = np.random.RandomState(5)
    d = rs.standard_normal((2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    out = np.zeros((d.shape[0], 16), np.float32)
    hc.hc_sh4(d, d.shape[0], out)
    np.testing.assert_allclose(out, O.sh_levels4(torch.from_numpy(d)).numpy(), rtol=0, atol=1e-6)
    # orthonormality, as the reference's own test checks it (tests/utils/test_spherical_harmonics.py:8-16, atol 1.5e-2)
    n = 200000
    d = rs.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    sh = np.zeros((n, 16), np.float32)
    hc.hc_sh4(d, n, sh)
    gram = 4 * math.pi * (sh.astype(np.float64).T @ sh.astype(np.float64)) / n
    np.testing.assert_allclose(gram, np.eye(16), atol=1.5e-2)


def test_spacing_functions(hc):
    x = np.concatenate([np.linspace(0.0, 0.999, 500), np.linspace(1.0, 1000.0, 500)]).astype(np.float32)
    fwd, inv = np.zeros_like(x), np.zeros_like(x)
    hc.hc_spacing(x, x.shape[0], fwd, inv)
    np.testing.assert_array_equal(fwd, O.spacing_fn(torch.from_numpy(x)).numpy())
    s = fwd[fwd < 1.0]
    back = np.zeros_like(s)
    hc.hc_spacing(s, s.shape[0], np.zeros_like(s), back)
    np.testing.assert_array_equal(back, O.spacing_fn_inv(torch.from_numpy(s)).numpy())
    np.testing.assert_allclose(back, x[fwd < 1.0], rtol=2e-4)  # inverse of each other (ray_samplers.py:244-245)
    # the reference's piecewise sampler at near 0.05 / far 1000, 4 samples (kat.npz): bin edges s = i / 4
    kat = load_golden("kat")
    s_near, s_far = float(O.spacing_fn(torch.tensor(0.05))), float(O.spacing_fn(torch.tensor(1000.0)))
    edges = [hc.hc_spacing_to_euclidean(0, np.float32(i / 4), np.float32(s_near), np.float32(s_far)) for i in range(5)]
    np.testing.assert_allclose(edges[:4], kat["pw_starts"], rtol=1e-6)
    np.testing.assert_allclose(edges[1:], kat["pw_ends"], rtol=1e-6)
    assert hc.hc_spacing_to_euclidean(1, 0.25, 2.0, 6.0) == 3.0  # UniformSampler: identity spacing


def test_frustum_positions(hc):
    rs = np.random.RandomState(6)
    rays, S = 37, 11
    o = rs.standard_normal((rays, 3)).astype(np.float32)
    d = rs.standard_normal((rays, 3)).astype(np.float32)
    t = np.sort(rs.uniform(0.05, 20.0, (rays, S + 1)).astype(np.float32), axis=-1)
    out = np.zeros((rays * S, 3), np.float32)
    hc.hc_positions(o, d, np.ascontiguousarray(t), rays, S, out)
    want = O.sample_positions(torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(t)).reshape(-1, 3).numpy()
    np.testing.assert_array_equal(out, want)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("explicit", [False, True])
def test_burst_position_loaders_equal_the_plain_one(hc, variant, explicit):
    """load_position_burst / load_positions_burst<N> (csrc/common.h: the loads of a lane's positions issued in one go — hash
    forward, the scatter's route kernels) against load_position on ray-form and explicit points: same bits."""
    rs = np.random.RandomState(16)
    rays, S = 29, 7  # (rays * S not a multiple of 4: the last group of the <4> variant is clamped)
    o = rs.standard_normal((rays, 3)).astype(np.float32)
    d = rs.standard_normal((rays, 3)).astype(np.float32)
    t = np.ascontiguousarray(np.sort(rs.uniform(0.05, 20.0, (rays, S + 1)).astype(np.float32), axis=-1))
    want = np.zeros((rays * S, 3), np.float32)
    hc.hc_positions(o, d, t, rays, S, want)
    got = np.zeros_like(want)
    pos = want.copy()
    addr = lambda a: a.ctypes.data  # noqa: E731
    if explicit:
        hc.hc_positions_burst(None, None, None, addr(pos), rays, S, variant, got)
    else:
        hc.hc_positions_burst(addr(o), addr(d), addr(t), None, rays, S, variant, got)
    np.testing.assert_array_equal(got, want)


def test_nan_to_num(hc):
    v = np.array([np.nan, np.inf, -np.inf, 1.5, -0.0, 3.4028234663852886e38], np.float32)
    mine = v.copy()
    hc.hc_nan_to_num(mine, v.shape[0], 0.0)
    np.testing.assert_array_equal(mine, torch.nan_to_num(torch.from_numpy(v)).numpy())
    mine = v.copy()
    hc.hc_nan_to_num(mine, v.shape[0], 7.0)
    np.testing.assert_array_equal(mine, torch.nan_to_num(torch.from_numpy(v), nan=7.0).numpy())


# ---- the scatter's 64-bit fixed point (csrc/scatter.h) -----------------------------------------------------------------
def _scale(hc, max_value, headroom):
    k, empty, bad = C.c_int(), C.c_int(), C.c_int()
    bits = int(np.float32(max_value).view(np.uint32))
    hc.hc_fixed_scale(bits, headroom, C.byref(k), C.byref(empty), C.byref(bad))
    return k.value, bool(empty.value), bool(bad.value)


def test_fixed_scale_flags_and_range(hc):
    assert _scale(hc, 0.0, 20)[1] and _scale(hc, 1e-39, 20)[1]  # zero / denormal maximum: nothing to add
    assert _scale(hc, np.inf, 20)[2] and _scale(hc, np.nan, 20)[2]  # a non-finite gradient poisons the level
    for mx in (1e-30, 3e-7, 1.0, 123.456, 1e20):
        for headroom in (18, 20, 24):
            k, empty, bad = _scale(hc, mx, headroom)
            assert not empty and not bad
            # |max * 2^k| < 2^(62 - headroom): 2^headroom summands cannot overflow 63 bits
            assert abs(Fraction(float(np.float32(mx)))) * Fraction(2) ** k < Fraction(2) ** (62 - headroom)
            assert abs(Fraction(float(np.float32(mx)))) * Fraction(2) ** k >= Fraction(2) ** (60 - headroom)


def test_to_fixed_is_exact_truncation(hc):
    """to_fixed(v, k) == trunc(v * 2^k) in exact rational arithmetic, for both signs, down to values that keep only a
    few bits (the hardware-conversion split into a high and a low part must not lose or double a bit)."""
    rs = np.random.RandomState(7)
    for mx, headroom in ((1.0, 20), (3.7e-4, 18), (812.0, 22)):
        k, _, _ = _scale(hc, mx, headroom)
        mags = np.float32(mx) * np.exp2(-rs.uniform(0, 45, 6000)).astype(np.float32)
        v = (mags * rs.choice([-1.0, 1.0], mags.shape[0])).astype(np.float32)
        v[:4] = [np.float32(mx), -np.float32(mx), 0.0, -0.0]
        got = np.zeros(v.shape[0], np.int64)
        hc.hc_to_fixed(v, v.shape[0], k, got)
        for val, g in zip(v, got):
            exact = Fraction(float(val)) * Fraction(2) ** k
            want = math.floor(exact) if exact >= 0 else -math.floor(-exact)
            assert int(g) == want, (float(val), k)


def test_fixed_sums_do_not_depend_on_the_order(hc):
    """The property the deterministic training rests on: a tile's sum is the same bits whatever order
    the records arrive in — also for cancelling sums, where float addition would leave order-dependent residue."""
    rs = np.random.RandomState(8)
    base = rs.standard_normal(3000).astype(np.float32)
    v = np.concatenate([base, -base, (rs.standard_normal(500) * 1e-6).astype(np.float32)])  # cancels to ~1e-5
    k, _, _ = _scale(hc, float(np.abs(v).max()), 20)
    sums = set()
    for seed in range(6):
        order = np.random.RandomState(seed).permutation(v.shape[0]).astype(np.int64)
        sums.add(np.float32(hc.hc_fixed_sum(v, order, v.shape[0], k)).tobytes())
    assert len(sums) == 1
    got = np.frombuffer(next(iter(sums)), np.float32)[0]
    exact = float(sum(Fraction(float(x)) for x in v))
    assert abs(got - exact) <= v.shape[0] * 2.0 ** -k + abs(exact) * 2.0 ** -23
    float_sums = {np.float32(np.sum(v[np.random.RandomState(s).permutation(v.shape[0])], dtype=np.float32)).tobytes()
                  for s in range(6)}
    assert len(float_sums) > 1  # the same data summed in fp32 does depend on the order


@pytest.mark.parametrize("mode", ["SO3xR3", "SE3"])
def test_camera_exponential_maps_and_their_backward(hc, mode):
    """csrc/camera.h (nsamd_camera_apply / nsamd_camera_backward run these per camera on the device) against this package's
    torch mirror of cameras/lie_groups.py:25-117 — itself pinned to the reference's CameraOptimizer by
    tests/golden/camera_opt.npz — and torch autograd through it: every branch (the 1e-4 clamp of SO3xR3, the Taylor forms of
    SE3 below |w| = 1e-2, the exact zero the parameter starts from), and the L2 regulariser of camera_optimizers.py:179-185."""
    from nerfstudio_amd.cameras.camera_optimizers import CameraOptimizer, CameraOptimizerConfig
    from nerfstudio_amd.cameras.lie_groups import exp_map_SE3, exp_map_SO3xR3

    rs = np.random.RandomState(3)
    n = 64
    pose = rs.normal(0, 0.3, (n, 6)).astype(np.float32)
    pose[:8, 3:] *= 1e-3                      # |w| ~ 5e-4: below both branch points
    pose[8:16, 3:] *= 2e-2                    # |w| ~ 1e-2: around them
    pose[16] = 0.0                            # the initial state of the parameter
    pose[17, 3:] = 0.0                        # pure translation
    pose[18, :3] = 0.0                        # pure rotation
    up = rs.normal(0, 1, (n, 3, 4)).astype(np.float32)
    fn = exp_map_SO3xR3 if mode == "SO3xR3" else exp_map_SE3
    p = torch.from_numpy(pose.astype(np.float64)).requires_grad_(True)
    out64 = fn(p)
    (out64 * torch.from_numpy(up.astype(np.float64))).sum().backward()
    want_out32 = fn(torch.from_numpy(pose)).numpy()
    code = 1 if mode == "SO3xR3" else 2
    got = np.zeros((n, 3, 4), np.float32)
    hc.hc_cam_exp_map(code, pose, n, got.reshape(-1))
    # forward: fp32, the reference's operations -> within a few ulp of the fp32 torch evaluation; float64 is further away
    # where the reference's own fp32 forms cancel ((1 - cos t) / t^2 right above the Taylor branch point)
    np.testing.assert_allclose(got, want_out32, rtol=0, atol=3e-7)
    np.testing.assert_allclose(got, out64.detach().numpy(), rtol=0, atol=2e-6)
    gp = np.zeros((n, 6), np.float32)
    hc.hc_cam_exp_map_bwd(code, pose, up.reshape(-1), n, 0.0, 0.0, 0, gp.reshape(-1))
    want = p.grad.numpy()
    np.testing.assert_allclose(gp, want, rtol=2e-5, atol=2e-6)
    # + the regulariser (mean over cameras of the two norms; zero gradient where a norm is zero, as torch masks it)
    cfg = CameraOptimizerConfig(mode=mode, trans_l2_penalty=1e-2, rot_l2_penalty=1e-3)
    opt = CameraOptimizer(cfg, num_cameras=n, device="cpu").double()
    with torch.no_grad():
        opt.pose_adjustment.copy_(torch.from_numpy(pose.astype(np.float64)))
    ld = {}
    opt.get_loss_dict(ld)
    c = opt(torch.arange(n))
    ((c * torch.from_numpy(up.astype(np.float64))).sum() + ld["camera_opt_regularizer"]).backward()
    hc.hc_cam_exp_map_bwd(code, pose, up.reshape(-1), n, 1e-2, 1e-3, 1, gp.reshape(-1))
    np.testing.assert_allclose(gp, opt.pose_adjustment.grad.numpy(), rtol=2e-5, atol=2e-6)
