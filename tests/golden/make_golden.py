#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running THE REFERENCE ITSELF (read-only import of
/root/reference, torch path, CPU) on seeded inputs.  Run in the authoring container only:

    python tests/golden/make_golden.py

The fixtures are what pins `oracle/nerfacto_oracle.py` (tests/test_oracle_vs_golden.py) and, on the GPU box, the HIP
path (tests/test_gpu_*.py).  /root/reference does not exist on the GPU box; nothing at test time imports it.

Parameters are produced by `oracle.nerfacto_oracle.init_params` (a numpy RandomState stream) and LOADED INTO the
reference modules, so fixtures only need to store the seed, not the tables.  The jitter draws are injected by
temporarily replacing `torch.rand`; `torch.searchsorted` is wrapped to record the reference's integer indices.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_refstubs"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

import warnings  # noqa: E402

warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerfstudio.cameras.cameras import Cameras, CameraType  # noqa: E402
from nerfstudio.cameras.rays import Frustums, RayBundle, RaySamples  # noqa: E402
from nerfstudio.field_components.encodings import HashEncoding, SHEncoding  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.spatial_distortions import SceneContraction  # noqa: E402
from nerfstudio.fields.density_fields import HashMLPDensityField  # noqa: E402
from nerfstudio.fields.nerfacto_field import NerfactoField  # noqa: E402
from nerfstudio.model_components.losses import distortion_loss, interlevel_loss  # noqa: E402
from nerfstudio.model_components.ray_generators import RayGenerator  # noqa: E402
from nerfstudio.model_components.ray_samplers import (  # noqa: E402
    PDFSampler,
    ProposalNetworkSampler,
    UniformLinDispPiecewiseSampler,
)
from nerfstudio.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer  # noqa: E402
from nerfstudio.model_components.scene_colliders import NearFarCollider  # noqa: E402

from oracle import nerfacto_oracle as orc  # noqa: E402

torch.set_num_threads(4)
AABB = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


class replay_rand:
    """Context manager: torch.rand(shape) returns the queued tensors (in order)."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self._orig = torch.rand

        def fake(*size, **kw):
            t = self.draws.pop(0)
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            assert tuple(t.shape) == shape, (t.shape, shape)
            return t.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


class record_searchsorted:
    def __enter__(self):
        self._orig = torch.searchsorted
        self.calls = []

        def wrapped(*a, **k):
            r = self._orig(*a, **k)
            self.calls.append(r.clone())
            return r

        torch.searchsorted = wrapped
        return self

    def __exit__(self, *a):
        torch.searchsorted = self._orig


def small_cfg(main_log2=10, prop_log2=8, num_images=7):
    return orc.NerfactoCfg(
        main_grid=orc.HashGridCfg(16, 16, 2048, main_log2),
        prop_grids=(orc.HashGridCfg(5, 16, 128, prop_log2), orc.HashGridCfg(5, 16, 256, prop_log2)),
        num_images=num_images,
    )


def build_reference(cfg, params):
    """Reference modules wired like NerfactoModel.populate_modules (models/nerfacto.py:144-253), torch impl."""
    sc = SceneContraction(order=float("inf"))
    fld = NerfactoField(
        AABB,
        num_images=cfg.num_images,
        num_levels=cfg.main_grid.num_levels,
        base_res=cfg.main_grid.min_res,
        max_res=cfg.main_grid.max_res,
        log2_hashmap_size=cfg.main_grid.log2_hashmap_size,
        spatial_distortion=sc,
        implementation="torch",
        average_init_density=cfg.average_init_density,
        use_average_appearance_embedding=cfg.use_average_appearance_embedding,
    )
    props = torch.nn.ModuleList(
        [
            HashMLPDensityField(
                AABB,
                spatial_distortion=sc,
                hidden_dim=cfg.prop_hidden_dim,
                log2_hashmap_size=g.log2_hashmap_size,
                num_levels=g.num_levels,
                max_res=g.max_res,
                base_res=g.min_res,
                use_linear=False,
                average_init_density=cfg.average_init_density,
                implementation="torch",
            )
            for g in cfg.prop_grids
        ]
    )
    sd = {k[len("field.") :]: v.clone() for k, v in params.items() if k.startswith("field.")}
    missing, unexpected = fld.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("aabb" in m or "max_res" in m or "num_levels" in m or "log2" in m) for m in missing), missing
    for i, p in enumerate(props):
        sd = {k[len(f"proposal_networks.{i}.") :]: v.clone() for k, v in params.items() if k.startswith(f"proposal_networks.{i}.")}
        sd["mlp_base.0.hash_table"] = sd["encoding.hash_table"]
        missing, unexpected = p.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(("aabb" in m or "max_res" in m or "num_levels" in m or "log2" in m) for m in missing), missing
    return fld, props


# ------------------------------------------------------------------------------------------------------------------
def gen_kat():
    enc = HashEncoding(num_levels=2, log2_hashmap_size=5, implementation="torch")
    hk = enc.hash_fn(torch.tensor([[[3, 7, 11], [1, 2, 3]]]))
    sh = SHEncoding(levels=4, implementation="torch")(torch.tensor([[0.0, 1.0, 0.0]]))
    sc = SceneContraction(order=float("inf"))
    cin = torch.tensor([[2.0, 0, 0], [-4.0, 2, 1], [0.5, 0.2, 0.1]])
    cout = sc(cin)
    # piecewise sampler, eval, 4 samples
    rb = RayBundle(origins=torch.zeros(1, 3), directions=torch.tensor([[0.0, 0, 1]]), pixel_area=torch.ones(1, 1))
    rb = NearFarCollider(0.05, 1000.0)(rb)
    smp = UniformLinDispPiecewiseSampler(num_samples=4)
    smp.eval()
    rs = smp(rb)
    w = rs.get_weights(torch.ones_like(rs.frustums.starts))
    pdf = PDFSampler(num_samples=3, include_original=False)
    pdf.eval()
    rs2 = pdf(rb, rs, w)
    rgbs = torch.tensor([[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1.0, 1, 1]]])
    r = RGBRenderer("last_sample")
    r.train()
    comp = r(rgb=rgbs, weights=w)
    dmed = DepthRenderer("median")(weights=w, ray_samples=rs)
    dexp = DepthRenderer("expected")(weights=w, ray_samples=rs)
    # Frustums.get_positions KAT from the reference's own tests/cameras/test_rays.py:11-30
    fr = Frustums(
        origins=torch.ones((5, 3)),
        directions=torch.tensor([[0.0, 1.0, 0.0]]).expand(5, 3) if False else torch.ones((5, 3)) * torch.tensor([0.0, 1.0, 0.0]),
        starts=torch.ones((5, 1)) * 2,
        ends=torch.ones((5, 1)) * 3,
        pixel_area=torch.ones((5, 1)),
    )
    scal = {}
    for name, (L, lo, hi, t) in {"main": (16, 16, 2048, 19), "prop0": (5, 16, 128, 17), "prop1": (5, 16, 256, 17)}.items():
        scal[f"scalings_{name}"] = HashEncoding(
            num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=4, implementation="torch"
        ).scalings
    save(
        "kat",
        hash_in=np.array([[3, 7, 11], [1, 2, 3]]),
        hash_out=hk[0],
        sh_in=np.array([[0.0, 1.0, 0.0]], np.float32),
        sh_out=sh,
        contract_in=cin,
        contract_out=cout,
        pw_starts=rs.frustums.starts[0, :, 0],
        pw_ends=rs.frustums.ends[0, :, 0],
        pw_weights=w[0, :, 0],
        pdf_starts=rs2.frustums.starts[0, :, 0],
        pdf_ends=rs2.frustums.ends[0, :, 0],
        rgb_last_sample=comp[0],
        depth_median=dmed[0],
        depth_expected=dexp[0],
        frustum_positions=fr.get_positions(),
        **scal,
    )


def gen_hashgrid():
    torch.manual_seed(1)
    L, lo, hi, log2T = 6, 4, 64, 8
    enc = HashEncoding(num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=log2T, implementation="torch")
    rs = np.random.RandomState(11)
    table = torch.from_numpy(rs.standard_normal((L * 2**log2T, 2)).astype(np.float32))
    enc.hash_table.data.copy_(table)
    x = torch.from_numpy(rs.uniform(0, 1, (240, 3)).astype(np.float32))
    # edge cases: exact grid nodes (ceil==floor), 0, 1, masked-out zeros, values on coarse lattice
    x[:8] = torch.tensor(
        [[0, 0, 0], [1, 1, 1], [0.25, 0.5, 0.75], [0.5, 0.5, 0.5], [1, 0, 0.5], [0.125, 1.0, 0.0], [1e-7, 1 - 1e-7, 0.3], [0.999999, 0.000001, 0.5]]
    )
    x = x.clone().requires_grad_(True)
    out = enc(x)
    g = torch.from_numpy(rs.standard_normal(out.shape).astype(np.float32))
    (out * g).sum().backward()
    save(
        "hashgrid",
        cfg=np.array([L, lo, hi, log2T, 2]),
        scalings=enc.scalings,
        table=table,
        x=x,
        out=out,
        gout=g,
        dx=x.grad,
        dtable=enc.hash_table.grad,
    )


def field_inputs(rs, M, cfg):
    pos = rs.standard_normal((M, 3)).astype(np.float32)
    pos[: M // 3] *= 0.4  # inside the unit cube
    pos[M // 3 : 2 * M // 3] *= 3.0  # contracted region
    pos[2 * M // 3 :] *= 40.0  # far field
    pos[0] = [0.0, 0.0, 0.0]
    pos[1] = [1.0, 1.0, 1.0]
    pos[2] = [-2.0, 0.5, 0.5]  # lands exactly on the contracted cube's face: (2 - 1/2)*(-1) = -1.5 -> 0.125
    pos[3] = [1e9, 0.0, 0.0]  # contracts to 2.0 -> normalised 1.0 -> selector False
    d = rs.standard_normal((M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cam = rs.randint(0, cfg.num_images, (M,)).astype(np.int64)
    return torch.from_numpy(pos), torch.from_numpy(d.astype(np.float32)), torch.from_numpy(cam)


def gen_fields():
    cfg = small_cfg()
    seed, std = 5, 0.5
    params = orc.init_params(cfg, seed=seed, table_std=std)
    fld, props = build_reference(cfg, params)
    rs = np.random.RandomState(21)
    M = 384
    pos, dirs, cam = field_inputs(rs, M, cfg)
    out = {"cfg_main_log2": cfg.main_grid.log2_hashmap_size, "cfg_prop_log2": cfg.prop_grids[0].log2_hashmap_size,
           "num_images": cfg.num_images, "seed": seed, "table_std": std, "positions": pos, "directions": dirs, "cams": cam}
    # proposal densities through the public density_fn (base_field.py:48-68)
    for i, p in enumerate(props):
        p.train()
        p.zero_grad()
        pp = pos.clone().requires_grad_(True)
        dens = p.density_fn(pp)[:, 0]
        g = torch.from_numpy(rs.standard_normal(dens.shape).astype(np.float32))
        (dens * g).sum().backward()
        out[f"prop{i}_density"] = dens
        out[f"prop{i}_g"] = g
        out[f"prop{i}_dpos"] = pp.grad
        out[f"prop{i}_dtable"] = p.encoding.hash_table.grad
        for j in range(2):
            out[f"prop{i}_dW{j}"] = p.mlp_base[1].layers[j].weight.grad
            out[f"prop{i}_db{j}"] = p.mlp_base[1].layers[j].bias.grad
    # main field on a [M/4, 4] "ray sample" grid so that Field.forward is exercised as the model does
    for mode in ("train", "eval"):
        fld.train(mode == "train")
        fld.zero_grad()
        R, S = M // 4, 4
        o = pos.reshape(R, S, 3).clone().requires_grad_(True)
        fr = Frustums(origins=o, directions=dirs.reshape(R, S, 3), starts=torch.zeros(R, S, 1), ends=torch.zeros(R, S, 1),
                      pixel_area=torch.ones(R, S, 1))
        rsamp = RaySamples(frustums=fr, camera_indices=cam.reshape(R, S, 1))
        fo = fld(rsamp)
        dens, rgb = fo[FieldHeadNames.DENSITY], fo[FieldHeadNames.RGB]
        out[f"main_{mode}_density"] = dens.reshape(M)
        out[f"main_{mode}_rgb"] = rgb.reshape(M, 3)
        if mode == "train":
            g1 = torch.from_numpy(rs.standard_normal((M,)).astype(np.float32))
            g2 = torch.from_numpy(rs.standard_normal((M, 3)).astype(np.float32))
            ((dens.reshape(M) * g1).sum() + (rgb.reshape(M, 3) * g2).sum()).backward()
            out["main_g_density"], out["main_g_rgb"] = g1, g2
            out["main_dpos"] = o.grad.reshape(M, 3)
            out["main_dtable"] = fld.mlp_base.model[0].hash_table.grad
            for j in range(2):
                out[f"main_base_dW{j}"] = fld.mlp_base.model[1].layers[j].weight.grad
                out[f"main_base_db{j}"] = fld.mlp_base.model[1].layers[j].bias.grad
            for j in range(3):
                out[f"main_head_dW{j}"] = fld.mlp_head.layers[j].weight.grad
                out[f"main_head_db{j}"] = fld.mlp_head.layers[j].bias.grad
            out["main_demb"] = fld.embedding_appearance.embedding.weight.grad
    save("fields", **out)


def gen_samplers():
    rs = np.random.RandomState(31)
    N = 24
    o = torch.zeros(N, 3)
    d = torch.tensor([[0.0, 0, 1]]).expand(N, 3).contiguous()
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(N, 1))
    nears = torch.full((N, 1), 0.05)
    fars = torch.full((N, 1), 1000.0)
    nears[N // 2 :] = torch.from_numpy(rs.uniform(0.5, 2.5, (N - N // 2, 1)).astype(np.float32))
    fars[N // 2 :] = torch.from_numpy(rs.uniform(4.0, 9.0, (N - N // 2, 1)).astype(np.float32))
    rb.nears, rb.fars = nears, fars
    out = {"nears": nears, "fars": fars}
    j0 = torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32))
    j1 = torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32))
    j2 = torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32))
    out.update(j0=j0, j1=j1, j2=j2)
    for mode in ("train", "eval"):
        smp = UniformLinDispPiecewiseSampler(num_samples=256, single_jitter=True)
        smp.train(mode == "train")
        with replay_rand([j0]):
            rs0 = smp(rb)
        sb = torch.cat([rs0.spacing_starts[..., 0], rs0.spacing_ends[:, -1:, 0]], -1)
        tb = torch.cat([rs0.frustums.starts[..., 0], rs0.frustums.ends[:, -1:, 0]], -1)
        out[f"{mode}_l0_s_bins"], out[f"{mode}_l0_t_bins"] = sb, tb
        # a peaky synthetic density so that weights are far from uniform; a few all-zero rays and a saturating ray
        dens = np.exp(rs.standard_normal((N, 256)) * 2.5).astype(np.float32) * 0.05
        dens[0] = 0.0
        dens[1] = 1e4
        dens[2, :100] = 0.0
        dens = torch.from_numpy(dens)
        w0 = rs0.get_weights(dens[..., None])
        out[f"{mode}_l0_density"], out[f"{mode}_l0_weights"] = dens, w0[..., 0]
        pdf = PDFSampler(include_original=False, single_jitter=True)
        pdf.train(mode == "train")
        with replay_rand([j1]), record_searchsorted() as rec:
            rs1 = pdf(rb, rs0, w0, num_samples=96)
        out[f"{mode}_l1_inds"] = rec.calls[0]
        out[f"{mode}_l1_s_bins"] = torch.cat([rs1.spacing_starts[..., 0], rs1.spacing_ends[:, -1:, 0]], -1)
        out[f"{mode}_l1_t_bins"] = torch.cat([rs1.frustums.starts[..., 0], rs1.frustums.ends[:, -1:, 0]], -1)
        dens1 = torch.from_numpy((np.exp(rs.standard_normal((N, 96)) * 2.0) * 0.5).astype(np.float32))
        w1 = rs1.get_weights(dens1[..., None])
        out[f"{mode}_l1_density"], out[f"{mode}_l1_weights"] = dens1, w1[..., 0]
        # annealed second resample 96 -> 48 (ray_samplers.py:601)
        anneal = 0.37
        with replay_rand([j2]), record_searchsorted() as rec:
            rs2 = pdf(rb, rs1, torch.pow(w1, anneal), num_samples=48)
        out[f"{mode}_l2_inds"] = rec.calls[0]
        out[f"{mode}_l2_s_bins"] = torch.cat([rs2.spacing_starts[..., 0], rs2.spacing_ends[:, -1:, 0]], -1)
        out[f"{mode}_l2_t_bins"] = torch.cat([rs2.frustums.starts[..., 0], rs2.frustums.ends[:, -1:, 0]], -1)
        out["anneal"] = anneal
    # get_weights backward
    dens = out["train_l0_density"].clone().requires_grad_(True)
    smp = UniformLinDispPiecewiseSampler(num_samples=256, single_jitter=True)
    smp.train()
    with replay_rand([j0]):
        rs0 = smp(rb)
    w = rs0.get_weights(dens[..., None])[..., 0]
    g = torch.from_numpy(rs.standard_normal(w.shape).astype(np.float32))
    (w * g).sum().backward()
    out["weights_g"], out["weights_ddensity"] = g, dens.grad
    save("samplers", **out)


def gen_render():
    rs = np.random.RandomState(41)
    N, S = 40, 48
    t_bins = np.sort(rs.uniform(0.05, 30.0, (N, S + 1)).astype(np.float32), axis=-1)
    t_bins = torch.from_numpy(t_bins)
    fr = Frustums(origins=torch.zeros(N, S, 3), directions=torch.ones(N, S, 3), starts=t_bins[:, :-1, None],
                  ends=t_bins[:, 1:, None], pixel_area=torch.ones(N, S, 1))
    rsamp = RaySamples(frustums=fr, deltas=(t_bins[:, 1:] - t_bins[:, :-1])[..., None])
    dens = np.exp(rs.standard_normal((N, S)) * 2.0).astype(np.float32) * 0.3
    dens[0] = 0.0
    dens[1] = 50.0
    dens = torch.from_numpy(dens).requires_grad_(True)
    rgb = torch.from_numpy(rs.uniform(0, 1, (N, S, 3)).astype(np.float32)).requires_grad_(True)
    w = rsamp.get_weights(dens[..., None])
    out = {"t_bins": t_bins, "density": dens, "rgb": rgb, "weights": w[..., 0]}
    for bg in ("last_sample", "white", "black", "random"):
        r = RGBRenderer(bg)
        r.train()
        out[f"rgb_train_{bg}"] = r(rgb=rgb, weights=w)
    r = RGBRenderer("last_sample")
    r.eval()
    rgb_nan = rgb.detach().clone()
    rgb_nan[3, 5, 1] = float("nan")
    rgb_nan[4, 7, 0] = 7.5
    out["rgb_eval_in"] = rgb_nan
    out["rgb_eval_last_sample"] = r(rgb=rgb_nan, weights=w.detach())
    out["accumulation"] = AccumulationRenderer()(weights=w)
    with record_searchsorted() as rec:
        out["depth_median"] = DepthRenderer("median")(weights=w.detach(), ray_samples=rsamp)
    out["depth_median_idx"] = torch.clamp(rec.calls[0], 0, S - 1)
    dexp = DepthRenderer("expected")(weights=w, ray_samples=rsamp)
    out["depth_expected"] = dexp
    r = RGBRenderer("last_sample")
    r.train()
    comp = r(rgb=rgb, weights=w)
    g_rgb = torch.from_numpy(rs.standard_normal((N, 3)).astype(np.float32))
    g_acc = torch.from_numpy(rs.standard_normal((N, 1)).astype(np.float32))
    g_dep = torch.from_numpy(rs.standard_normal((N, 1)).astype(np.float32))
    ((comp * g_rgb).sum() + (AccumulationRenderer()(weights=w) * g_acc).sum() + (dexp * g_dep).sum()).backward()
    out.update(g_rgb=g_rgb, g_acc=g_acc, g_dep=g_dep, d_density=dens.grad, d_rgb=rgb.grad)
    save("render", **out)


def _mk_samples(s_bins):
    N, S1 = s_bins.shape
    S = S1 - 1
    fr = Frustums(origins=torch.zeros(N, S, 3), directions=torch.ones(N, S, 3), starts=s_bins[:, :-1, None],
                  ends=s_bins[:, 1:, None], pixel_area=torch.ones(N, S, 1))
    return RaySamples(frustums=fr, spacing_starts=s_bins[:, :-1, None], spacing_ends=s_bins[:, 1:, None])


def gen_losses():
    rs = np.random.RandomState(51)
    N = 20
    bins, ws = [], []
    for S in (256, 96, 48):
        b = np.sort(rs.uniform(0, 1, (N, S + 1)).astype(np.float32), axis=-1)
        b[:, 0], b[:, -1] = 0.0, 1.0
        w = rs.uniform(0, 1, (N, S)).astype(np.float32) ** 3
        w /= w.sum(-1, keepdims=True) * rs.uniform(1.0, 1.5, (N, 1)).astype(np.float32)
        bins.append(torch.from_numpy(b))
        ws.append(torch.from_numpy(w.astype(np.float32)).requires_grad_(True))
    # share some edges between levels so searchsorted ties are exercised
    bins[1][:, 10] = bins[2][:, 5]
    bins[0][:, 100] = bins[2][:, 20]
    bins = [torch.sort(b, dim=-1)[0] for b in bins]
    rsl = [_mk_samples(b) for b in bins]
    wl = [w[..., None] for w in ws]
    li = interlevel_loss(wl, rsl)
    ld = distortion_loss(wl, rsl)
    (li + 0.5 * ld).backward()
    save("losses", s_bins0=bins[0], s_bins1=bins[1], s_bins2=bins[2], w0=ws[0], w1=ws[1], w2=ws[2], interlevel=li,
         distortion=ld, dw0=ws[0].grad, dw1=ws[1].grad, dw2=ws[2].grad)


def gen_pipeline():
    cfg = small_cfg(main_log2=10, prop_log2=8, num_images=5)
    seed, std = 9, 0.5
    params = orc.init_params(cfg, seed=seed, table_std=std)
    fld, props = build_reference(cfg, params)
    N = 16
    o, d, cam, tgt = orc.synthetic_rays(N, cfg.num_images, seed=3)
    o[N // 2 :] *= 6.0  # half the rays start outside the unit cube (config-5 style unbounded rays)
    rs = np.random.RandomState(61)
    jit = [torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32)) for _ in range(3)]
    out = {"seed": seed, "table_std": std, "main_log2": 10, "prop_log2": 8, "num_images": cfg.num_images,
           "origins": o, "directions": d, "cams": cam, "target": tgt, "j0": jit[0], "j1": jit[1], "j2": jit[2]}
    sampler = ProposalNetworkSampler(num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96),
                                     num_proposal_network_iterations=2, single_jitter=True)
    collider = NearFarCollider(0.05, 1000.0)
    for mode in ("train", "eval"):
        training = mode == "train"
        for m in (fld, props, sampler, collider):
            m.train(training)
        fld.zero_grad()
        props.zero_grad()
        rgb_r = RGBRenderer("last_sample")
        rgb_r.train(training)
        rb = RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((N, 1), 1e-6), camera_indices=cam[:, None])
        rb = collider(rb)
        with replay_rand(jit if training else []), record_searchsorted() as rec:
            rsamp, wl, rsl = sampler(rb, density_fns=[p.density_fn for p in props])
            inds = [c for c in rec.calls]
        fo = fld(rsamp)
        w = rsamp.get_weights(fo[FieldHeadNames.DENSITY])
        wl.append(w)
        rsl.append(rsamp)
        rgb = rgb_r(rgb=fo[FieldHeadNames.RGB], weights=w)
        with torch.no_grad():
            depth = DepthRenderer("median")(weights=w, ray_samples=rsamp)
        dexp = DepthRenderer("expected")(weights=w, ray_samples=rsamp)
        acc = AccumulationRenderer()(weights=w)
        out[f"{mode}_rgb"], out[f"{mode}_depth"], out[f"{mode}_expected_depth"], out[f"{mode}_acc"] = rgb, depth, dexp, acc
        out[f"{mode}_inds1"], out[f"{mode}_inds2"] = inds[0], inds[1]
        for i in range(3):
            out[f"{mode}_w{i}"] = wl[i][..., 0]
            out[f"{mode}_s_bins{i}"] = torch.cat([rsl[i].spacing_starts[..., 0], rsl[i].spacing_ends[:, -1:, 0]], -1)
            out[f"{mode}_t_bins{i}"] = torch.cat([rsl[i].frustums.starts[..., 0], rsl[i].frustums.ends[:, -1:, 0]], -1)
        out[f"{mode}_density"] = fo[FieldHeadNames.DENSITY][..., 0]
        out[f"{mode}_rgb_samples"] = fo[FieldHeadNames.RGB]
        for i in range(2):
            out[f"{mode}_prop_depth_{i}"] = DepthRenderer("median")(weights=wl[i], ray_samples=rsl[i])
        if training:
            l_rgb = torch.nn.functional.mse_loss(tgt, rgb)
            l_int = interlevel_loss(wl, rsl)
            l_dist = 0.002 * distortion_loss(wl, rsl)
            (l_rgb + l_int + l_dist).backward()
            out.update(loss_rgb=l_rgb, loss_interlevel=l_int, loss_distortion=l_dist)
            out["g_main_table"] = fld.mlp_base.model[0].hash_table.grad
            out["g_emb"] = fld.embedding_appearance.embedding.weight.grad
            for j in range(2):
                out[f"g_base_W{j}"] = fld.mlp_base.model[1].layers[j].weight.grad
                out[f"g_base_b{j}"] = fld.mlp_base.model[1].layers[j].bias.grad
            for j in range(3):
                out[f"g_head_W{j}"] = fld.mlp_head.layers[j].weight.grad
                out[f"g_head_b{j}"] = fld.mlp_head.layers[j].bias.grad
            for i, p in enumerate(props):
                out[f"g_prop{i}_table"] = p.encoding.hash_table.grad
                for j in range(2):
                    out[f"g_prop{i}_W{j}"] = p.mlp_base[1].layers[j].weight.grad
                    out[f"g_prop{i}_b{j}"] = p.mlp_base[1].layers[j].bias.grad
    save("pipeline", **out)


def gen_camera_opt():
    """nerfacto with its DEFAULT camera optimiser (CameraOptimizerConfig(mode="SO3xR3"), method_configs.py:102; SE3 as well):
    the pose correction is applied to the ray bundle (camera_optimizers.py:148-153) and the loss gradient flows back through
    positions = o + d t (hash encoding, selector, contraction) of all three sampling levels into `pose_adjustment`.
    Fixture: the corrected rays, rgb, the three losses + regulariser, and dL/dpose_adjustment."""
    from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig

    cfg = small_cfg(main_log2=10, prop_log2=8, num_images=5)
    seed, std = 12, 0.5
    params = orc.init_params(cfg, seed=seed, table_std=std)
    N = 24
    o, d, cam, tgt = orc.synthetic_rays(N, cfg.num_images, seed=5)
    o[N // 2:] *= 4.0  # half the rays start outside the unit cube: contraction Jacobian on the gradient path
    rs = np.random.RandomState(71)
    jit = [torch.from_numpy(rs.uniform(0, 1, (N, 1)).astype(np.float32)) for _ in range(3)]
    pose = (rs.standard_normal((cfg.num_images, 6)) * np.array([0.05] * 3 + [0.1] * 3)).astype(np.float32)
    pose[1, 3:] = 0.0  # a camera with no rotation: the clamp branch of exp_map_SO3xR3
    out = {"seed": seed, "table_std": std, "main_log2": 10, "prop_log2": 8, "num_images": cfg.num_images, "origins": o,
           "directions": d, "cams": cam, "target": tgt, "j0": jit[0], "j1": jit[1], "j2": jit[2], "pose_adjustment": pose}
    for mode in ("SO3xR3", "SE3"):
        fld, props = build_reference(cfg, params)
        cam_opt = CameraOptimizerConfig(mode=mode).setup(num_cameras=cfg.num_images, device="cpu")
        with torch.no_grad():
            cam_opt.pose_adjustment.copy_(torch.from_numpy(pose))
        sampler = ProposalNetworkSampler(num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96),
                                         num_proposal_network_iterations=2, single_jitter=True)
        collider = NearFarCollider(0.05, 1000.0)
        rgb_r = RGBRenderer("last_sample")
        for m in (fld, props, sampler, collider, rgb_r, cam_opt):
            m.train(True)
        rb = RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.full((N, 1), 1e-6), camera_indices=cam[:, None])
        rb = collider(rb)
        cam_opt.apply_to_raybundle(rb)
        with replay_rand(jit):
            rsamp, wl, rsl = sampler(rb, density_fns=[p.density_fn for p in props])
        fo = fld(rsamp)
        w = rsamp.get_weights(fo[FieldHeadNames.DENSITY])
        wl.append(w)
        rsl.append(rsamp)
        rgb = rgb_r(rgb=fo[FieldHeadNames.RGB], weights=w)
        loss = {"rgb": torch.nn.functional.mse_loss(tgt, rgb), "interlevel": interlevel_loss(wl, rsl),
                "distortion": 0.002 * distortion_loss(wl, rsl)}
        reg = {}
        cam_opt.get_loss_dict(reg)
        loss["camera_opt_regularizer"] = reg["camera_opt_regularizer"]
        rb.origins.retain_grad()
        rb.directions.retain_grad()
        sum(loss.values()).backward()
        out.update({f"{mode}_origins": rb.origins, f"{mode}_directions": rb.directions, f"{mode}_rgb": rgb,
                    f"{mode}_d_origins": rb.origins.grad, f"{mode}_d_directions": rb.directions.grad,
                    f"{mode}_g_pose": cam_opt.pose_adjustment.grad, f"{mode}_g_main_table": fld.mlp_base.model[0].hash_table.grad,
                    f"{mode}_losses": torch.stack([loss[k] for k in ("rgb", "interlevel", "distortion", "camera_opt_regularizer")])})
    save("camera_opt", **out)


def gen_raygen():
    rs = np.random.RandomState(71)
    C, H, W = 3, 40, 56
    q = rs.standard_normal((C, 3, 3))
    rot = np.stack([np.linalg.qr(m)[0] for m in q]).astype(np.float32)
    c2w = np.concatenate([rot, rs.standard_normal((C, 3, 1)).astype(np.float32)], axis=-1)
    fx = rs.uniform(40, 60, (C,)).astype(np.float32)
    fy = rs.uniform(40, 60, (C,)).astype(np.float32)
    cx = np.full((C,), W / 2, np.float32) + rs.uniform(-1, 1, (C,)).astype(np.float32)
    cy = np.full((C,), H / 2, np.float32) + rs.uniform(-1, 1, (C,)).astype(np.float32)
    cams = Cameras(camera_to_worlds=torch.from_numpy(c2w), fx=torch.from_numpy(fx), fy=torch.from_numpy(fy),
                   cx=torch.from_numpy(cx), cy=torch.from_numpy(cy), width=W, height=H, camera_type=CameraType.PERSPECTIVE)
    gen = RayGenerator(cams)
    idx = np.stack([rs.randint(0, C, 128), rs.randint(0, H, 128), rs.randint(0, W, 128)], -1).astype(np.int64)
    idx[0] = [0, 0, 0]
    idx[1] = [C - 1, H - 1, W - 1]
    rb = gen(torch.from_numpy(idx))
    save("raygen", c2w=c2w, fx=fx, fy=fy, cx=cx, cy=cy, hw=np.array([H, W]), ray_indices=idx, origins=rb.origins,
         directions=rb.directions, pixel_area=rb.pixel_area, camera_indices=rb.camera_indices,
         directions_norm=rb.metadata["directions_norm"])


def gen_vanilla():
    """BASELINE configs[0] (vanilla NeRF, the reference's CPU-runnable case): NeRFEncoding, the 8x256 skip MLP field,
    UniformSampler(64) + PDFSampler(128, include_original=True) and the NeRFModel.get_outputs wiring
    (models/vanilla_nerf.py:139-196), composed from the reference's own modules with the oracle's seeded parameters."""
    from nerfstudio.field_components.encodings import NeRFEncoding
    from nerfstudio.fields.vanilla_nerf_field import NeRFField
    from nerfstudio.model_components.ray_samplers import PDFSampler, UniformSampler

    from oracle import vanilla_oracle as van

    cfg = van.VanillaCfg()
    rs = np.random.RandomState(91)
    out = {}
    # encoding known-answer block
    x = torch.from_numpy(rs.uniform(-1.5, 1.5, (7, 3)).astype(np.float32))
    out["enc_x"] = x
    out["enc_pos"] = NeRFEncoding(3, 10, 0.0, 8.0, include_input=True)(x)
    out["enc_dir"] = NeRFEncoding(3, 4, 0.0, 4.0, include_input=True)(x)
    out["enc_plain"] = NeRFEncoding(3, 6, 0.0, 5.0, include_input=False)(x)
    # the two fields with oracle-initialised parameters
    params = {}
    params.update(van.init_field_params(cfg, 92, "field_coarse."))
    params.update(van.init_field_params(cfg, 93, "field_fine."))
    pe = NeRFEncoding(3, cfg.pos_frequencies, 0.0, cfg.pos_max_exp, include_input=True)
    de = NeRFEncoding(3, cfg.dir_frequencies, 0.0, cfg.dir_max_exp, include_input=True)
    fields = {}
    for name in ("field_coarse", "field_fine"):
        f = NeRFField(position_encoding=pe, direction_encoding=de)
        sd = {k[len(name) + 1:]: v for k, v in params.items() if k.startswith(name + ".")}
        missing, unexpected = f.load_state_dict(sd, strict=False)
        assert not unexpected and not missing, (missing, unexpected)
        fields[name] = f
    # the parameters are NOT stored (4.4 MB): the oracle regenerates them from the seeds; a checksum pins that
    out["param_checksum"] = torch.stack([v.double().sum() for v in params.values()])
    N = 12
    o = torch.from_numpy((rs.standard_normal((N, 3)) * 0.3 + np.array([0, 0, -4.0])).astype(np.float32))
    d = torch.from_numpy(rs.standard_normal((N, 3)).astype(np.float32) * 0.15 + torch.tensor([0.0, 0, 1.0]).numpy())
    d = d / d.norm(dim=-1, keepdim=True)
    tgt = torch.from_numpy(rs.uniform(0, 1, (N, 3)).astype(np.float32))
    out.update(origins=o, directions=d, target=tgt)
    # vanilla NeRF keeps the samplers' default single_jitter=False: one draw per bin edge (ray_samplers.py:107, :323)
    j0 = torch.from_numpy(rs.uniform(0, 1, (N, cfg.num_coarse_samples + 1)).astype(np.float32))
    j1 = torch.from_numpy(rs.uniform(0, 1, (N, cfg.num_importance_samples + 1)).astype(np.float32))
    out.update(j0=j0, j1=j1)
    rgb_r, acc_r, dep_r = RGBRenderer(background_color="white"), AccumulationRenderer(), DepthRenderer()
    for mode in ("train", "eval"):
        training = mode == "train"
        rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(N, 1))
        rb = NearFarCollider(cfg.near_plane, cfg.far_plane, reset_near_plane=False)(rb)
        su, sp = UniformSampler(num_samples=cfg.num_coarse_samples), PDFSampler(num_samples=cfg.num_importance_samples)
        for m in (su, sp, rgb_r, *fields.values()):
            m.train(training)
        for p_ in params.values():
            p_.requires_grad_(True)
        for f in fields.values():
            f.zero_grad()
        with replay_rand([j0] if training else []):
            rs_u = su(rb)
        fo_c = fields["field_coarse"].forward(rs_u)
        w_c = rs_u.get_weights(fo_c[FieldHeadNames.DENSITY])
        with replay_rand([j1] if training else []), record_searchsorted() as rec:
            rs_p = sp(rb, rs_u, w_c)
        fo_f = fields["field_fine"].forward(rs_p)
        w_f = rs_p.get_weights(fo_f[FieldHeadNames.DENSITY])
        res = {"rgb_coarse": rgb_r(rgb=fo_c[FieldHeadNames.RGB], weights=w_c), "rgb_fine": rgb_r(rgb=fo_f[FieldHeadNames.RGB], weights=w_f),
               "accumulation_coarse": acc_r(w_c), "accumulation_fine": acc_r(w_f), "depth_coarse": dep_r(w_c, rs_u),
               "depth_fine": dep_r(w_f, rs_p), "weights_coarse": w_c[..., 0], "weights_fine": w_f[..., 0],
               "density_coarse": fo_c[FieldHeadNames.DENSITY][..., 0], "rgb_samples_coarse": fo_c[FieldHeadNames.RGB],
               "pdf_inds": rec.calls[0],
               "s_bins_fine": torch.cat([rs_p.spacing_starts[..., 0], rs_p.spacing_ends[:, -1:, 0]], -1),
               "t_bins_fine": torch.cat([rs_p.frustums.starts[..., 0], rs_p.frustums.ends[:, -1:, 0]], -1),
               "t_bins_coarse": torch.cat([rs_u.frustums.starts[..., 0], rs_u.frustums.ends[:, -1:, 0]], -1)}
        if training:
            loss = torch.mean((tgt - res["rgb_coarse"]) ** 2) + torch.mean((tgt - res["rgb_fine"]) ** 2)
            loss.backward()
            res["loss"] = loss
            # gradients: 4.4 MB in full -> per tensor the L2 norm, the sum and 48 elements at seeded positions
            pick = np.random.RandomState(94)
            for name, f in fields.items():
                for k, p_ in f.named_parameters():
                    g_ = p_.grad.reshape(-1)
                    idx = torch.from_numpy(pick.randint(0, g_.numel(), 48))
                    res[f"gidx_{name}.{k}"] = idx
                    res[f"gval_{name}.{k}"] = g_[idx]
                    res[f"gstat_{name}.{k}"] = torch.stack([g_.double().norm(), g_.double().sum()])
        for k, v in res.items():
            out[f"{mode}_{k}"] = v
    save("vanilla", **out)


def gen_schedulers():
    """Learning rates torch's LambdaLR sets with the reference's ExponentialDecayScheduler (engine/schedulers.py:109-142),
    read off the optimiser after `step` x (optimizer.step(); scheduler.step())."""
    from nerfstudio.engine.schedulers import ExponentialDecayScheduler, ExponentialDecaySchedulerConfig

    steps = [0, 1, 10, 999, 1000, 30000, 199999, 200000, 250000]
    out = {}
    for name, cfg in (("nerfacto", ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=200000)),
                      ("warm_cos", ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=5000, warmup_steps=100,
                                                                   lr_pre_warmup=1e-8)),
                      ("warm_lin", ExponentialDecaySchedulerConfig(lr_final=None, max_steps=5000, warmup_steps=100,
                                                                   ramp="linear"))):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-2)
        sch = ExponentialDecayScheduler(cfg).get_scheduler(opt, 1e-2)
        lrs = []
        for s_ in range(max(steps) + 1):
            if s_ in steps:
                lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[name] = np.array(lrs, dtype=np.float64)
    save("schedulers", steps=np.array(steps), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["kat", "hashgrid", "fields", "samplers", "render", "losses", "pipeline", "raygen",
                             "schedulers", "vanilla"]
    for w in which:
        globals()["gen_" + w]()
