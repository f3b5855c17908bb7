"""CPU tests of the drop-in boundary against the REFERENCE'S OWN containers (authoring container only: needs
/root/reference; skipped on the GPU box). The mirror modules must take the reference's `RayBundle` / `RaySamples` /
`Frustums` (cameras/rays.py:34-295 — TensorDataclasses without this package's `pack` field) exactly as they take this
package's: every call below has to get as far as the kernel launch, i.e. fail with the RuntimeError of
`_native.require_cuda` ("runs on an MI355X only") and not with an AttributeError / TypeError on the container.
Plus: the `nerfacto-hip` method plugin (nerfstudio_amd/plugin.py) reads only fields the reference's NerfactoModelConfig
has, and builds the hip modules from them."""
import ast
import os
import sys
import types
from types import SimpleNamespace

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "nerfstudio")), reason="needs /root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref():
    """The reference's container / collider classes, imported read-only with the stubs of tests/golden/_refstubs."""
    sys.path.insert(0, os.path.join(HERE, "golden", "_refstubs"))
    sys.path.insert(1, REF)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    from nerfstudio.cameras.rays import Frustums, RayBundle, RaySamples
    from nerfstudio.model_components.scene_colliders import NearFarCollider

    return SimpleNamespace(Frustums=Frustums, RayBundle=RayBundle, RaySamples=RaySamples, NearFarCollider=NearFarCollider)


def _ref_bundle(ref, n=6):
    g = torch.Generator().manual_seed(0)
    rb = ref.RayBundle(origins=torch.randn(n, 3, generator=g) * 0.3,
                       directions=torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1),
                       pixel_area=torch.full((n, 1), 1e-6), camera_indices=torch.zeros(n, 1, dtype=torch.long))
    return ref.NearFarCollider(0.05, 1000.0)(rb)


def _ref_samples(ref, rb, s=8):
    edges = torch.linspace(0.0, 1.0, s + 1)[None].expand(len(rb), s + 1)
    t = 0.05 + edges * 2.0
    return rb.get_ray_samples(bin_starts=t[..., :-1, None], bin_ends=t[..., 1:, None], spacing_starts=edges[..., :-1, None],
                              spacing_ends=edges[..., 1:, None], spacing_to_euclidean_fn=lambda x: 0.05 + 2.0 * x)


def _reaches_the_kernel(fn):
    with pytest.raises(RuntimeError, match="MI355X"):
        fn()


def test_samplers_take_the_reference_ray_bundle_and_samples(ref):
    from nerfstudio_amd.model_components.ray_samplers import PDFSampler, ProposalNetworkSampler, UniformLinDispPiecewiseSampler, UniformSampler

    rb = _ref_bundle(ref)
    assert not hasattr(rb, "pack") and type(rb).__module__.startswith("nerfstudio.cameras")
    for cls in (UniformLinDispPiecewiseSampler, UniformSampler):
        sampler = cls(num_samples=8, single_jitter=True)
        _reaches_the_kernel(lambda: sampler(ray_bundle=rb))
    rs = _ref_samples(ref, rb)
    assert not hasattr(rs, "pack")
    pdf = PDFSampler(num_samples=4, include_original=False, single_jitter=True)
    _reaches_the_kernel(lambda: pdf(ray_bundle=rb, ray_samples=rs, weights=torch.rand(len(rb), 8, 1)))
    prop = ProposalNetworkSampler(num_proposal_samples_per_ray=(8,), num_nerf_samples_per_ray=4,
                                  num_proposal_network_iterations=1, single_jitter=True)
    _reaches_the_kernel(lambda: prop(ray_bundle=rb, density_fns=[lambda x: x.sum(-1, keepdim=True)]))


def test_fields_renderers_and_losses_take_the_reference_ray_samples(ref):
    from nerfstudio_amd.cameras.rays import t_bins_of
    from nerfstudio_amd.field_components.spatial_distortions import SceneContraction
    from nerfstudio_amd.fields.base_field import point_spec
    from nerfstudio_amd.fields.density_fields import HashMLPDensityField
    from nerfstudio_amd.fields.nerfacto_field import NerfactoField
    from nerfstudio_amd.model_components.losses import distortion_loss, interlevel_loss
    from nerfstudio_amd.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer

    rb = _ref_bundle(ref)
    rs = _ref_samples(ref, rb)
    # without a pack the kernels get materialised positions = the reference's own frustum centres
    spec, shape = point_spec(rs)
    assert shape == (6, 8) and spec.positions is not None
    assert torch.equal(spec.positions, rs.frustums.get_positions().reshape(-1, 3))
    assert torch.equal(t_bins_of(rs)[:, :-1], rs.frustums.starts[..., 0]) and t_bins_of(rs).shape == (6, 9)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    sc = SceneContraction(order=float("inf"))
    field = NerfactoField(aabb, num_images=3, log2_hashmap_size=8, spatial_distortion=sc, average_init_density=0.01)
    prop = HashMLPDensityField(aabb, hidden_dim=16, num_levels=5, max_res=64, log2_hashmap_size=8, spatial_distortion=sc)
    _reaches_the_kernel(lambda: field(rs))
    _reaches_the_kernel(lambda: prop.get_density(rs))
    _reaches_the_kernel(lambda: prop.density_fn(rs.frustums.get_positions()))
    w = torch.rand(6, 8, 1)
    _reaches_the_kernel(lambda: RGBRenderer("last_sample")(rgb=torch.rand(6, 8, 3), weights=w))
    _reaches_the_kernel(lambda: RGBRenderer.combine_rgb(torch.rand(6, 8, 3), w, background_color="white"))
    _reaches_the_kernel(lambda: AccumulationRenderer()(weights=w))
    for method in ("median", "expected"):
        _reaches_the_kernel(lambda: DepthRenderer(method)(weights=w, ray_samples=rs))
    _reaches_the_kernel(lambda: interlevel_loss([w, w], [rs, rs]))
    _reaches_the_kernel(lambda: distortion_loss([w], [rs]))


def test_own_containers_behave_like_the_reference_tensor_dataclasses(ref):
    """Same constructor arguments -> same batch shape, same broadcast fields, same results of indexing / reshape /
    flatten / broadcast_to as the reference's TensorDataclass (utils/tensor_dataclass.py:27-331)."""
    from nerfstudio_amd.cameras import rays as mine

    kw = dict(origins=torch.randn(4, 1, 3), directions=torch.randn(1, 5, 3), pixel_area=torch.ones(1, 1),
              camera_indices=torch.arange(4)[:, None, None].expand(4, 5, 1), nears=torch.rand(4, 5, 1), fars=torch.rand(4, 5, 1) + 1,
              metadata={"directions_norm": torch.rand(4, 5, 1)})
    a, b = ref.RayBundle(**kw), mine.RayBundle(**kw)
    ops = [lambda x: x, lambda x: x[1], lambda x: x[:, 2:4], lambda x: x[..., 0], lambda x: x.flatten(), lambda x: x.reshape((2, 10)),
           lambda x: x.flatten()[3:7], lambda x: x[torch.tensor([0, 2])]]
    for op in ops:
        ra, rb_ = op(a), op(b)
        assert tuple(ra.shape) == tuple(rb_.shape) and ra.ndim == rb_.ndim and ra.size == rb_.size and len(ra) == len(rb_)
        for name in ("origins", "directions", "pixel_area", "camera_indices", "nears", "fars"):
            assert torch.equal(getattr(ra, name), getattr(rb_, name)), name
        assert torch.equal(ra.metadata["directions_norm"], rb_.metadata["directions_norm"])
    fa = ref.Frustums(origins=kw["origins"], directions=kw["directions"], starts=torch.rand(4, 5, 1), ends=torch.rand(4, 5, 1) + 1,
                      pixel_area=kw["pixel_area"])
    fb = mine.Frustums(origins=fa.origins, directions=fa.directions, starts=fa.starts, ends=fa.ends, pixel_area=fa.pixel_area)
    assert tuple(fa.shape) == tuple(fb.shape) == (4, 5) and torch.equal(fa.get_positions(), fb.get_positions())
    assert torch.equal(fa.broadcast_to((2, 4, 5)).origins, fb.broadcast_to((2, 4, 5)).origins)
    sa, sb = ref.RaySamples(frustums=fa, camera_indices=kw["camera_indices"]), mine.RaySamples(frustums=fb, camera_indices=kw["camera_indices"])
    assert tuple(sa[2].shape) == tuple(sb[2].shape) == (5,) and torch.equal(sa[2].frustums.starts, sb[2].frustums.starts)
    with pytest.raises((ValueError, RuntimeError)):
        mine.RayBundle(origins=torch.zeros(4, 3), directions=torch.zeros(5, 3), pixel_area=torch.zeros(4, 1))  # shapes do not broadcast
    with pytest.raises(RuntimeError):
        b[0] = b[1]


def _reference_config_fields():
    """Field names of the reference's NerfactoModelConfig and its base ModelConfig, read from the sources (the module
    itself needs tyro / torchmetrics, which are not installed here)."""
    names = set()
    for path, cls in (("nerfstudio/models/nerfacto.py", "NerfactoModelConfig"), ("nerfstudio/models/base_model.py", "ModelConfig")):
        tree = ast.parse(open(os.path.join(REF, path)).read())
        node = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == cls)
        names |= {s.target.id for s in node.body if isinstance(s, ast.AnnAssign)}
    return names


def test_method_plugin_uses_only_reference_config_fields_and_builds_the_hip_modules():
    import inspect
    import re

    from nerfstudio_amd import plugin
    from nerfstudio_amd.fields.density_fields import HashMLPDensityField
    from nerfstudio_amd.fields.nerfacto_field import NerfactoField
    from nerfstudio_amd.model_components.ray_samplers import ProposalNetworkSampler
    from nerfstudio_amd.nerfacto import NerfactoModelConfig

    ref_fields = _reference_config_fields()
    used = set(re.findall(r"cfg\.([a-z_0-9]+)", inspect.getsource(plugin.install_hip_modules) + inspect.getsource(plugin.hip_loss_terms)))
    assert used and used <= ref_fields, sorted(used - ref_fields)
    # this package's own config mirrors the reference's field names (it stands in for it here: the reference's needs tyro)
    mine = {f for f in NerfactoModelConfig.__dataclass_fields__}
    assert used <= mine, sorted(used - mine)
    cfg = NerfactoModelConfig(log2_hashmap_size=8, proposal_net_args_list=[
        {"hidden_dim": 16, "log2_hashmap_size": 7, "num_levels": 5, "max_res": r, "use_linear": False} for r in (32, 64)])
    model = SimpleNamespace(config=cfg, scene_box=SimpleNamespace(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=7)
    plugin.install_hip_modules(model)
    assert isinstance(model.field, NerfactoField) and model.field.embedding_appearance.embedding.weight.shape == (7, 32)
    assert len(model.proposal_networks) == 2 and all(isinstance(p, HashMLPDensityField) for p in model.proposal_networks)
    assert len(model.density_fns) == 2 and isinstance(model.proposal_sampler, ProposalNetworkSampler)
    assert model.renderer_rgb.background_color == cfg.background_color
    # parameter names are the reference's torch-path names (checkpoint interchange)
    names = dict(model.field.named_parameters())
    assert "mlp_base.model.0.hash_table" in names and "mlp_head.layers.2.weight" in names
    # nerfstudio itself (tyro, viser, torchmetrics) is not installed here: the entry point says so instead of half-working
    with pytest.raises(ImportError, match="nerfstudio"):
        plugin.nerfacto_hip()
    # registration strings a maintainer uses
    text = open(os.path.join(os.path.dirname(HERE), "pyproject.toml")).read()
    assert 'nerfacto-hip = "nerfstudio_amd.plugin:nerfacto_hip_spec"' in text and "nerfstudio.method_configs" in text
    assert 'instant-ngp-hip = "nerfstudio_amd.plugin:instant_ngp_hip_spec"' in text
    with pytest.raises(AttributeError):
        plugin.no_such_spec  # noqa: B018  (the lazy attributes are the two specifications only)


def test_field_output_dicts_interchange_with_the_reference_enum(ref):
    """plugin.HipNerfactoModel inherits the REFERENCE's get_outputs, which indexes the dictionary a field of THIS package
    returned with the reference's own `FieldHeadNames` class (models/nerfacto.py:304-324); and the reference's
    scale_gradients_by_distance_squared rebuilds such a dictionary key by key (model_components/losses.py:534-569). The two
    enum classes are distinct objects: members of the same name must find each other in both directions."""
    from nerfstudio.field_components.field_heads import FieldHeadNames as RefNames
    from nerfstudio.model_components.losses import scale_gradients_by_distance_squared

    from nerfstudio_amd.field_components.field_heads import FieldHeadNames as OurNames

    assert {m.name: m.value for m in RefNames} == {m.name: m.value for m in OurNames}
    ours = {OurNames.DENSITY: torch.ones(2, 3, 1), OurNames.RGB: torch.zeros(2, 3, 3), OurNames.NORMALS: torch.ones(2, 3, 3)}
    for m in (RefNames.DENSITY, RefNames.RGB, RefNames.NORMALS):
        assert m in ours and ours[m] is ours[OurNames[m.name]]
    assert RefNames.PRED_NORMALS not in ours and OurNames.RGB != RefNames.DENSITY and not (OurNames.RGB != RefNames.RGB)
    theirs = {RefNames.DENSITY: 1, RefNames.RGB: 2}
    assert theirs[OurNames.DENSITY] == 1 and theirs[OurNames.RGB] == 2 and OurNames.SH not in theirs
    assert len({OurNames.RGB, RefNames.RGB}) == 1 and OurNames("rgb") is OurNames.RGB
    # the reference's gradient scaling over a dictionary of this package's keys and the reference's RaySamples
    rs = _ref_samples(ref, _ref_bundle(ref, n=2), s=3)
    outs = {OurNames.DENSITY: torch.ones(2, 3, 1, requires_grad=True), OurNames.RGB: torch.ones(2, 3, 3, requires_grad=True)}
    scaled = scale_gradients_by_distance_squared(outs, rs)
    assert torch.equal(scaled[RefNames.DENSITY], outs[OurNames.DENSITY]) and RefNames.RGB in scaled


def test_field_head_names_match_any_enum_of_that_name_only():
    """(no reference needed) equality is by class NAME + member name + value: an unrelated enum with an equal value is not it."""
    import enum

    from nerfstudio_amd.field_components.field_heads import FieldHeadNames as OurNames

    Same = enum.Enum("FieldHeadNames", {"RGB": "rgb", "DENSITY": "density"})
    Other = enum.Enum("Colours", {"RGB": "rgb"})
    assert OurNames.RGB == Same.RGB and Same.RGB == OurNames.RGB and {OurNames.RGB: 1}[Same.RGB] == 1
    assert OurNames.RGB != Other.RGB and Other.RGB not in {OurNames.RGB: 1} and OurNames.RGB != "rgb"
