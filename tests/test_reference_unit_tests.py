"""The reference's own unit tests, UNMODIFIED, against this package's mirror classes.

tests/refunit/run.py (own process) aliases `nerfstudio.<module>` to this package's module of the same name — falling back
to the reference's symbol for classes this package does not have (other methods' heads and encodings the same test files
import) —, replaces the kernel wrappers by the oracle's torch restatements (tests/cpu_kernels.py) and calls every test
function of the reference's test files for the in-scope components: field heads, MLP, embedding, encodings, NerfactoField,
the samplers, the renderers, Frustums / RaySamples, TensorDataclass. They pass as written, except where this package
refuses a configuration on purpose (implementation='torch' / 'tcnn', encodings outside the nerfacto shapes): those must fail
with exactly that refusal. Authoring container only (needs /root/reference)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="needs /root/reference")

FILES = ["field_components/test_field_outputs.py", "field_components/test_mlp.py", "field_components/test_embedding.py",
         "field_components/test_encodings.py", "field_components/test_fields.py", "model_components/test_ray_sampler.py",
         "model_components/test_renderers.py", "cameras/test_rays.py", "utils/test_tensor_dataclass.py"]
# what each file must actually have exercised of THIS package (the proxy would otherwise let a test pass on the reference)
MIRRORED = {"field_components/test_field_outputs.py": {"DensityFieldHead", "FieldHead", "FieldHeadNames", "RGBFieldHead"},
            "field_components/test_mlp.py": {"MLP"}, "field_components/test_embedding.py": {"Embedding"},
            "field_components/test_fields.py": {"Frustums", "NerfactoField", "RaySamples"},
            "model_components/test_ray_sampler.py": {"NearFarCollider", "PDFSampler", "RayBundle", "UniformSampler"},
            "model_components/test_renderers.py": {"Frustums", "RaySamples"}, "cameras/test_rays.py": {"Frustums"},
            "utils/test_tensor_dataclass.py": {"TensorDataclass"}}
# refusals by design: the test asks this package for something it says it does not provide (never a silent fallback)
REFUSED = {"field_components/test_encodings.py::test_tensor_hash_encoder": ("ValueError", "implementation='hip' only"),
           "field_components/test_encodings.py::test_tensor_sh_encoder": ("ValueError", "SHEncoding(levels=4)"),
           "field_components/test_encodings.py::test_nerf_encoder": ("ValueError", "3-vectors")}


@pytest.fixture(scope="module")
def results():
    proc = subprocess.run([sys.executable, os.path.join(HERE, "refunit", "run.py"), *FILES], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return json.loads([line for line in proc.stdout.splitlines() if line.startswith("{")][-1])


def test_the_reference_tests_ran_on_this_packages_classes(results):
    for f, want in MIRRORED.items():
        assert want <= set(results[f]["mirrored"]), (f, results[f]["mirrored"])
    ran = [k for k in results if "::" in k]
    assert len(ran) >= 33, ran


def test_the_reference_unit_tests_pass_unmodified(results):
    failed = {k: v for k, v in results.items() if "::" in k and v[0] != "pass" and k not in REFUSED}
    assert not failed, failed
    passed = [k for k, v in results.items() if "::" in k and v[0] == "pass"]
    for must in ("field_components/test_fields.py::test_nerfacto_field", "model_components/test_ray_sampler.py::test_pdf_sampler",
                 "model_components/test_renderers.py::test_rgb_renderer", "model_components/test_renderers.py::test_depth_renderer",
                 "cameras/test_rays.py::test_frustum_get_gaussian_blob", "utils/test_tensor_dataclass.py::test_broadcasting"):
        assert must in passed, must


def test_out_of_scope_configurations_are_refused_loudly(results):
    for k, (exc, text) in REFUSED.items():
        assert results[k][0] == "fail" and results[k][1] == exc and text in results[k][2], (k, results[k])
